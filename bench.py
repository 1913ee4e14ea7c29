#!/usr/bin/env python3
"""bench.py -- LocalBA windows/sec on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (the whole two-stage LocalBAPRVIDP solve, src/Optimizer.cpp:458-517 of
the reference) over one BATCH of synthetic windows per GPU.  Workload at every N: BASELINE.json configs[2]
(50 KF = 49 free + fixed predecessor / 5 000 inverse-depth landmarks / 30 000 EdgePRIDP + 49 PRV + 49 bias
edges, Gauss-Newton 5+10), `--batch` windows per GPU, inputs already resident in HBM when the timed region
starts (vba_batch_upload before, vba_batch_run timed).  Weak scaling: every rank solves its own batch, no
data-path collective (windows are independent); torch.distributed (RCCL) only carries the
barrier and the max-over-ranks time.

One JSON line on rank 0, with `roofline` (dominant kernel class, HIP events on the backend's own stream) and
`cpu_baseline` (the CPU oracle = restatement of the reference's g2o path, 1 core, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def bench_pose(args, rank, local_rank, world, dist, torch):
    """Extra measurement: frames/s of vba_pose_optimize on a batch of tracked frames (300 correspondences, last
    keyframe and last-frame variants alternating), end to end per call (host gather + H2D + the one kernel + D2H); the CPU
    oracle's PoseOptimization restatement on the same frames beside it."""
    import numpy as np
    from mc_slam_amd import synth, backend, shard
    nb = args.batch
    distinct = [synth.make_frame(seed=shard.window_seed(g), n_obs=300, last_is_frame=bool(i % 2))
                for i, g in enumerate(shard.window_ids(16 * world, rank, world))]
    frames = [distinct[i % len(distinct)] for i in range(nb)]
    ba = backend.LocalBA(local_rank)
    packed = ba.pose_pack(frames)
    meter = shard.ThroughputMeter(dist, torch.cuda.synchronize)
    for _ in range(args.warmup):
        ba.pose_run(packed)
    # the call updates the frames in place: the (Python-side) reset of the inputs between steps stays outside the clock
    total, dt = 0, 0.0
    for _ in range(args.steps):
        ba.pose_reset(packed)
        meter.start()
        ba.pose_call(packed)
        n_step, t_step = meter.stop(nb, device="cuda")
        total += n_step
        dt += t_step
    res = [b.get(s_) for b, s_ in zip(packed[2], packed[1])]
    out = None
    if rank == 0:
        cpu, verified = None, "batch self-consistent"
        for i, r in enumerate(res):
            j = i % len(distinct)
            if r.its_done != res[j].its_done or (r.nav != res[j].nav).any():
                raise SystemExit("bench: frame %d did not solve like its twin %d" % (i, j))
        if not args.no_cpu_baseline:
            import oracle_lib
            t0 = time.perf_counter()
            ros = [oracle_lib.pose_optimize(f) for f in distinct]
            tc = time.perf_counter() - t0
            for r, ro in zip(res, ros):
                if r.its_done != ro.its_done or (r.outlier != ro.outlier).any() or np.abs(r.nav[:3] - ro.nav[:3]).max() > 1e-6:
                    raise SystemExit("bench: a frame does not match the CPU oracle")
            verified += "; %d distinct frames == oracle (LM iterations, outlier bitmap, P 1e-6 m)" % len(distinct)
            cpu = {"value": len(distinct) / tc, "unit": "frames/s", "cores": 1, "kind": "port",
                   "sample": "%d frames, single thread, oracle/libvba_oracle.so" % len(distinct)}
        out = {"metric": "IMU-aided PoseOptimization frames/sec (300 correspondences) [extra measurement]", "value": total / dt, "unit": "frames/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "Optimizer::PoseOptimization(Frame*, KeyFrame*|Frame*, IMUPreintegrator, gw, bComputeMarg): 4 x optimize(10) LM + "
                                      "reclassification + marginals per frame", "frames_per_gpu_per_step": nb, "distinct_frames": len(distinct)},
               "roofline": None, "cpu_baseline": cpu, "verified": verified}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="windows per GPU per step (default: 4096 for c3, 2048 for c2, 16 for c4, 4 for gba, 4096 frames for pose)")
    ap.add_argument("--distinct", type=int, default=16, help="distinct seeded windows generated per rank (cycled to fill the batch)")
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c4", "gba", "pose"],
                    help="c3 = BASELINE configs[2] (the headline metric); c2 / c4 = configs[1] / configs[3], gba = map-scale global BA, pose = IMU-aided per-frame pose optimisation: extra measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()
    if args.batch is None:
        args.batch = {"c2": 2048, "c3": 4096, "c4": 16, "gba": 4, "pose": 4096}[args.workload]

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if os.environ.get("BENCH_TEST_ONE_GPU"):   # rehearsal of the N>1 path on a one-GPU box: every rank on cuda:0, gloo
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU path")
    torch.cuda.set_device(local_rank)

    from mc_slam_amd import synth, backend, shard

    if args.workload == "pose":
        return bench_pose(args, rank, local_rank, world, dist, torch)

    # synthetic windows of configs[2]; window w of the job belongs to rank w % world, seed 100 + w
    n_distinct = min(args.distinct, args.batch)
    gids = shard.window_ids(n_distinct * world, rank, world)
    make = {"c2": synth.config_c2, "c3": synth.config_c3, "c4": synth.config_c4, "gba": synth.config_gba}[args.workload]
    if args.workload in ("c4", "gba"):
        n_distinct = min(n_distinct, 2)
        gids = gids[:n_distinct]
    wins = [make(seed=shard.window_seed(g)) for g in gids]
    batch = [wins[i % len(wins)] for i in range(args.batch)]
    ba = backend.LocalBA(local_rank)
    ba.upload(batch)

    meter = shard.ThroughputMeter(dist, torch.cuda.synchronize)
    for _ in range(args.warmup):
        ba.run()
    meter.start()
    for _ in range(args.steps):
        ba.run()          # vba_batch_run returns after the stream has drained
    total_windows, dt = meter.stop(args.batch * args.steps, device="cuda")
    value = total_windows / dt

    # the timed work must be the real work: every window of the last timed run finished both stages, windows built
    # from the same seed agree bit for bit, and window 0 matches the CPU oracle (chi2 <= 1e-4 rel, translations <= 1e-6 m)
    sol, res = ba.download()
    for i, r in enumerate(res):
        j = i % len(wins)
        if r.status != 0 or r.its_done[0] < 1 or (min(r.its_done) < 1 and batch[0].protocol == 0) or r.its_done != res[j].its_done or r.chi2_vis != res[j].chi2_vis:
            raise SystemExit("bench: window %d did not solve like its twin %d: %s vs %s" % (i, j, (r.status, r.its_done), res[j].its_done))
    verified = "batch self-consistent"
    if rank == 0 and not args.no_cpu_baseline:
        import oracle_lib
        if args.workload in ("c4", "gba"):
            ok = True   # the oracle's dense solve takes minutes at n_p = 2985: full-size C4 parity is a pytest property test
        else:
            qo, ro = oracle_lib.solve(wins[0])
        ok = ok if args.workload in ("c4", "gba") else (ro.its_done == res[0].its_done and abs(ro.chi2_vis - res[0].chi2_vis) <= 1e-4 * ro.chi2_vis
              and np.abs(qo.kf_pose[:, :3] - sol[0].kf_pose[:, :3]).max() <= 1e-6)
        if not ok:
            raise SystemExit("bench: window 0 does not match the CPU oracle")
        if args.workload not in ("c4", "gba"):
            verified += "; window 0 == oracle (chi2 1e-4 rel, t 1e-6 m)"

    out = None
    if rank == 0:
        # roofline of the dominant kernel class: a separate profiled run (HIP events around every launch of the
        # class, on the backend's stream), never mixed into `value`
        ba.set_profile(True)
        ba.run()
        pf = ba.get_profile()
        ba.set_profile(False)
        _, res = ba.download()
        classes = {k: v for k, v in pf.items() if k != "total_ms"}
        dom = max(classes, key=lambda k: classes[k]["ms"])
        its = [sum(r.its_done) for r in res]
        # algorithmic work of the dominant class
        n_p = (6 if batch[0].variant == 0 else 15) * batch[0].n_kf_free
        solves = float(sum(its))
        if dom in ("factor", "trsv", "schur"):
            # dense FP64 factorisation of the reduced system: n^3/3 flop per solve; its HBM floor is the matrix
            # read + written once (n_p^2 * 8 B * 2)
            alg_bytes = solves * 2.0 * n_p * n_p * 8.0
        else:
            alg_bytes = classes[dom]["bytes"]
        dur_s = classes[dom]["ms"] * 1e-3
        launches = max(1, classes[dom]["launches"])
        achieved = alg_bytes / dur_s / 1e9 if dur_s > 0 else 0.0
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                    "frac": achieved / 8000.0, "traffic": None,
                    "avg_launch_ms": classes[dom]["ms"] / launches, "launches": launches,
                    "class_ms": {k: round(v["ms"], 4) for k, v in classes.items() if v["launches"]},
                    "profiled_total_ms": pf["total_ms"]}
        # HBM traffic of the dominant class from the committed rocprofv3 PMC passes (FETCH_SIZE corrected x2 per the
        # MI355X guide, + WRITE_SIZE), per class launch, when a summary for this batch size exists
        try:
            import glob
            tj = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_traffic.json")))[-1]
            tr = json.load(open(tj))
            if tr["batch"] == args.batch:
                kmap = {"schur": ["k_schur_all", "k_schur_diag", "k_schur_off"], "linearize": ["k_lin2"],
                        "factor": ["k_chol_step", "k_chol_panel", "k_chol_update", "k_chol_diag_ll", "k_chol_panel_ll"], "trsv": ["k_trsv"], "update": ["k_update"]}
                ks = [tr["kernels"][k] for k in kmap.get(dom, []) if k in tr["kernels"]]
                n_it = tr["kernels"]["k_schur_all" if "k_schur_all" in tr["kernels"] else "k_schur_diag"]["active_launches"]
                n_cls = tr["kernels"]["k_lin2"]["active_launches"] if dom == "linearize" else n_it
                roofline["traffic"] = sum((k["fetch_corrected"] + k["write"]) * k["active_launches"] for k in ks) / n_cls
                roofline["traffic_source"] = os.path.basename(tj)
        except Exception:
            pass
        # the dense solve (north_star: "MFMA utilisation reported against gfx950 peak"): FP64 flop the factorisation class
        # executed with v_mfma_f64_16x16x4 (tile products of the symbolic lists x 2*32^3) over its HIP-event time, against
        # the 78.6 TFLOP/s FP64 matrix peak of the MI355X
        if classes["factor"]["ms"] > 0:
            tf = classes["factor"]["flops"] / (classes["factor"]["ms"] * 1e-3) / 1e12
            roofline["mfma_factor"] = {"achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6, "dtype": "f64"}

        cpu = None
        if not args.no_cpu_baseline:
            import oracle_lib
            oracle_lib.lib()
            n_done, t_cpu = 0, 0.0
            while t_cpu < args.cpu_seconds and n_done < 4 * len(wins) and not (args.workload == "c4" and n_done >= 1) and args.workload != "gba":
                t1 = time.perf_counter()
                oracle_lib.solve(wins[n_done % len(wins)], solver_mode=1)
                t_cpu += time.perf_counter() - t1
                n_done += 1
            cpu = None if n_done == 0 else {"value": n_done / t_cpu, "unit": "windows/s", "cores": 1, "kind": "port",
                   "sample": "%d solves of the same C3 windows, single thread, oracle/libvba_oracle.so (restatement of "
                             "the reference's g2o path, -O3; the reference itself cannot be built here)" % n_done}
            if cpu is not None and args.workload in ("c2", "c3"):
                # the same oracle on every host core at once, one window per thread (the reference itself
                # solves on one thread, src/System.cpp:198; this is the generous reading).  ctypes releases the GIL.
                from concurrent.futures import ThreadPoolExecutor
                ncores = max(1, min(os.cpu_count() or 1, 64))
                per = 2
                t1 = time.perf_counter()
                with ThreadPoolExecutor(max_workers=ncores) as ex:
                    list(ex.map(lambda k: oracle_lib.solve(wins[k % len(wins)], solver_mode=1), range(per * ncores)))
                t_all = time.perf_counter() - t1
                cpu["all_cores"] = {"value": per * ncores / t_all, "unit": "windows/s", "cores": ncores,
                                    "sample": "%d solves, %d threads, one window per thread" % (per * ncores, ncores)}
        out = {
            "metric": {"c3": "LocalBA windows/sec (50 KF, 5k pts, 30k obs, IMU edges)",
                       "c2": "vision-only LocalBundleAdjustment windows/sec (20 KF, 2k pts, 12k obs) [extra measurement]",
                       "c4": "synthetic VI graph solves/sec (200 KF, 50k pts, 500k obs) [extra measurement]",
                       "gba": "global BA solves/sec (300 KF, 30k pts, 180k obs, IMU chain) [extra measurement]"}[args.workload],
            "value": value, "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": {"c3": "BASELINE configs[2]: LocalBAPRVIDP window, 50 KF (49 free) / 5000 IDP landmarks / "
                                         "30000 EdgePRIDP + 49 PRV + 49 bias edges, GN 5+10",
                                    "c2": "BASELINE configs[1]: vision-only LocalBundleAdjustment, 20 KF (18 free) / 2000 XYZ landmarks / "
                                          "12000 EdgeSE3ProjectXYZ, LM 5+10",
                                    "c4": "BASELINE configs[3]: synthetic VI graph, 200 KF / 50000 IDP landmarks / 500000 EdgePRIDP + IMU "
                                          "chain, GN 5+10",
                                    "gba": "GlobalBundleAdjustmentNavStatePRV, 300 KF / 30000 XYZ landmarks / 180000 EdgeNavStatePRPointXYZ "
                                           "+ IMU chain, LM optimize(10)"}[args.workload],
                       "windows_per_gpu_per_step": args.batch, "distinct_windows": len(wins),
                       "parallelism": "independent windows sharded %d per GPU, no data-path collective" % args.batch,
                       "mean_outer_iterations": float(np.mean(its))},
            "roofline": roofline, "cpu_baseline": cpu, "verified": verified,
        }
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()

"""N>1 path on CPU: 2 gloo ranks shard independent windows, solve them (with the CPU oracle standing in for the
device, since no GPU exists here) and gather the throughput exactly as bench.py does on RCCL."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mc_slam_amd import shard


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import oracle_lib
    from mc_slam_amd import synth
    ids = shard.window_ids(n_total, rank, world)
    wins = [synth.make_window(2, n_kf=6, n_fixed=1, n_pt=60, n_obs=240, seed=shard.window_seed(g)) for g in ids]
    meter = shard.ThroughputMeter(dist)
    meter.start()
    chi = [oracle_lib.solve(w)[1].chi2_vis for w in wins]
    total, dt = meter.stop(len(wins))
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.array([total, dt] + ids + chi))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_windows_and_agree_on_throughput(tmp_path):
    world, n_total = 2, 5
    mp.spawn(_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "r0.npy"); r1 = np.load(tmp_path / "r1.npy")
    assert r0[0] == r1[0] == n_total          # SUM of per-rank window counts
    assert r0[1] == r1[1] and r0[1] > 0       # MAX elapsed, identical on both ranks
    ids0 = list(r0[2:2 + 3].astype(int)); ids1 = list(r1[2:2 + 2].astype(int))
    assert sorted(ids0 + ids1) == list(range(n_total)) and ids0 == [0, 2, 4] and ids1 == [1, 3]
    # every rank solved different windows (distinct seeds -> distinct chi2)
    chis = list(r0[5:]) + list(r1[4:])
    assert len(set(np.round(chis, 6))) == n_total


def test_window_ids_cover_everything_once():
    for world in (1, 2, 4, 8):
        allw = sorted(sum((shard.window_ids(13, r, world) for r in range(world)), []))
        assert allw == list(range(13))

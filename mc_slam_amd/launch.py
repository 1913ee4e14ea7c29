"""Self-launcher of the one-process-per-GPU job: `bench.py --gpus N` without torchrun.

Windows are sharded one rank per GPU (mc_slam_amd/shard.py); this module only starts the ranks.  The parent
process never touches the GPU (no torch.cuda / HIP call, it does not even import torch): it spawns N children
with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relays rank 0's stdout (the JSON line),
forwards the children's stderr, and exits non-zero when any child fails.  No process is ever re-exec'd.
"""
import os
import socket
import subprocess
import sys
import threading
import time


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def cpu_share(local_rank: int, local_world: int, cpus=None):
    """The host cores of one rank: the cores this process may run on, cut into `local_world` contiguous shares (contiguous core
    ids sit on one socket / NUMA node on the usual enumerations).  Every rank gets at least one core; with fewer cores than ranks
    the shares overlap round-robin."""
    if cpus is None:
        cpus = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else range(os.cpu_count() or 1)
    cpus = sorted(cpus)
    n = len(cpus)
    if local_world <= 1 or n == 0:
        return cpus
    if n < local_world:
        return [cpus[local_rank % n]]
    per = n // local_world
    return cpus[local_rank * per:(local_rank + 1) * per]


def host_threads_for(n_cores: int) -> int:
    """threads of the backend's host pool (packing, structure build, scatter) for a rank that owns n_cores: as many as it has
    cores, at most 16, at least 2 (mirrors host_threads() in csrc/vislam_ba.hip, which applies the same rule when the variable
    is absent)"""
    return max(2, min(16, n_cores))


def rank_env(base_env, rank: int, world: int, port: int, cpus=None):
    """environment of child `rank`: what torch.distributed.run would set on one node, plus this rank's share of the host:
    VBA_RANK_CPUS (the cores pin_rank() binds it to) and VBA_UPLOAD_THREADS (the backend's host pool), so that N ranks never
    run more host threads than the node has cores"""
    env = dict(base_env)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this pool
    share = cpu_share(rank, world, cpus)
    if world > 1 and share:
        env.setdefault("VBA_RANK_CPUS", ",".join(str(c) for c in share))
        env.setdefault("VBA_UPLOAD_THREADS", str(host_threads_for(len(share))))
    return env


def pin_rank(env=None):
    """Bind this process to its rank's share of the host cores -- call it in a rank BEFORE anything touches the GPU (threads the
    runtime starts later inherit the mask).  The share comes from VBA_RANK_CPUS (our launcher) or is derived from LOCAL_RANK /
    LOCAL_WORLD_SIZE (torchrun).  A single rank is left alone.  Returns the list of cores the process runs on afterwards; also
    exports VBA_UPLOAD_THREADS for the backend when the launcher did not."""
    env = os.environ if env is None else env
    world = int(env.get("LOCAL_WORLD_SIZE", env.get("WORLD_SIZE", "1")) or 1)
    if not hasattr(os, "sched_setaffinity"):
        return list(range(os.cpu_count() or 1))
    if world > 1 and env.get("VBA_NO_PIN") is None:
        if env.get("VBA_RANK_CPUS"):
            share = [int(c) for c in env["VBA_RANK_CPUS"].split(",") if c.strip() != ""]
        else:
            share = cpu_share(int(env.get("LOCAL_RANK", "0") or 0), world)
        allowed = os.sched_getaffinity(0)
        share = [c for c in share if c in allowed]
        if share:
            try:
                os.sched_setaffinity(0, share)
                env.setdefault("VBA_RANK_CPUS", ",".join(str(c) for c in share))   # the marker the library reads: this rank IS pinned to its share
            except OSError:
                pass
        if "VBA_UPLOAD_THREADS" not in env:
            env["VBA_UPLOAD_THREADS"] = str(host_threads_for(len(os.sched_getaffinity(0))))
    return sorted(os.sched_getaffinity(0))


def needs_self_launch(n_gpus: int, env) -> bool:
    """True when this process was started plainly (`python bench.py --gpus N`, N > 1) and must start the ranks itself;
    under torchrun (WORLD_SIZE set) every process already is a rank."""
    return n_gpus > 1 and "WORLD_SIZE" not in env


def spawn_ranks(n: int, argv, env=None, out=None, err=None, poll_s: float = 0.05) -> int:
    """Start `n` ranks of `argv` (a full command line), wait for all of them, relay rank 0's stdout to `out` and every
    rank's stderr to `err`.  Returns 0 when every rank exited 0, otherwise the first non-zero exit code; as soon as one
    rank fails the others are terminated (they would otherwise wait in a collective for ever)."""
    env = os.environ if env is None else env
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    port = free_port()
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen(list(argv), env=rank_env(env, r, n, port), stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=subprocess.PIPE, text=True))
    lock = threading.Lock()

    def pump(stream, sink, prefix, json_only=False):
        for line in stream:
            with lock:
                if json_only and not line.lstrip().startswith("{"):   # library chatter on stdout (e.g. "[Gloo] Rank 0 is connected ...")
                    err.write("[rank 0] " + line)
                    err.flush()
                    continue
                sink.write(prefix + line)
                sink.flush()

    pumps = [threading.Thread(target=pump, args=(procs[0].stdout, out, "", True), daemon=True)]
    pumps += [threading.Thread(target=pump, args=(p.stderr, err, "[rank %d] " % r), daemon=True) for r, p in enumerate(procs)]
    for t in pumps:
        t.start()
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:          # exactly the children started above, by PID
                    procs[q].terminate()
        if alive:
            time.sleep(poll_s)
    for t in pumps:
        t.join(timeout=5)
    return rc

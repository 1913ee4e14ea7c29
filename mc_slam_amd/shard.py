"""Sharding of independent local-BA windows over ranks (one process per GPU) and the throughput gather.

Windows are independent units (one function-local optimiser per call in the reference, src/Optimizer.cpp:130),
so there is no data-path collective: window w belongs to rank w % world, and the only communication is the
barrier around the timed region plus a MAX-reduce of the elapsed time (RCCL on GPUs, gloo in the CPU tests).
"""
import time


def window_ids(n_total: int, rank: int, world: int):
    """global ids of the windows rank `rank` owns (round-robin: w -> rank w % world)."""
    return list(range(rank, n_total, world))


def window_seed(global_id: int, base: int = 100) -> int:
    """sharded-run seeds start at 100; one distinct seed per global window id."""
    return base + global_id


class ThroughputMeter:
    """barrier -> timed region -> barrier, elapsed = MAX over ranks, value = all windows / elapsed."""

    def __init__(self, dist=None, device_sync=None):
        self.dist = dist if (dist is not None and dist.is_initialized()) else None
        self.sync = device_sync or (lambda: None)
        self.t0 = None

    def _barrier(self):
        self.sync()
        if self.dist is not None:
            self.dist.barrier()
        self.sync()

    def start(self):
        self._barrier()
        self.t0 = time.perf_counter()

    def stop(self, windows_this_rank: int, device=None):
        """returns (total_windows, elapsed_max_s) identical on every rank"""
        self._barrier()
        dt = time.perf_counter() - self.t0
        if self.dist is None:
            return windows_this_rank, dt
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device=device or "cpu")
        n = torch.tensor([float(windows_this_rank)], dtype=torch.float64, device=device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        self.dist.all_reduce(n, op=self.dist.ReduceOp.SUM)
        return int(round(n.item())), float(t.item())

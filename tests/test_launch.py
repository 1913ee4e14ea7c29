"""`bench.py --gpus N` without torchrun: the launcher starts N ranks, relays rank 0's line, propagates failures.
CPU only: the children are tiny Python scripts (no GPU, no torch), the parent never imports torch."""
import io
import json
import os
import subprocess
import sys

from mc_slam_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD_OK = r"""
import json, os, sys
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
sys.stderr.write("hello from %d\n" % r)
print("[Gloo] chatter of a library on stdout")
print(json.dumps({"rank": r, "n_gpus": w, "argv": sys.argv[1:]}))
"""
CHILD_FAIL = r"""
import os, sys, time
if os.environ["RANK"] == "1":
    sys.exit(7)
time.sleep(30)      # a rank waiting in a collective for the dead one: the launcher must end it
"""


def test_needs_self_launch_only_without_torchrun():
    assert launch.needs_self_launch(2, {})
    assert not launch.needs_self_launch(1, {})
    assert not launch.needs_self_launch(8, {"WORLD_SIZE": "8"})


def test_rank_env():
    e = launch.rank_env({"X": "1"}, 3, 8, 1234)
    assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_PORT"], e["X"]) == ("3", "3", "8", "1234", "1")
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_ranks_share_the_host_cores_and_never_oversubscribe_them():
    """8 ranks on a 64-core node: disjoint contiguous shares of 8 cores, a host pool of 8 threads each (the backend's default of 16
    per handle would put 128 packing threads on 64 cores); one rank keeps everything and gets no pinning variables"""
    cores = list(range(64))
    envs = [launch.rank_env({}, r, 8, 1, cpus=cores) for r in range(8)]
    shares = [[int(c) for c in e["VBA_RANK_CPUS"].split(",")] for e in envs]
    assert sorted(sum(shares, [])) == cores and all(len(s) == 8 and s == list(range(s[0], s[0] + 8)) for s in shares)
    assert sum(int(e["VBA_UPLOAD_THREADS"]) for e in envs) <= len(cores)
    assert all(e["LOCAL_WORLD_SIZE"] == "8" for e in envs)
    e1 = launch.rank_env({}, 0, 1, 1, cpus=cores)
    assert "VBA_RANK_CPUS" not in e1 and "VBA_UPLOAD_THREADS" not in e1
    # the caller's own settings win
    assert launch.rank_env({"VBA_UPLOAD_THREADS": "3"}, 1, 2, 1, cpus=cores)["VBA_UPLOAD_THREADS"] == "3"
    # fewer cores than ranks: every rank still gets a core, at least two threads
    tiny = [launch.rank_env({}, r, 8, 1, cpus=[0, 1, 2]) for r in range(8)]
    assert all(len(e["VBA_RANK_CPUS"].split(",")) == 1 and e["VBA_UPLOAD_THREADS"] == "2" for e in tiny)
    # a node with 192 cores: the pool stops at 16 threads per rank
    assert launch.rank_env({}, 0, 8, 1, cpus=range(192))["VBA_UPLOAD_THREADS"] == "16"


CHILD_PIN = r"""
import json, os, sys
sys.path.insert(0, os.environ["REPO_ROOT"])
from mc_slam_amd import launch
cpus = launch.pin_rank(os.environ)
import ctypes
lib = ctypes.CDLL(os.path.join(os.environ["REPO_ROOT"], "mc_slam_amd", "csrc", "libvislam_ba.so"))
sys.stderr.write(json.dumps({"rank": int(os.environ["RANK"]), "cpus": cpus, "affinity": sorted(os.sched_getaffinity(0)),
                             "threads_env": os.environ.get("VBA_UPLOAD_THREADS"), "threads_lib": lib.vba_host_threads()}) + "\n")
print(json.dumps({"ok": 1}))
"""


def test_spawned_ranks_pin_themselves_and_the_library_sizes_its_pool(tmp_path):
    """end to end on CPU: 2 ranks started by the launcher bind themselves to disjoint halves of the allowed cores before anything
    else runs, and libvislam_ba.so (no GPU call: the hook only reads the environment) reports the pool the launcher asked for"""
    f = tmp_path / "child.py"
    f.write_text(CHILD_PIN)
    env = dict(os.environ)
    env["REPO_ROOT"] = ROOT
    env.pop("VBA_UPLOAD_THREADS", None); env.pop("VBA_RANK_CPUS", None)
    out, err = io.StringIO(), io.StringIO()
    assert launch.spawn_ranks(2, [sys.executable, str(f)], env=env, out=out, err=err) == 0
    rows = sorted((json.loads(l.split("] ", 1)[1]) for l in err.getvalue().splitlines() if l.startswith("[rank") and "{" in l), key=lambda r: r["rank"])
    assert len(rows) == 2
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) >= 2:
        assert not set(rows[0]["affinity"]) & set(rows[1]["affinity"])
        assert sorted(rows[0]["affinity"] + rows[1]["affinity"]) == allowed[:2 * (len(allowed) // 2)]
    for r in rows:
        assert r["cpus"] == r["affinity"] and int(r["threads_env"]) == r["threads_lib"] == max(2, min(16, len(r["affinity"])))


def test_library_pool_follows_local_world_size_without_the_launcher():
    """under torchrun nobody exports VBA_UPLOAD_THREADS: the library divides the cores it may run on by LOCAL_WORLD_SIZE itself"""
    code = ("import ctypes, os; l = ctypes.CDLL(os.path.join(%r, 'mc_slam_amd', 'csrc', 'libvislam_ba.so')); print(l.vba_host_threads())" % ROOT)
    n = len(os.sched_getaffinity(0))
    for world, want in ((1, max(min(2, n), min(16, n))), (4, max(min(2, n), min(16, max(1, n // 4))))):
        env = dict(os.environ)
        env.pop("VBA_UPLOAD_THREADS", None)
        env["LOCAL_WORLD_SIZE"] = str(world)
        got = int(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120).stdout.strip())
        assert got == want, (world, got, want)


def test_library_pool_in_a_restricted_cpuset_still_divides_by_the_local_world():
    """a cpuset-limited container shows fewer cores than the machine has; without the launcher's marker (VBA_RANK_CPUS) the ranks
    still SHARE what it shows: 4 ranks on a mask of m cores take m / 4 threads each, not m"""
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 4:
        pytest.skip("needs at least four cores")
    mask = allowed[:max(4, len(allowed) // 2)]
    code = ("import ctypes, os; os.sched_setaffinity(0, %r); l = ctypes.CDLL(os.path.join(%r, 'mc_slam_amd', 'csrc', 'libvislam_ba.so')); print(l.vba_host_threads())"
            % (mask, ROOT))
    m = len(mask)
    for extra, want in (({}, max(min(2, m), min(16, max(1, m // 4)))),
                        ({"VBA_RANK_CPUS": ",".join(map(str, mask))}, max(min(2, m), min(16, m)))):   # pinned by the launcher: the whole share
        env = dict(os.environ)
        env.pop("VBA_UPLOAD_THREADS", None); env.pop("VBA_RANK_CPUS", None); env.pop("WORLD_SIZE", None)
        env["LOCAL_WORLD_SIZE"] = "4"
        env.update(extra)
        got = int(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120).stdout.strip())
        assert got == want, (extra, got, want)


def test_spawn_relays_rank0_and_forwards_stderr(tmp_path):
    f = tmp_path / "child.py"
    f.write_text(CHILD_OK)
    out, err = io.StringIO(), io.StringIO()
    rc = launch.spawn_ranks(3, [sys.executable, str(f), "--steps", "2"], env=dict(os.environ), out=out, err=err)
    assert rc == 0
    lines = [l for l in out.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1                       # rank 0's JSON line only
    j = json.loads(lines[0])
    assert j == {"rank": 0, "n_gpus": 3, "argv": ["--steps", "2"]}
    for r in range(3):
        assert "[rank %d] hello from %d" % (r, r) in err.getvalue()
    assert "[rank 0] [Gloo] chatter" in err.getvalue()     # non-JSON stdout of rank 0 goes to stderr, not into the result line


def test_failed_rank_gives_nonzero_exit_and_ends_the_others(tmp_path):
    import time
    f = tmp_path / "child.py"
    f.write_text(CHILD_FAIL)
    t0 = time.time()
    rc = launch.spawn_ranks(2, [sys.executable, str(f)], env=dict(os.environ), out=io.StringIO(), err=io.StringIO())
    assert rc == 7
    assert time.time() - t0 < 20


def test_bench_parent_does_not_import_torch_before_launching(tmp_path):
    """`python bench.py --gpus 2` in a process without WORLD_SIZE goes to the launcher before torch is imported: run it
    with a `torch` that explodes on import in the PARENT only (children get WORLD_SIZE and are replaced by a stub)."""
    stub = tmp_path / "stub"
    stub.mkdir()
    (stub / "torch.py").write_text("raise RuntimeError('the launcher parent must not import torch')\n")
    # the children of this test: the same bench.py, but WORLD_SIZE is set for them, so they pass the launcher, try to
    # import torch, and fail -- which the parent must report as a non-zero exit, without ever importing torch itself
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env["PYTHONPATH"] = str(stub) + os.pathsep + env.get("PYTHONPATH", "")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "pose", "--batch", "4"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
    assert "[rank 0]" in p.stderr and "[rank 1]" in p.stderr          # both children were started ...
    assert "must not import torch" in p.stderr                        # ... and they, not the parent, hit the stub
    assert "Traceback" not in p.stderr.split("[rank")[0]              # nothing blew up in the parent itself

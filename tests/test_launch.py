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

"""
bench.py starts its own ranks: ``python bench.py --gpus N`` outside a launcher spawns
``python -m torch.distributed.run`` as a CHILD process, lets rank 0's JSON line through and returns the
child's exit code (the parent never touches the GPU and never replaces itself).  Checked here with a
stand-in rank script on the gloo backend -- bench.py's own ranks need a GPU.
"""
import json
import os
import subprocess
import sys

from conftest import ROOT

_RANK_SCRIPT = """
import json, os, sys
import torch, torch.distributed as dist
dist.init_process_group("gloo")
t = torch.tensor([float(dist.get_rank() + 1)])
dist.all_reduce(t)
if dist.get_rank() == 0:
    print(json.dumps({"n_gpus": dist.get_world_size(), "sum": t.item(), "argv": sys.argv[1:],
                      "ipc": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}), flush=True)
dist.destroy_process_group()
sys.exit(int(os.environ.get("STUB_EXIT", "0")))
"""

_PARENT = """
import sys
sys.path.insert(0, %r)
import bench
assert "torch" not in sys.modules, "the parent must not import torch before it spawns the ranks"
rc = bench.launch_ranks(2, ["--gpus", "2", "--steps", "3"], script=%r)
assert "torch" not in sys.modules
sys.exit(rc)
"""


def _run(tmp_path, stub_exit):
    script = tmp_path / "rank_stub.py"
    script.write_text(_RANK_SCRIPT)
    env = dict(os.environ, STUB_EXIT=str(stub_exit))
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, "-c", _PARENT % (ROOT, str(script))], env=env, capture_output=True,
                          text=True, timeout=300)


def test_parent_spawns_the_ranks_and_relays_the_line(tmp_path):
    done = _run(tmp_path, 0)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["sum"] == 3.0 and line["argv"] == ["--gpus", "2", "--steps", "3"]
    assert line["ipc"] == "0"


def test_parent_returns_the_ranks_exit_code(tmp_path):
    assert _run(tmp_path, 7).returncode != 0


def test_gpus_flag_without_launcher_goes_through_launch_ranks(monkeypatch):
    """main() hands over to launch_ranks before anything imports torch.cuda; a launcher's world size
    that contradicts --gpus is refused."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}
    monkeypatch.setattr(bench, "launch_ranks", lambda gpus, argv, script=None: seen.update(gpus=gpus, argv=argv) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert seen == {"gpus": 4, "argv": ["--gpus", "4", "--steps", "2"]}
    monkeypatch.setenv("WORLD_SIZE", "2")
    try:
        bench.main()
        raise AssertionError("a world size that contradicts --gpus must be refused")
    except SystemExit as e:
        assert "--gpus 4" in str(e.code)


_GUARDED = """
import sys, threading
sys.path.insert(0, %r)
import bench
guard = bench.ExtrasGuard(0, %s)
guard.arm(lambda why: print("LINE " + why, flush=True))
if %s:
    threading.Event().wait()          # an extra that never returns (a collective whose peers are gone)
assert guard.disarm()
print("LINE in time", flush=True)
"""


def test_a_stalled_extra_still_yields_the_line_and_exit_code_zero():
    """With N > 1 the gather and the strong-scaling configurations run under a deadline: when one stalls, the line
    goes out with the reason and the process ends with exit code 0; in time, the deadline is never heard of."""
    stalled = subprocess.run([sys.executable, "-c", _GUARDED % (ROOT, "0.5", "True")], capture_output=True, text=True, timeout=120)
    assert stalled.returncode == 0, stalled.stderr[-2000:]
    assert stalled.stdout.count("LINE") == 1 and "timed out after" in stalled.stdout
    fine = subprocess.run([sys.executable, "-c", _GUARDED % (ROOT, "30", "False")], capture_output=True, text=True, timeout=120)
    assert fine.returncode == 0 and fine.stdout.strip() == "LINE in time"

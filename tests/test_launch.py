"""motion_planning_amd/launch.py: `bench.py --gpus N` (and tools/run_config4.py / run_config5.py) start their own ranks.
CPU tier: the command that would be run, the no-GPU refusal, and a real two-rank gloo child started through it."""
import json
import os
import subprocess
import sys
from pathlib import Path

from motion_planning_amd import launch

ROOT = Path(__file__).resolve().parents[1]


def test_launcher_command_is_the_drivers():
    cmd = launch.launcher_command("bench.py", ["--gpus", "4", "--steps", "3"], 4, port=29511)
    assert cmd[1:] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
                       "--master-port", "29511", "bench.py", "--gpus", "4", "--steps", "3"]


def test_under_launcher_reads_the_rendezvous_variables(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    assert not launch.under_launcher()
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    assert launch.under_launcher()


def _clean_env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MP_BENCH_ONE_DEVICE"):
        env.pop(k, None)
    return env


def test_too_few_gpus_is_an_error_not_a_one_rank_run():
    """No GPU in this container: every --gpus N > 1 command must exit 2 with a message and print no JSON line."""
    for script in ("bench.py", "tools/run_config4.py", "tools/run_config5.py"):
        r = subprocess.run([sys.executable, str(ROOT / script), "--gpus", "2"], capture_output=True, text=True, timeout=120,
                           env=_clean_env(), cwd=str(ROOT))
        assert r.returncode == 2, (script, r.stderr[-500:])
        assert "--gpus 2" in r.stderr and "{" not in r.stdout


def test_mismatched_world_size_is_refused():
    env = _clean_env()
    env.update(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120,
                       env=env, cwd=str(ROOT))
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr


def test_self_launch_runs_two_real_ranks(tmp_path):
    """self_launch(one_device=True) starts a two-rank gloo job whose ranks see WORLD_SIZE = 2 and can talk."""
    child = tmp_path / "child.py"
    child.write_text(
        "import json, os, sys\n"
        "import torch, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "t = torch.tensor([float(os.environ['RANK']) + 1.0])\n"
        "dist.all_reduce(t)\n"
        "if dist.get_rank() == 0:\n"
        "    print(json.dumps({'world': dist.get_world_size(), 'sum': float(t.item()), 'argv': sys.argv[1:]}))\n"
        "dist.destroy_process_group()\n")
    drv = tmp_path / "drv.py"
    drv.write_text(
        f"import sys\nsys.path.insert(0, {str(ROOT)!r})\n"
        "from motion_planning_amd import launch\n"
        f"sys.exit(launch.self_launch({str(child)!r}, ['--flag', '7'], 2, one_device=True))\n")
    r = subprocess.run([sys.executable, str(drv)], capture_output=True, text=True, timeout=170, env=_clean_env())
    assert r.returncode == 0, r.stderr[-1500:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line == {"world": 2, "sum": 3.0, "argv": ["--flag", "7"]}

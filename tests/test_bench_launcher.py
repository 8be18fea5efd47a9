"""`python bench.py --gpus N` starts its own N ranks: the launcher's command line, its refusal when the node has
fewer GPUs than ranks, and -- on this GPU-less container -- that the child ranks really start (each rank then stops at bench.py's own
"needs a GPU" assertion, whose exit code the parent hands back)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launcher_command_passes_every_argument_through():
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5", "--no-secondary", "--pipeline-depth", "1"]
    cmd = bench.launcher_command(8, argv, port=29544)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29544"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv  # the bench's own arguments, unchanged and in order, after the script
    a = bench.parse_args(cmd[i + 1:])
    assert (a.gpus, a.steps, a.warmup, a.no_secondary, a.pipeline_depth) == (8, 20, 5, True, 1)
    p1 = bench.launcher_command(2, [])
    p2 = bench.launcher_command(2, [])
    assert int(p1[p1.index("--master-port") + 1]) > 0 and int(p2[p2.index("--master-port") + 1]) > 0


def test_refuses_more_rccl_ranks_than_gpus(capsys):
    import torch
    a = bench.parse_args(["--gpus", str(torch.cuda.device_count() + 1)])
    if a.gpus < 2:
        a.gpus = 2
    assert bench.spawn_ranks(a, ["--gpus", str(a.gpus)]) == 2
    assert "needs" in capsys.readouterr().err


def test_bare_invocation_spawns_ranks_and_returns_their_exit_code():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-only check (on a GPU box the ranks would run the bench)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-secondary"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0                      # no GPU here: every rank stops at the assertion ...
    assert "bench.py needs a GPU" in r.stderr    # ... which proves the ranks were started with RANK / WORLD_SIZE set
    assert "WORLD_SIZE=" not in r.stderr          # (and not at the world-size check)

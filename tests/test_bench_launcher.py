"""bench.py --gpus N started as ONE process launches its own ranks (torch.distributed.run, one child process per GPU) before any GPU
call and relays their one JSON line; rehearsed here on the CPU with the gloo backend (--dry-run-cpu: no CAVIaR work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launcher_argv_is_the_drivers_command_line():
    import bench
    argv = bench.launcher_argv(4, 29511, ["--gpus", "4", "--steps", "7", "--warmup", "1"])
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and argv[argv.index("--nproc-per-node") + 1] == "4"
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29511"
    i = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "1"]


def test_two_rank_dry_run_on_cpu():
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0", "--dry-run-cpu"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["steps"] == 3 and out["dry_run"] is True


def test_rotation_is_a_latin_square():
    import bench
    for world in (1, 2, 8):
        for step in range(world):
            assert sorted(bench.circuit_for(r, step, world) for r in range(world)) == list(range(world))
        for r in range(world):
            assert sorted(bench.circuit_for(r, s, world) for s in range(world)) == list(range(world))

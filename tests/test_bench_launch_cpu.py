"""CPU: bench.py's argument contract around --gpus (no GPU touched): a launcher whose WORLD_SIZE disagrees with --gpus is an error
(round 2 parsed the flag and silently ran one rank), and --base-file / --query-file go together."""
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, env):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = _run(["--gpus", "4"], env)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)
    out = _run(["--gpus", "0"], env)
    assert out.returncode != 0

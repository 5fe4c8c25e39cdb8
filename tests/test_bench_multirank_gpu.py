"""The N > 1 path of bench.py, rehearsed on the one-GPU test box: two ranks launched exactly as the driver launches them
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`), both on cuda:0 with gloo standing
in for RCCL (which wants one GPU per rank).  Checks the contract of the JSON line rank 0 prints."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_prints_one_aggregate_line():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, CVCS_BENCH_ONE_DEVICE="1", CVCS_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--tile", "64"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "tiles/s"
    assert d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["value"] == pytest.approx(2 * 2 * 2 / (d["ms_per_step"] * 2 / 1e3), rel=1e-3)   # whole-job tiles / max-over-ranks time
    assert d["roofline"]["bound"] in ("mfma", "hbm") and d["roofline"]["achieved"] > 0      # the family with the largest time, whatever bounds it
    assert "cpu_baseline" not in d          # rank 0 at N = 1 only
    assert d["loss"] == d["loss"]           # finite

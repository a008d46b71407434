"""Two ranks sharing cuda:0 (gloo carries the collectives; RCCL refuses two ranks on one
device): the sharded pipeline end to end on real kernels, checked against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_one_gpu():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533",
                        os.path.join(ROOT, "scripts", "two_rank_one_gpu.py")], capture_output=True, text=True, env=env,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("match_ok=True ransac_ok=True") == 2

"""The N>1 logic of bench.py (sharding, per-iteration all-reduce through the callback, max-over-
ranks timing, one JSON line on rank 0) exercised with TWO processes on the one GPU of the test box
(gloo backend staging the fused buffer through the host).  The real 8-GPU run uses RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_two_ranks_one_gpu_match_single_rank():
    common = ["--nseq", "20000", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--dist-backend", "gloo"] + common, capture_output=True, text=True, cwd=ROOT, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [l for l in two.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                    # rank 0 only
    j2 = json.loads(lines[0])
    assert j2["n_gpus"] == 2 and j2["steps"] == 6 and j2["scaling"] == "strong"
    # same total work, same model trajectory: the log-likelihood after the last pass agrees
    assert j2["llh_last"] == pytest.approx(j1["llh_last"], rel=1e-6)
    assert j2["config"]["n_seqs"] == j1["config"]["n_seqs"]

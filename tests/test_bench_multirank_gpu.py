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


def _run_bench(*extra):
    common = ["--nseq", "20000", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-extras"]
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + list(extra), capture_output=True, text=True, cwd=ROOT)


@pytest.mark.timeout(600)
def test_inprocess_ranks_without_a_launcher():
    """`python bench.py --gpus N` with no launcher runs N ranks inside the process (one context + host thread each,
    the library's communicator).  On this 1-GPU box: (a) asking for 2 GPUs FAILS LOUDLY instead of printing a
    1-GPU line, (b) the same code path with --local-ranks (2 contexts on device 0, host-staged sum) reproduces the
    single-rank trajectory, (c) a 1-rank RCCL communicator through ncclCommInitAll does too."""
    refused = _run_bench("--gpus", "2")
    assert refused.returncode != 0
    assert not [l for l in refused.stdout.splitlines() if l.startswith("{")]
    assert "only 1 HIP device" in refused.stderr

    one = _run_bench()
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert j1["n_gpus"] == 1 and j1["metric"].endswith("20kx200bp k=2 W=20")

    two = _run_bench("--gpus", "2", "--local-ranks")
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [l for l in two.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j2 = json.loads(lines[0])
    assert j2["n_gpus"] == 2 and [r["rank"] for r in j2["ranks"]] == [0, 1] and all(r["world"] == 2 for r in j2["ranks"])
    assert sum(r["n_seqs"] for r in j2["ranks"]) == 20000
    assert j2["ranks_agree_bitwise"] is True
    assert j2["llh_last"] == j1["llh_last"]                  # integer sums: the same bits whatever the number of ranks

    rccl1 = _run_bench("--gpus", "1", "--local-ranks", "--force-dist")      # N = 1 through the in-process path, host-staged
    assert rccl1.returncode == 0, rccl1.stderr[-3000:]
    j3 = json.loads([l for l in rccl1.stdout.splitlines() if l.startswith("{")][-1])
    assert j3["llh_last"] == j1["llh_last"]


@pytest.mark.timeout(600)
def test_two_processes_under_the_launcher_with_the_library_communicator():
    """The driver's form -- `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` -- rehearsed with two
    PROCESSES on the one GPU: the library's own communicator on the kernels' stream (here the shared-memory one, since RCCL
    refuses two ranks on a device), `attribution`, and the in-kernel all-reduce extra through hipIpc-mapped inboxes with both
    ranks ending on the same model."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29619", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--dist-backend", "gloo", "--shm-comm", "--nseq", "100000", "--steps", "12", "--warmup", "4",
                          "--blocks", "112", "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [l for l in two.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and "bamm_comm_init_shm" in j["allreduce"]
    assert [(r["rank"], r["world"]) for r in j["ranks"]] == [(0, 2), (1, 2)]
    a = j["attribution"]
    assert a["kernel_us"] > 0 and a["allreduce_us"] > 0 and a["step_us"] >= a["kernel_us"]
    assert isinstance(j["ms_per_step_peer_allreduce"], float), j["ms_per_step_peer_allreduce"]
    assert len(j["peer_allreduce"]["model_sha"]) == 16
    # two identical timed regions, one per way of summing over the ranks: the line's value comes from the faster one (both sound)
    assert isinstance(j["ms_per_step_rccl"], float) and j["headline_collective"] in ("rccl", "peer")
    assert j["ms_per_step"] == pytest.approx(min(j["ms_per_step_rccl"], j["ms_per_step_peer_allreduce"]), rel=1e-9)
    assert ("in-kernel all-reduce" in j["allreduce"]) == (j["headline_collective"] == "peer")
    # the first-contact kit: the communicator self-test ran in front of the timed region (both ways of summing over the
    # ranks end on one model), and the line says where the ranks' devices sit and who reaches whom
    st = j["selftest_comm"]
    assert st["rccl"]["ok"] and st["peer"]["ok"] and st["rccl"]["model_sha"] == st["peer"]["model_sha"], st
    topo = j["topology"]
    assert topo["visible_devices"] >= 1 and len(topo["pci_bus_id"]) == topo["visible_devices"]
    assert len(topo["can_access_peer"]) == topo["visible_devices"] and topo["can_access_peer"][0][0] is True
    assert all(r["pci_bus_id"] == topo["pci_bus_id"][r["device"]] and ":" in r["pci_bus_id"] for r in j["ranks"])


@pytest.mark.timeout(600)
def test_first_call_figures_on_one_gpu():
    """from_seed.first_create_ms / first_call_ms: the first handle of the process, nothing warmed up."""
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--nseq", "50000", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    j = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    fs = j["from_seed"]
    assert fs["first_call_passes"] == 20 and 0 < fs["first_call_ms"] < 200 and 0 < fs["first_create_ms"] < 500, fs
    assert fs["ms_per_step_iterate"] > 0 and fs["ms_per_step_optimize"] > 0

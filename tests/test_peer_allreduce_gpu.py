"""The pass's all-reduce INSIDE the sequence kernel's tail (include/bamm_em.h: bamm_em_comm_mode, tuning "peer_allreduce";
csrc/update_kernel.h: peer_allreduce_tail; csrc/comm.cpp: comm_peer_setup) -- the reductions of
/root/reference/src/refinement/EM.cpp:148,240,509-513 without a collective launch between two passes.

A 1-GPU box cannot put the inboxes on different devices; what it can check is everything else: N contexts on device 0
(a host thread, a stream and a kernel per rank, resident side by side), the set-up vote, the flags and sequence numbers,
the last block's exchange of self-validating entries, the bounded polls, and that the model is the single-rank model bit for bit.  Across processes the
inboxes travel as hipIpc handles (second test)."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import bammmotif2_amd as bm
from tests.cases import SMALL_CASES, Case
from tests.test_parity_gpu import make_em

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# shapes whose pass is ONE launch of a kernel built with the tail: the mixed-row kernel (layout 8: the bench / config 2 / 3 / 5
# shapes), and k_em_grp's uniform rows -- the same shape with the layout forced, k = 1, and a single-stranded set
SHAPES = {
    "mix_k2_ds": (dict(SMALL_CASES[6]), 8),
    "grp_k2_ds_uniform_rows": (dict(SMALL_CASES[6]), 3),
    "grp_k1_ds": (dict(name="peer_k1", N=200, L0=200, W=12, K=1, seed=31), -1),
    "grp_k2_ss": (dict(name="peer_ss", N=220, L0=200, W=20, K=2, ss=True, seed=32), -1),
}


def _local_ranks(n, c, orc, calls, peer=True, timeout_ms=None, blocks=None, layout=8):
    """n contexts on device 0, the sequences sharded over them, the host-staged communicator for whatever still goes
    through a collective (the set-up votes), peer_allreduce on every context."""
    seq, kmer, off, vbg = c.encode(orc)
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    ctxs = [bm.Context(0) for _ in range(n)]
    for x in ctxs:
        x.set_tuning(peer_allreduce=int(peer), group_layout=layout)   # 8 = mixed rows: the planner takes them by itself only
                                                                      # for tens of thousands of sequences
        if timeout_ms:
            x.set_tuning(peer_timeout_ms=timeout_ms)
        x.set_launch(blocks or max(1, 224 // n), 0)           # the ranks' kernels are resident side by side: they wait for each other
    comms = bm.Comm.init_local(ctxs, 4 ** (c.K + 1) * c.W + 3)
    sets, ems = [], []
    for r in range(n):
        b, e = pk.shard_range(c.W, r, n)
        ss = bm.SeqSet(ctxs[r], pk, b, e)
        em = bm.EM(ctxs[r], ss, c.K, c.W, vbg, c.A, c.v0, c.q, bg_order=c.bg_order, optimizeQ=True, n_seqs_global=c.N,
                   n_seqs_bound=c.N, max_iterations=60)
        em.set_comm(comms[r])
        sets.append(ss); ems.append(em)
    out, errs = [None] * n, [None] * n

    def worker(r):
        try:
            out[r] = calls(ems[r], r)
        except Exception as e:
            errs[r] = e
            for x in comms:
                x.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=180)
    assert not any(t.is_alive() for t in th)
    for x in ems + sets + comms + ctxs:
        x.close()
    return out, errs


@pytest.mark.parametrize("n", [2, 3])
@pytest.mark.parametrize("shape", list(SHAPES), ids=list(SHAPES))
def test_ranks_on_one_device_equal_one_rank_bit_for_bit(shape, n, gpu_ctx, orc):
    spec, layout = SHAPES[shape]
    c = Case(**spec)
    gpu_ctx.set_tuning(group_layout=layout)
    try:
        one, ss, *_ = make_em(gpu_ctx, c, orc, optimizeQ=True, max_iterations=60)
    finally:
        gpu_ctx.set_tuning(group_layout=-1)
    assert one.plan_mixed() == (c.N if layout == 8 else 0) and one.plan()[1] == 0       # one grouped launch, mixed rows or uniform ones
    one.iterate(7)
    it = one.optimize()
    want = (one.getV(), one.getQ(), one.trace()[0], it, one.getCounts())
    one.close(); ss.close()

    def calls(em, r):
        mode, note = em.comm_mode()
        em.iterate(7)                                        # every pass sums over the ranks in its own launch's tail
        k = em.optimize()                                    # look-ahead, the stop rule in the fused prologues
        return em.getV(), em.getQ(), em.trace()[0], k, em.getCounts(), mode, note

    out, errs = _local_ranks(n, c, orc, calls, layout=layout)
    assert errs == [None] * n, [str(e) for e in errs]
    for r in range(n):
        assert out[r][5] == 2, out[r][6]                     # the in-kernel mode was agreed on
        assert out[r][3] == want[3] and out[r][1] == want[1]
        assert np.array_equal(out[r][0], want[0]) and np.array_equal(out[r][2], want[2]) and np.array_equal(out[r][4], want[4])


def test_passes_without_the_tail_are_still_summed_over_the_ranks(gpu_ctx, orc):
    """Mode 2 agreed, but EStep() alone (an E-only launch carries no tail), the MStep() replay driven by hand, optimize_q()
    and EM::mask's kernels never pass through the tail: they must take the communicator's collective, not each rank's own
    shard (round-4 advice: run_allreduce returned early for every pass of a mode-2 handle)."""
    c = Case(**SMALL_CASES[6])
    seq, kmer, off, vbg = c.encode(orc)

    def drive(make):
        em = make()
        mode = em.comm_mode()[0]
        em.iterate(2)                                        # the tail (mode 2) / no collective at all (one rank)
        em.EStep()
        llh = em.getLLH()
        em.MStep()
        em.optimize_q()
        q = em.getQ()
        v = em.getV()
        em.iterate(2)
        v2, n2 = em.getV(), em.getCounts()
        em.close()
        em = make()                                          # EM::mask runs on a fresh handle (EM.cpp:261)
        mode_m = em.comm_mode()[0]
        it = em.mask(0.2)
        out = (llh, q, v, v2, n2, it, em.getV(), em.getCounts(), mode, mode_m)
        em.close()
        return out

    gpu_ctx.set_tuning(group_layout=8)
    pk1 = bm.PackedSeqs.from_kmers(kmer, off)
    ss1 = bm.SeqSet(gpu_ctx, pk1)
    try:
        want = drive(lambda: bm.EM(gpu_ctx, ss1, c.K, c.W, vbg, c.A, c.v0, c.q, bg_order=c.bg_order, optimizeQ=False, max_iterations=60))
    finally:
        gpu_ctx.set_tuning(group_layout=-1)
    ss1.close()

    n = 2
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    ctxs = [bm.Context(0) for _ in range(n)]
    for x in ctxs:
        x.set_tuning(peer_allreduce=1, group_layout=8)
        x.set_launch(max(1, 224 // n), 0)
    comms = bm.Comm.init_local(ctxs, max(4 ** (c.K + 1) * c.W + 3, 2049))      # (EM::mask's cut-off histogram: 2049 words)
    sets = []
    for r in range(n):
        b_, e_ = pk.shard_range(c.W, r, n)
        sets.append(bm.SeqSet(ctxs[r], pk, b_, e_))
    out, errs = [None] * n, [None] * n

    def worker(r):
        def make():
            em = bm.EM(ctxs[r], sets[r], c.K, c.W, vbg, c.A, c.v0, c.q, bg_order=c.bg_order, optimizeQ=False, n_seqs_global=c.N,
                       n_seqs_bound=c.N, max_iterations=60)
            em.set_comm(comms[r])
            return em
        try:
            out[r] = drive(make)
        except Exception as e:
            errs[r] = e
            for x in comms:
                x.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=180)
    assert not any(t.is_alive() for t in th)
    for x in sets + comms + ctxs:
        x.close()
    assert errs == [None, None], [str(e) for e in errs]
    for r in range(n):
        assert out[r][8] == 2 and out[r][9] == 2              # the in-kernel mode was agreed on by both handles
        assert out[r][0] == want[0] and out[r][1] == want[1] and out[r][5] == want[5]
        for i in (2, 3, 4, 6, 7):
            assert np.array_equal(out[r][i], want[i]), i


def test_without_the_tuning_or_on_an_unfit_handle_the_collective_stays(gpu_ctx, orc):
    c = Case(**SMALL_CASES[6])
    out, errs = _local_ranks(2, c, orc, lambda em, r: (em.comm_mode(), em.iterate(3), em.getV())[::2], peer=False)
    assert errs == [None, None] and out[0][0][0] == 1 and np.array_equal(out[0][1], out[1][1])
    c4 = Case("k4", N=40, L0=120, W=12, K=4, seed=5)          # sliced path: not one grouped launch with the fused update
    out, errs = _local_ranks(2, c4, orc, lambda em, r: (em.comm_mode(), em.iterate(2), em.getV())[::2])
    assert errs == [None, None] and out[0][0][0] == 1 and "not one launch of a grouped-column kernel built with the tail" in out[0][0][1]
    assert np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("shape", ["mix_k2_ds", "grp_k1_ds"])
def test_a_rank_that_arrives_too_late_is_an_error_not_a_hang(shape, gpu_ctx, orc):
    """Rank 1 starts 1.5 s late; rank 0's first pass waits 300 ms (peer_timeout_ms) for sums that do not come: the wait is
    bounded, the handle's later launches do nothing, and the next read fails with BAMM_ERR_COMM -- on rank 1 as well,
    whose second pass then waits in vain for the pass rank 0 never ran."""
    import time
    spec, layout = SHAPES[shape]
    c = Case(**spec)
    gate = threading.Barrier(2)

    def calls(em, r):
        assert em.comm_mode()[0] == 2
        gate.wait()
        if r == 1:
            time.sleep(1.5)
        em.iterate(4)
        return em.getV()

    out, errs = _local_ranks(2, c, orc, calls, timeout_ms=300, layout=layout)
    for r in range(2):
        assert isinstance(errs[r], bm.abi.BammError), (out, errs)
        assert errs[r].code == bm.abi.ERR_COMM, errs[r]
    assert "did not arrive" in str(errs[0]) or "did not arrive" in str(errs[1])


WORKER = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import bammmotif2_amd as bm
import oracle
from tests.cases import SMALL_CASES, Case
rank, world, uid_path, out_path = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
orc = oracle.Oracle(); orc.set_threads(1)
c = Case(**SMALL_CASES[6])
seq, kmer, off, vbg = c.encode(orc)
pk = bm.PackedSeqs.from_kmers(kmer, off)
ctx = bm.Context(0)
ctx.set_tuning(peer_allreduce=1, group_layout=8)
ctx.set_launch(100, 0)
if rank == 0:
    open(uid_path + ".tmp", "wb").write(bm.Comm.unique_id()); os.replace(uid_path + ".tmp", uid_path)
import time
t0 = time.time()
while not os.path.exists(uid_path):
    assert time.time() - t0 < 60
    time.sleep(0.05)
comm = bm.Comm.init_rank(ctx, open(uid_path, "rb").read(), rank, world)
b, e = pk.shard_range(c.W, rank, world)
ss = bm.SeqSet(ctx, pk, b, e)
em = bm.EM(ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q, bg_order=c.bg_order, optimizeQ=True, n_seqs_global=c.N, n_seqs_bound=c.N, max_iterations=60)
em.set_comm(comm)
mode, note = em.comm_mode()
em.iterate(7)
k = em.optimize()
np.savez(out_path, v=em.getV(), q=em.getQ(), llh=em.trace()[0], k=k, mode=mode, note=note)
em.close(); ss.close(); comm.close(); ctx.close()
"""


def test_two_processes_share_the_inboxes_through_ipc_handles(gpu_ctx, orc, tmp_path):
    """One process per rank, as under torch.distributed.run: the inboxes cross the process boundary as hipIpc handles
    carried by the communicator's own all-reduce.  Both ranks sit on device 0 here, which RCCL refuses for a communicator
    -- so this runs only where two devices are visible; on a 1-GPU box the in-process tests above stand for it."""
    if bm.device_count() < 2:
        pytest.skip("one visible device: RCCL refuses two ranks on it (the in-process tests cover the protocol)")
    c = Case(**SMALL_CASES[6])
    gpu_ctx.set_tuning(group_layout=8)
    try:
        one, ss, *_ = make_em(gpu_ctx, c, orc, optimizeQ=True, max_iterations=60)
    finally:
        gpu_ctx.set_tuning(group_layout=-1)
    one.iterate(7)
    it = one.optimize()
    want = (one.getV(), one.getQ(), one.trace()[0], it)
    one.close(); ss.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    uid = str(tmp_path / "uid.bin")
    procs = []
    for r in range(2):
        env = dict(os.environ, HIP_VISIBLE_DEVICES=str(r), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", uid, str(tmp_path / f"out{r}.npz")], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    for r in range(2):
        o = np.load(tmp_path / f"out{r}.npz")
        assert int(o["mode"]) == 2, str(o["note"])
        assert int(o["k"]) == want[3] and float(o["q"]) == want[1]
        assert np.array_equal(o["v"], want[0]) and np.array_equal(o["llh"], want[2])


WORKER_SHM = WORKER.replace('''if rank == 0:
    open(uid_path + ".tmp", "wb").write(bm.Comm.unique_id()); os.replace(uid_path + ".tmp", uid_path)
import time
t0 = time.time()
while not os.path.exists(uid_path):
    assert time.time() - t0 < 60
    time.sleep(0.05)
comm = bm.Comm.init_rank(ctx, open(uid_path, "rb").read(), rank, world)
''', '''comm = bm.Comm.init_shm(ctx, uid_path, rank, world, 4 ** (c.K + 1) * c.W + 3)      # uid_path: the segment's name here
''')
assert WORKER_SHM != WORKER


def test_two_processes_on_one_device_share_the_inboxes_through_ipc_handles(gpu_ctx, orc, tmp_path):
    """The cross-process half of the in-kernel all-reduce on a 1-GPU box: one PROCESS per rank, both on device 0, the
    set-up traffic (pids, pointers, hipIpc handles, the votes) over the shared-memory communicator (bamm_comm_init_shm; RCCL
    refuses two ranks on one device), the inboxes opened with hipIpcOpenMemHandle in the other process -- what every rank
    does under torch.distributed.run.  Mode 2 agreed, the model bit for bit the one-rank model."""
    c = Case(**SMALL_CASES[6])
    gpu_ctx.set_tuning(group_layout=8)
    try:
        one, ss, *_ = make_em(gpu_ctx, c, orc, optimizeQ=True, max_iterations=60)
    finally:
        gpu_ctx.set_tuning(group_layout=-1)
    one.iterate(7)
    it = one.optimize()
    want = (one.getV(), one.getQ(), one.trace()[0], it)
    one.close(); ss.close()
    script = tmp_path / "worker_shm.py"
    script.write_text(WORKER_SHM)
    name = f"/bamm_test_{os.getpid()}"
    # a segment of that name left behind by a creator that was killed before its shm_unlink (names are recycled with pids):
    # right size, magic set, one rank already "arrived" at the barrier -- round-4 advice: every rank attached to it, nobody
    # unlinked it, and the first rank passed the barrier alone.  Rank 0 now replaces it; an attacher that opened it first
    # gets no answer in it and opens the name again.
    words = 4 ** (c.K + 1) * c.W + 3
    stale = np.zeros((32 + 2 * 64 * 8 + 2 * words * 8) // 4, np.uint32)
    stale[0] = 0x42414d4d; stale[1] = 1; stale[5] = 2         # magic, arrived = 1, n = 2
    stale[6:8] = np.array([words], np.uint64).view(np.uint32)  # cap
    stale.tofile("/dev/shm" + name)
    procs = []
    for r in (1, 0):                                          # the attacher first: it finds the stale segment
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", name, str(tmp_path / f"shm{r}.npz")], env=env))
        if r == 1:
            import time
            time.sleep(1.0)
    for p in procs:
        assert p.wait(timeout=300) == 0
    for r in range(2):
        o = np.load(tmp_path / f"shm{r}.npz")
        assert int(o["mode"]) == 2, str(o["note"])
        assert int(o["k"]) == want[3] and float(o["q"]) == want[1]
        assert np.array_equal(o["v"], want[0]) and np.array_equal(o["llh"], want[2])
    assert not os.path.exists("/dev/shm" + name)             # no name is left behind

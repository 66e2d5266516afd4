"""RCCL inside the drop-in (include/bamm_em.h: bamm_comm_*, bamm_em_set_comm): the per-pass all-reduce of
the integer accumulator issued by libbamm_em itself on the context's stream.  A 1-GPU box can only form
1-rank communicators; what is checked is that the native path runs, changes nothing (an all-reduce over one
rank is the identity, and the accumulator holds integers) and is refused where it must be."""
import subprocess

import numpy as np
import pytest

import bammmotif2_amd as bm
from bammmotif2_amd import build
from tests.cases import SMALL_CASES, Case
from tests.test_parity_gpu import make_em

pytestmark = pytest.mark.gpu


def test_library_opens_rccl_only_on_demand():
    out = subprocess.run(["ldd", build.LIB], capture_output=True, text=True).stdout
    assert "rccl" not in out                     # dlopen at the first bamm_comm_* call, not a link-time dependency


@pytest.mark.parametrize("how", ["init_all", "init_rank"])
def test_one_rank_communicator_equals_no_communicator(how, gpu_ctx, orc):
    c = Case(**SMALL_CASES[6])
    plain, ss0, kmer, off, vbg = make_em(gpu_ctx, c, orc, optimizeQ=True)
    native, ss1, *_ = make_em(gpu_ctx, c, orc, optimizeQ=True)
    if how == "init_all":
        comm = bm.Comm.init_all([gpu_ctx])[0]
    else:
        comm = bm.Comm.init_rank(gpu_ctx, bm.Comm.unique_id(), 0, 1)
    info = comm.info()
    assert info["rank"] == 0 and info["world"] == 1 and info["rccl_version"] > 0
    native.set_comm(comm)
    plain.iterate(4); native.iterate(4)
    assert np.array_equal(plain.getV(), native.getV()) and plain.getQ() == native.getQ()
    assert np.array_equal(plain.trace()[0], native.trace()[0])
    assert plain.optimize() == native.optimize()
    assert np.array_equal(plain.getCounts(), native.getCounts())
    # EStep / MStep / getR go through the same point
    plain.EStep(); native.EStep()
    assert plain.getLLH() == native.getLLH()
    plain.MStep(); native.MStep()
    assert np.array_equal(plain.getV(), native.getV())
    native.set_comm(None)
    for x in (plain, native, ss0, ss1):
        x.close()
    comm.close()


def test_mask_histogram_goes_through_the_communicator(gpu_ctx, orc):
    """EM::mask's cut-off select all-reduces its window histogram (int64 as well)."""
    c = Case(**SMALL_CASES[0])
    a, sa, *_ = make_em(gpu_ctx, c, orc, epsilon=0.0, max_iterations=3)
    b, sb, *_ = make_em(gpu_ctx, c, orc, epsilon=0.0, max_iterations=3)
    comm = bm.Comm.init_all([gpu_ctx])[0]
    b.set_comm(comm)
    assert a.mask(0.1) == b.mask(0.1)
    assert a.last_mask == b.last_mask and np.array_equal(a.getV(), b.getV())
    for x in (a, b, sa, sb):
        x.close()
    comm.close()


def test_communicator_misuse_is_refused(gpu_ctx, orc):
    with pytest.raises(bm.abi.BammError) as e:
        bm.Comm.init_all([gpu_ctx, gpu_ctx])                  # one rank per GPU
    assert e.value.code == bm.abi.ERR_ARG
    other = bm.Context(0)
    comm = bm.Comm.init_all([other])[0]
    c = Case(**SMALL_CASES[0])
    em, ss, *_ = make_em(gpu_ctx, c, orc)
    with pytest.raises(bm.abi.BammError):
        em.set_comm(comm)                                    # communicator of another context
    em.close(); ss.close(); comm.close(); other.close()


def _two_local_ranks(c, orc, bounds, calls):
    """Two contexts on device 0, the sequences sharded over them, the host-staged communicator (bamm_comm_init_local),
    one host thread per rank: calls(em, rank) runs on each.  Returns (results, errors)."""
    import threading
    seq, kmer, off, vbg = c.encode(orc)
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    ctxs = [bm.Context(0), bm.Context(0)]
    comms = bm.Comm.init_local(ctxs, 4 ** (c.K + 1) * c.W + 3)
    assert [x.info()["rank"] for x in comms] == [0, 1] and comms[0].info()["rccl_version"] == 0
    sets, ems = [], []
    for r in range(2):
        b, e = pk.shard_range(c.W, r, 2)
        ss = bm.SeqSet(ctxs[r], pk, b, e)
        em = bm.EM(ctxs[r], ss, c.K, c.W, vbg, c.A, c.v0, c.q, bg_order=c.bg_order, optimizeQ=True, n_seqs_global=c.N,
                   n_seqs_bound=bounds[r], max_iterations=60)
        em.set_comm(comms[r])
        sets.append(ss); ems.append(em)
    out, errs = [None, None], [None, None]

    def worker(r):
        try:
            out[r] = calls(ems[r], r)
        except Exception as e:                               # a rank that fails alone must not leave its peer in the collective
            errs[r] = e
            for x in comms:
                x.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in th)
    for x in ems + sets + comms + ctxs:
        x.close()
    return out, errs


def test_two_ranks_in_one_process_equal_one_rank(gpu_ctx, orc):
    """The multi-rank host logic end to end on one GPU: shards, a handle + host thread per rank, an all-reduce per pass
    (host-staged here), the fused update reading the all-reduced ring slot, optimize()'s look-ahead on every rank --
    the model equals the single-rank run bit for bit (integer sums), on both ranks."""
    c = Case(**SMALL_CASES[6])
    one, ss, *_ = make_em(gpu_ctx, c, orc, optimizeQ=True, max_iterations=60)
    one.iterate(5)
    it = one.optimize()
    want = (one.getV(), one.getQ(), one.trace()[0], it)
    one.close(); ss.close()

    def calls(em, r):
        em.iterate(5)
        n = em.optimize()
        return em.getV(), em.getQ(), em.trace()[0], n

    out, errs = _two_local_ranks(c, orc, (c.N, c.N), calls)
    assert errs == [None, None], errs
    for r in range(2):
        assert out[r][3] == want[3] and out[r][1] == want[1]
        assert np.array_equal(out[r][0], want[0]) and np.array_equal(out[r][2], want[2])


def test_ranks_with_different_accumulator_units_fail_together(gpu_ctx, orc):
    """bamm_em_params.n_seqs_bound sizes the unit of the int64 accumulator; ranks that disagree would mis-scale each
    other's counts.  The first pass over a communicator checks it with the peers: BOTH ranks get the error, nobody hangs."""
    c = Case(**SMALL_CASES[6])
    out, errs = _two_local_ranks(c, orc, (c.N, 1 << 30), lambda em, r: em.iterate(2))
    assert all(isinstance(e, bm.abi.BammError) for e in errs), errs
    assert "accumulator units differ" in str(errs[0]) and "accumulator units differ" in str(errs[1])

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

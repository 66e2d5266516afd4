"""SeqGenerator's negative sampler on the device (csrc/negs.hip; include/bamm_em.h: bamm_sample_negatives) against the host
restatement (host/fdr.cpp::sample_negatives), which the CPU suite pins to the reference's own negatives
(tests/test_eval_cpu.py::test_negative_sampler_matches_reference; /root/reference/src/seq_generator/SeqGenerator.cpp:63-348).
Every base must be equal: the rand() stream entered by jump-ahead per negative, the float tables in the reference's order."""
import ctypes as C
import os

import numpy as np
import pytest

import bammmotif2_amd as bm
from bammmotif2_amd import build, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host():
    build.build_host()
    h = C.CDLL(build.HOST_LIB)
    h.bh_last_error.restype = C.c_char_p
    return h


def host_negatives(host, packed, m_fold, generic, stride):
    n, m = C.c_uint64(), C.c_uint64()
    assert host.bh_sample_negatives_strided(packed._p, 2, C.c_uint64(m_fold), int(generic), C.c_uint64(stride), C.byref(n), C.byref(m), None, None) == 0, host.bh_last_error()
    codes, off = np.zeros(m.value, np.uint8), np.zeros(n.value + 1, np.uint64)
    assert host.bh_sample_negatives_strided(packed._p, 2, C.c_uint64(m_fold), int(generic), C.c_uint64(stride), C.byref(n), C.byref(m),
                                            codes.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p)) == 0
    return codes, off


CASES = [dict(name="ds_const", N=300, L0=100, ss=False, ragged=0, n_frac=0.0, m=3),
         dict(name="ds_ragged_N", N=200, L0=120, ss=False, ragged=50, n_frac=0.01, m=4),
         dict(name="ss_short", N=400, L0=12, ss=True, ragged=9, n_frac=0.0, m=10),
         dict(name="ss_long", N=20, L0=3000, ss=True, ragged=900, n_frac=0.001, m=2),
         dict(name="many_per_positive", N=6, L0=80, ss=False, ragged=10, n_frac=0.0, m=150)]


@pytest.mark.parametrize("generic", [False, True], ids=["seq_specific", "generic"])
@pytest.mark.parametrize("stride", [0, 5, 4])
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_device_negatives_are_the_host_negatives(case, stride, generic, gpu_ctx, host):
    pwm = synth.make_pwm(8, 3)
    codes, off = synth.make_sequences(case["N"], case["L0"], pwm, 17, 0.5, case["n_frac"], case["ragged"])
    packed = bm.PackedSeqs.from_codes(codes, off, case["ss"], seed=42)
    pos = bm.SeqSet(gpu_ctx, packed)
    want_codes, want_off = host_negatives(host, packed, case["m"], generic, stride)
    neg, res = bm.sample_negatives(gpu_ctx, pos, 2, case["m"], generic, stride)
    assert neg.n_seqs == len(want_off) - 1
    assert np.array_equal(neg.lengths, np.diff(want_off.astype(np.int64)))
    assert neg.n_exceptions == 0
    got = (neg.unpack_y(0) + 1).astype(np.uint8)             # the 2-bit stream back as codes 1..4
    assert np.array_equal(got, want_codes)
    # the resident set scores like a set packed on the host from the same codes
    ref = bm.PackedSeqs.from_codes(want_codes, want_off, True, seed=None)
    assert np.array_equal(ref.words, neg.words)
    res.close(); pos.close()


def test_random_shapes(gpu_ctx, host):
    """Twenty-four seeded random positive sets (1..800 records, 3..600 bases, with and without N, both strand modes), random
    m-fold, stride and flavour: the device's negatives base for base the host sampler's."""
    rs = np.random.RandomState(77)
    pwm = synth.make_pwm(8, 3)
    for trial in range(24):
        N = int(rs.choice([1, 3, 40, 800]))
        L0 = int(rs.choice([3, 9, 50, 600]))
        ragged = int(rs.randint(0, max(1, L0 - 2)))
        codes, off = synth.make_sequences(N, L0, pwm, 100 + trial, 0.0, float(rs.choice([0.0, 0.02])), ragged)
        ss = bool(rs.randint(2))
        m = int(rs.choice([1, 2, 7, 33]))
        stride = int(rs.choice([0, 2, 5]))
        generic = bool(rs.randint(2))
        packed = bm.PackedSeqs.from_codes(codes, off, ss, seed=42)
        seqs = bm.SeqSet(gpu_ctx, packed)
        want_codes, want_off = host_negatives(host, packed, m, generic, stride)
        npk, nseqs = bm.sample_negatives(gpu_ctx, seqs, 2, m, generic, stride)
        got = (npk.unpack_y(0) + 1).astype(np.uint8)         # the 2-bit stream back as codes 1..4
        assert np.array_equal(npk.lengths, np.diff(want_off.astype(np.int64))), (trial, N, L0, m, stride)
        assert np.array_equal(got, want_codes), (trial, N, L0, m, stride, generic)
        nseqs.close(); seqs.close()

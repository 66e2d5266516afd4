"""Size-independent properties checked at BASELINE.json's full size (1M x 200 bp, W=20, k=2),
where the CPU oracle would take minutes per pass:

* mass conservation: every window covers column 0, so sum_y n[K][y][0] == sum_n sum_i r_n(i), and
  the latter is also accumulated independently as 1 - (1-q)/Z_n;
* linearity in the sequence set: counts of two half shards add up to the counts of the whole set
  (the accumulation is integer arithmetic, so this holds to the last bit of the fp64 buffer);
* order-0 conditionals are distributions; lower-order counts are marginals of the top order;
* a second pass from the same model reproduces the same buffer bit for bit (no scheduling noise).
"""
import ctypes as C

import numpy as np
import pytest

import bammmotif2_amd as bm
from bammmotif2_amd import synth

pytestmark = pytest.mark.gpu


def read_buffer(em, ctx):
    hip = C.CDLL("libamdhip64.so")
    ptr, n = em.reduce_buffer()
    ctx.sync()
    host = np.zeros(n, np.int64)                    # 64-bit integers: counts 2^-40, llh 2^-24, sum_r 2^-30, n_seqs 1
    assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), n * 8, 2) == 0
    return host


def decode(buf, cells):
    """(counts [y][j] as float64, llh, sum_r, n_seqs) of the raw accumulator"""
    return buf[:cells] * 2.0 ** -40, buf[cells] * 2.0 ** -24, buf[cells + 1] * 2.0 ** -30, int(buf[cells + 2])


@pytest.mark.timeout(900)
def test_full_size_invariants(gpu_ctx):
    N, L0, W, K = 1_000_000, 200, 20, 2
    pwm = synth.make_pwm(W, 1234)
    codes, off = synth.make_sequences(N, L0, pwm, 1234)
    pk = bm.PackedSeqs.from_codes(codes, off, False, seed=42)
    vbg = pk.bg_model(2, np.array([1, 10, 10], np.float32))
    A = synth.alpha_matrix(synth.default_alpha(K), W)
    v0 = synth.bamm_from_pwm((0.7 * pwm + 0.075).astype(np.float32), K)
    cells = 4 ** (K + 1) * W

    whole = bm.SeqSet(gpu_ctx, pk)
    em = bm.EM(gpu_ctx, whole, K, W, vbg, A, v0, 0.3)
    em.iterate(8)                                   # into the regime where the sparse M-step is taken
    v8, q8 = em.getV(), em.getQ()
    em.accumulate()
    buf = read_buffer(em, gpu_ctx)
    nK, llh, sum_r, n_seq = decode(buf, cells)
    nK = nK.reshape(4 ** (K + 1), W)
    assert n_seq == N
    assert nK[:, 0].sum() == pytest.approx(sum_r, rel=2e-6)          # mass conservation (r is fp32)
    assert 0.25 * N < sum_r < 0.75 * N and np.isfinite(llh)
    col = nK.sum(axis=0)
    assert np.all(np.diff(col) <= 1e-6 * col[0])                      # later columns lose truncated windows only
    em.accumulate()
    assert np.array_equal(read_buffer(em, gpu_ctx), buf)              # bitwise reproducible

    b0, e0 = pk.shard_range(W, 0, 2)
    b1, e1 = pk.shard_range(W, 1, 2)
    parts = []
    for b, e in ((b0, e0), (b1, e1)):
        ss = bm.SeqSet(gpu_ctx, pk, b, e)
        h = bm.EM(gpu_ctx, ss, K, W, vbg, A, v8, q8)
        h.accumulate()
        parts.append(read_buffer(h, gpu_ctx))
        h.close(); ss.close()
    # linearity over shards: the counts are integer sums of the same addends -> exactly equal; the two
    # statistics are rounded to their fixed-point unit once per block
    assert np.array_equal((parts[0] + parts[1])[:cells], buf[:cells])
    assert (parts[0] + parts[1])[cells + 2] == buf[cells + 2]
    np.testing.assert_allclose((parts[0] + parts[1])[cells:cells + 2], buf[cells:cells + 2], rtol=1e-9)

    em.update()
    v = em.getV()
    n = em.getCounts()
    v0_tab = v[:4 * W].reshape(4, W)
    np.testing.assert_allclose(v0_tab.sum(axis=0), 1.0, atol=3e-7)   # Motif.h:110-118
    for k in range(K, 0, -1):                                          # EM.cpp:247-254
        hi = n[bm.v_offset(k, W):bm.v_offset(k + 1, W)].reshape(4, 4 ** k, W).astype(np.float64).sum(axis=0)
        lo = n[bm.v_offset(k - 1, W):bm.v_offset(k, W)].reshape(4 ** k, W)
        np.testing.assert_allclose(lo, hi, rtol=3e-7)
    # v <= 1 is NOT an invariant: 3-mers overlapping the strand separator (N) are left out of the counts
    # (Sequence.cpp:34-41 / EM.cpp:232), so a context can be rarer at column j-1 than its continuations at j.
    assert np.all(v > 0) and np.all(np.isfinite(v))
    em.close(); whole.close()


@pytest.mark.timeout(1200)
def test_full_size_invariants_c4(gpu_ctx):
    """BASELINE config 4 at full size (1M x 500 bp both strands, W = 30, k = 4: the column-sliced path, E pass with
    compacted lists + k_m_list slices), from the seed (pass 1: every window is listed, the longest lists) and after 12
    passes: mass conservation, two half shards == the whole set in int64, bitwise repeat, marginals.  EM.cpp:139-259."""
    N, L0, W, K = 1_000_000, 500, 30, 4
    pwm = synth.make_pwm(W, 1234)
    codes, off = synth.make_sequences(N, L0, pwm, 1234)
    pk = bm.PackedSeqs.from_codes(codes, off, False, seed=42)
    del codes
    vbg = pk.bg_model(2, np.array([1, 10, 10], np.float32))
    A = synth.alpha_matrix(synth.default_alpha(K), W)
    v0 = synth.bamm_from_pwm((0.7 * pwm + 0.075).astype(np.float32), K)
    cells = 4 ** (K + 1) * W

    whole = bm.SeqSet(gpu_ctx, pk)
    gpu_ctx.set_tuning(adaptive_lists=0)                      # lists in every pass: pass 1 is their worst case
    try:
        em = bm.EM(gpu_ctx, whole, K, W, vbg, A, v0, 0.3, max_iterations=40)
    finally:
        gpu_ctx.set_tuning(adaptive_lists=1)
    g, o, launches = em.plan()
    assert g == 0 and o == N                                  # the sliced path, one length class

    def check(em, v_in, q_in, label):
        em.accumulate()
        buf = read_buffer(em, gpu_ctx)
        nK, llh, sum_r, n_seq = decode(buf, cells)
        nK = nK.reshape(4 ** (K + 1), W)
        assert n_seq == N, label
        assert nK[:, 0].sum() == pytest.approx(sum_r, rel=2e-6), label          # mass conservation
        assert 0.05 * N < sum_r < 0.95 * N and np.isfinite(llh), label
        col = nK.sum(axis=0)
        assert np.all(np.diff(col) <= 1e-6 * col[0]), label                    # later columns lose truncated windows only
        em.accumulate()
        assert np.array_equal(read_buffer(em, gpu_ctx), buf), label            # bitwise reproducible
        parts = []
        for r in range(2):
            b, e = pk.shard_range(W, r, 2)
            ss = bm.SeqSet(gpu_ctx, pk, b, e)
            h = bm.EM(gpu_ctx, ss, K, W, vbg, A, v_in, q_in, n_seqs_bound=N)
            h.accumulate()
            parts.append(read_buffer(h, gpu_ctx))
            h.close(); ss.close()
        assert np.array_equal((parts[0] + parts[1])[:cells], buf[:cells]), label   # integer sums of the same addends
        assert (parts[0] + parts[1])[cells + 2] == buf[cells + 2], label
        np.testing.assert_allclose((parts[0] + parts[1])[cells:cells + 2], buf[cells:cells + 2], rtol=1e-9)
        return buf

    check(em, v0, 0.3, "from the seed")                       # pass 1: the longest lists
    # the dense walk (round-1 path, no lists) and the default handle (lists or dense r per pass, chosen on the
    # device: dense in pass 1) add the same integers, pass after pass
    gpu_ctx.set_tuning(e_list=0)
    try:
        dense = bm.EM(gpu_ctx, whole, K, W, vbg, A, v0, 0.3, max_iterations=4)
    finally:
        gpu_ctx.set_tuning(e_list=1)
    auto = bm.EM(gpu_ctx, whole, K, W, vbg, A, v0, 0.3, max_iterations=40)
    dense.accumulate(); em.accumulate(); auto.accumulate()
    first = read_buffer(em, gpu_ctx)[:cells]
    assert np.array_equal(read_buffer(dense, gpu_ctx)[:cells], first) and np.array_equal(read_buffer(auto, gpu_ctx)[:cells], first)
    dense.close()
    em.update(); auto.update()
    em.iterate(11); auto.iterate(11)
    v12, q12 = em.getV(), em.getQ()
    assert np.array_equal(auto.getV(), v12)                   # whichever flavour each pass took
    auto.close()
    check(em, v12, q12, "after 12 passes")
    em.update()
    v, n = em.getV(), em.getCounts()
    np.testing.assert_allclose(v[:4 * W].reshape(4, W).sum(axis=0), 1.0, atol=3e-7)   # Motif.h:110-118
    for k in range(K, 0, -1):                                                          # EM.cpp:247-254
        hi = n[bm.v_offset(k, W):bm.v_offset(k + 1, W)].reshape(4, 4 ** k, W).astype(np.float64).sum(axis=0)
        lo = n[bm.v_offset(k - 1, W):bm.v_offset(k, W)].reshape(4 ** k, W)
        np.testing.assert_allclose(lo, hi, rtol=3e-7)
    assert np.all(v > 0) and np.all(np.isfinite(v))
    em.close(); whole.close()

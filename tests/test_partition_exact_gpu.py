"""The pass's int64 accumulator -- counts, log-likelihood, sum of responsibilities, sequence count -- does not depend
on how the sequences are split over waves, blocks or ranks (sequences shard, one all-reduce per iteration).

Counts were exact integers from round 1 on; llh and sum_r used to be fp64 sums per wave and block, rounded to the
accumulator's units once per block, so their last bits followed the partition.  Each sequence's contribution is now
rounded to those units BEFORE it is summed (device_utils.h: stat_round_llh / stat_round_sumr), which makes every sum
exact.  Holds as long as a sequence goes through the same kernel flavour on either side: uniform and mixed rows
multiply a window's odds in different groupings and differ in the last bit of r, which is why a shard plans its
kernels from the global size the caller names (n_seqs_bound), not from its own.
Reference lines: EM.cpp:195 (llh), EM.cpp:509-513 (sum of r for q), EM.cpp:236-242 (counts)."""
import ctypes as C

import numpy as np
import pytest

import bammmotif2_amd as bm
from bammmotif2_amd import synth

pytestmark = pytest.mark.gpu

SHAPES = [
    dict(N=9000, L0=200, W=20, K=2, n_frac=0.003, ragged=50, tune=dict()),                  # uniform rows + per-column bucket
    dict(N=50000, L0=200, W=20, K=2, tune=dict()),                                           # mixed rows: planned from the GLOBAL size
    dict(N=50000, L0=200, W=20, K=2, tune=dict(group_layout=3)),                              # uniform rows at that size
    dict(N=6000, L0=150, W=15, K=1, ss=True, ragged=40, tune=dict()),
    dict(N=4000, L0=200, W=20, K=3, ragged=30, tune=dict()),
    dict(N=2500, L0=400, W=24, K=4, ragged=80, n_frac=0.002, tune=dict()),                   # column-sliced path
    dict(N=40, L0=6000, W=12, K=2, ragged=3000, tune=dict()),                                 # beyond the length classes
]


def accumulator(ctx, hip, pk, begin, end, shape, vbg, A, v0, launch=(0, 0)):
    ctx.set_launch(*launch)
    ctx.set_tuning(**shape["tune"])
    try:
        ss = bm.SeqSet(ctx, pk, begin, end)
        em = bm.EM(ctx, ss, shape["K"], shape["W"], vbg, A, v0, 0.3, n_seqs_bound=shape["N"])
    finally:
        ctx.set_launch(0, 0)
        ctx.set_tuning(group_layout=-1)
    em.accumulate()
    ctx.sync()
    p, n = em.reduce_buffer()
    h = np.zeros(n, np.int64)
    assert hip.hipMemcpy(h.ctypes.data_as(C.c_void_p), C.c_void_p(p), n * 8, 2) == 0
    em.close(); ss.close()
    return h


@pytest.mark.parametrize("shape", SHAPES, ids=[f"K{d['K']}_W{d['W']}_N{d['N']}_{i}" for i, d in enumerate(SHAPES)])
def test_accumulator_is_the_same_integers_for_any_partition(shape, gpu_ctx):
    hip = C.CDLL("libamdhip64.so")
    N, W, K = shape["N"], shape["W"], shape["K"]
    pwm = synth.make_pwm(W, 3)
    codes, off = synth.make_sequences(N, shape["L0"], pwm, 3, 0.5, shape.get("n_frac", 0.0), shape.get("ragged", 0))
    pk = bm.PackedSeqs.from_codes(codes, off, shape.get("ss", False), seed=42)
    vbg = pk.bg_model(2, np.array([1.0, 10.0, 10.0], np.float32))
    A = synth.alpha_matrix(synth.default_alpha(K), W)
    v0 = synth.bamm_from_pwm((0.7 * pwm + 0.075).astype(np.float32), K)
    whole = accumulator(gpu_ctx, hip, pk, 0, N, shape, vbg, A, v0)
    cells = 4 ** (K + 1) * W
    assert whole[cells + 2] == N and whole[cells] != 0 and whole[cells + 1] > 0
    # other launch shapes: other waves and blocks get the sequences
    for launch in ((64, 0), (37, 0)):                         # (another block size may mean another kernel flavour)
        assert np.array_equal(accumulator(gpu_ctx, hip, pk, 0, N, shape, vbg, A, v0, launch), whole), launch
    # shards, even and very uneven: the ranks' accumulators add up to the whole, word for word
    for cuts in ((0, N // 2, N), (0, 1, N // 7, N // 7 + 3, (5 * N) // 6, N)):
        total = np.zeros_like(whole)
        for b, e in zip(cuts[:-1], cuts[1:]):
            total += accumulator(gpu_ctx, hip, pk, b, e, shape, vbg, A, v0)
        assert np.array_equal(total[cells:], whole[cells:]), (cuts, total[cells:], whole[cells:])
        assert np.array_equal(total, whole), cuts


def test_shards_with_unlike_length_mixes_plan_like_the_whole_set(gpu_ctx):
    """Next to the planner's threshold (mixed rows from 40 000 sequences of 200 bp or the equivalent up): two length
    classes, the set sorted by length, so that one shard holds all of the long class and the other all of the short one.
    An estimate of a class's global size from the shard's own mix would put the long class above the threshold on its
    shard and below it on the whole set; the plan follows the global size alone, so the flavours -- and the integers --
    are those of the whole set."""
    hip = C.CDLL("libamdhip64.so")
    W, K = 20, 2
    pwm = synth.make_pwm(W, 5)
    c_long, o_long = synth.make_sequences(30000, 200, pwm, 5, 0.5)        # 7 positions per lane
    c_short, o_short = synth.make_sequences(30000, 130, pwm, 6, 0.5)      # 5 positions per lane
    codes = np.concatenate([c_long, c_short])
    off = np.concatenate([o_long, o_short[1:] + o_long[-1]])
    N = 60000
    pk = bm.PackedSeqs.from_codes(codes, off, False, seed=42)
    vbg = pk.bg_model(2, np.array([1.0, 10.0, 10.0], np.float32))
    A = synth.alpha_matrix(synth.default_alpha(K), W)
    v0 = synth.bamm_from_pwm((0.7 * pwm + 0.075).astype(np.float32), K)
    shape = dict(N=N, W=W, K=K, tune=dict())
    plans = []
    for b, e in ((0, N), (0, 30000), (30000, N)):
        ss = bm.SeqSet(gpu_ctx, pk, b, e)
        em = bm.EM(gpu_ctx, ss, K, W, vbg, A, v0, 0.3, n_seqs_bound=N)
        plans.append((em.plan()[0], em.plan_mixed()))
        em.close(); ss.close()
    assert plans[0] == (N, N) and plans[1] == (30000, 30000) and plans[2] == (30000, 30000), plans
    whole = accumulator(gpu_ctx, hip, pk, 0, N, shape, vbg, A, v0)
    total = accumulator(gpu_ctx, hip, pk, 0, 30000, shape, vbg, A, v0) + accumulator(gpu_ctx, hip, pk, 30000, N, shape, vbg, A, v0)
    assert np.array_equal(total, whole)

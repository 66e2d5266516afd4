"""Full-size shapes of BASELINE.json's configs 2 and 5 through the DEFAULT planner against the oracle: one E+M step
from the same seed, all of v against the fp64 restatement (1e-6) and against the faithful fp32 restatement of the
reference (EM.cpp:139-259, Motif.h:95-136; one thread = the reference's summation order) within 1e-5 + the reference's
own distance from exact arithmetic, measured in the same test (its fp32 accumulation drifts with N).
tests/deviation_report.py was the script form of this."""
import numpy as np
import pytest

import bammmotif2_amd as bm
from bammmotif2_amd import synth
from tests import margins

pytestmark = pytest.mark.gpu
W, K = 20, 2


@pytest.mark.parametrize("N,shape", [(50000, "config 2: 50k x 200 bp"), (200000, "config 5 shape: 200k x 200 bp")],
                         ids=["c2_50k", "c5_200k"])
def test_full_size_step_against_fp64_and_fp32_oracle(N, shape, gpu_ctx, orc):
    pwm = synth.make_pwm(W, 1234)
    A = synth.alpha_matrix(synth.default_alpha(K), W)
    v0 = synth.bamm_from_pwm((0.7 * pwm + 0.075).astype(np.float32), K)
    codes, in_off = synth.make_sequences(N, 200, pwm, 1234)
    _, kmer, off = orc.encode_set(codes, in_off, False, 42)
    vbg = orc.bg_model(kmer, off, 2, np.array([1, 10, 10], np.float32))
    v64, *_ = orc.em_step_f64(kmer, off, K, W, 2, vbg, A, v0, 0.3)
    ref = orc.optimize(kmer, off, K, W, 2, vbg, A, v0, 0.3, epsilon=0.0, max_iter=1)     # fp32, one thread
    pk = bm.PackedSeqs.from_codes(codes, in_off, False, seed=42)                          # the product's own packing
    assert np.array_equal(pk.bg_model(2, np.array([1, 10, 10], np.float32)), vbg)
    ss = bm.SeqSet(gpu_ctx, pk)
    em = bm.EM(gpu_ctx, ss, K, W, vbg, A, v0, 0.3)                                      # default planner
    assert em.plan()[0] == N and em.plan_mixed() == N, "these sizes run the bench kernel (k_em_mix) by the planner's own choice"
    em.iterate(1)
    v = em.getV()
    name = f"synthetic {N} x 200 bp ds"
    fl = "k_em_mix (planner)"
    ref_noise = margins.rel(ref["v"], v64, 1.0)
    margins.ROWS.append((name, "fp32 restatement", "v pass 1", "fp64 restatement", ref_noise, float("nan"), 0.0))
    margins.check(name, fl, "v pass 1", v, v64, 1e-6, against="fp64 restatement")
    margins.check(name, fl, "v pass 1", v, ref["v"], 1e-5 + ref_noise, against="fp32 restatement")
    margins.check(name, fl, "llh pass 1", em.getLLH(), ref["llh"], 1e-5, against="fp32 restatement")
    em.close(); ss.close()

import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bammmotif2_amd as bm
from bammmotif2_amd import synth
pre = sys.argv[1] if len(sys.argv) > 1 else "none"
N, L0, W, K = 200000, 2000, 20, 2
pwm = synth.make_pwm(W, 1234)
ctx = bm.Context(0)
if pre == "kernels_first":          # load kernels.hip's code object (4 MB) before the grouped one, as round 4's library did (k_make_s lived there)
    c2, o2 = synth.make_sequences(64, 200, pwm, 5)
    p2 = bm.PackedSeqs.from_codes(c2, o2, False, seed=42)
    s2 = bm.SeqSet(ctx, p2)
    ctx.set_tuning(grouped=0)
    e2 = bm.EM(ctx, s2, K, W, p2.bg_model(2, np.array([1, 10, 10], np.float32)), synth.alpha_matrix(synth.default_alpha(K), W), synth.bamm_from_pwm((0.7 * pwm + 0.075).astype(np.float32), K), 0.3)
    ctx.set_tuning(grouped=1)
    e2.iterate(1); ctx.sync(); e2.close(); s2.close()
codes, off = synth.make_sequences(N, L0, pwm, 1234)
pk = bm.PackedSeqs.from_codes(codes, off, False, seed=42)
vbg = pk.bg_model(2, np.array([1, 10, 10], np.float32))
A = synth.alpha_matrix(synth.default_alpha(K), W)
v0 = synth.bamm_from_pwm((0.7 * pwm + 0.3 * 0.25).astype(np.float32), K)
seqs = bm.SeqSet(ctx, pk)
em = bm.EM(ctx, seqs, K, W, vbg, A, v0, 0.3, max_iterations=60)
em.iterate(10); ctx.sync()
t0 = time.perf_counter(); em.iterate(20); ctx.sync()
print(pre, "%.4f ms per pass" % ((time.perf_counter() - t0) / 20 * 1e3))

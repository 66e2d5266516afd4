import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bammmotif2_amd as bm
from bammmotif2_amd import synth
N, L0, W, K = 1000000, 200, 20, 2
pwm = synth.make_pwm(W, 1234); codes, off = synth.make_sequences(N, L0, pwm, 1234)
pk = bm.PackedSeqs.from_codes(codes, off, False, seed=42)
ctx = bm.Context(0); ss = bm.SeqSet(ctx, pk)
vbg = pk.bg_model(2, np.array([1, 10, 10], np.float32))
A = synth.alpha_matrix(synth.default_alpha(K), W); v0 = synth.bamm_from_pwm((0.7 * pwm + 0.075).astype(np.float32), K)
em = bm.EM(ctx, ss, K, W, vbg, A, v0, 0.3, max_iterations=100)
em.iterate(25); ctx.sync()
t = time.perf_counter()
for _ in range(20): em.EStep()
ctx.sync(); e_ms = (time.perf_counter() - t) / 20 * 1e3
t = time.perf_counter(); em.iterate(20); ctx.sync(); f_ms = (time.perf_counter() - t) / 20 * 1e3
print("E-only ms", e_ms, "fused E+M ms", f_ms, "=> M share", 1 - e_ms / f_ms)

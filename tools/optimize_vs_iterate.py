"""Per-pass cost of optimize() (one host read-back of (llh, v_diff) per pass for the stop rule, EM.cpp:117-118)
against iterate() (no host in the loop), by set size."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bammmotif2_amd as bm
from bammmotif2_amd import synth
W, K, L0 = 20, 2, 200
for N in [300, 2000, 10000, 50000, 200000, 1000000]:
    pwm = synth.make_pwm(W, 1234); codes, off = synth.make_sequences(N, L0, pwm, 1234, plant_frac=0.5)
    pk = bm.PackedSeqs.from_codes(codes, off, False, seed=42)
    ctx = bm.Context(0); ss = bm.SeqSet(ctx, pk)
    vbg = pk.bg_model(2, np.array([1, 10, 10], np.float32))
    A = synth.alpha_matrix(synth.default_alpha(K), W); v0 = synth.bamm_from_pwm((0.7 * pwm + 0.075).astype(np.float32), K)
    n = 40
    em = bm.EM(ctx, ss, K, W, vbg, A, v0, 0.3, max_iterations=n); em.iterate(n); ctx.sync(); em.close()      # warm
    em = bm.EM(ctx, ss, K, W, vbg, A, v0, 0.3, max_iterations=n); ctx.sync()
    t = time.perf_counter(); em.iterate(n); ctx.sync(); it_us = (time.perf_counter() - t) / n * 1e6; em.close()
    em = bm.EM(ctx, ss, K, W, vbg, A, v0, 0.3, max_iterations=n, epsilon=0.0); ctx.sync()
    t = time.perf_counter(); done = em.optimize(); ctx.sync(); dt = time.perf_counter() - t
    iters = em.iteration(); em.close()
    print("N %7d: iterate %8.1f us/pass, optimize %8.1f us/pass over %d passes (+%.1f us, +%.0f %%)" %
          (N, it_us, dt / iters * 1e6, iters, dt / iters * 1e6 - it_us, (dt / iters * 1e6 / it_us - 1) * 100), flush=True)
    ss.close(); ctx.close()

"""Fraction of windows with a non-zero fixed-point addend, E-only and fused pass times, by order K
(1M x 200 bp both strands, W given).  The M-step's LDS adds skip exact zeros: this is what their lane
occupancy looks like once the model is informative (after 25 passes)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bammmotif2_amd as bm
from bammmotif2_amd import synth
N, L0 = 1000000, 200
widths = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [20]
orders = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 3]
for W in widths:
    pwm = synth.make_pwm(W, 1234); codes, off = synth.make_sequences(N, L0, pwm, 1234, plant_frac=0.5)
    pk = bm.PackedSeqs.from_codes(codes, off, False, seed=42)
    ctx = bm.Context(0); ss = bm.SeqSet(ctx, pk)
    vbg = pk.bg_model(2, np.array([1, 10, 10], np.float32))
    for K in orders:
        A = synth.alpha_matrix(synth.default_alpha(K), W)
        v0 = synth.bamm_from_pwm((0.7 * pwm + 0.075).astype(np.float32), K)
        em = bm.EM(ctx, ss, K, W, vbg, A, v0, 0.3, max_iterations=100, n_seqs_bound=N)
        em.iterate(25); ctx.sync()
        t = time.perf_counter()
        for _ in range(20): em.EStep()
        ctx.sync(); e_ms = (time.perf_counter() - t) / 20 * 1e3
        t = time.perf_counter(); em.iterate(20); ctx.sync(); f_ms = (time.perf_counter() - t) / 20 * 1e3
        em.EStep()
        r = em.getR(0, 4000)
        L = 2 * L0 + 1; LW1 = L - W + 1
        nz = float((r >= 2.0 ** -41).sum()) / (4000 * LW1)
        big = float((r >= 2.0 ** -8).sum()) / 4000
        print("W %d K %d: E-only %.3f ms, fused %.3f ms, non-zero windows %.3f of LW1 (%.0f per sequence), r >= 2^-8: %.2f per sequence, plan %s"
              % (W, K, e_ms, f_ms, nz, nz * LW1, big, em.plan()), flush=True)
        em.close()
    ss.close(); ctx.close()

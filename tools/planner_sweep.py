"""Does the planner pick the faster kernel?  For a grid of (K, W, L0): ms per pass through the planner's choice
and through the per-column kernel (set_tuning grouped=0), both strands, ~50M positions per case."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bammmotif2_amd as bm
from bammmotif2_amd import synth
orders = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2, 3]
widths = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [6, 8, 10, 12, 15, 16, 17, 20, 24, 30]
lens = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [50, 100, 200, 500]
TOTAL = 50_000_000
print("K W L0 N | planner ms (grouped, per-column seqs) | per-column ms | ratio", flush=True)
for L0 in lens:
    for W in widths:
        if W >= L0: continue
        N = TOTAL // (2 * L0 + 1)
        pwm = synth.make_pwm(W, 1234); codes, off = synth.make_sequences(N, L0, pwm, 1234, plant_frac=0.5)
        pk = bm.PackedSeqs.from_codes(codes, off, False, seed=42)
        vbg = pk.bg_model(2, np.array([1, 10, 10], np.float32))
        for K in orders:
            A = synth.alpha_matrix(synth.default_alpha(K), W)
            v0 = synth.bamm_from_pwm((0.7 * pwm + 0.075).astype(np.float32), K)
            res = []
            for kv in ({}, {"grouped": 0}):
                ctx = bm.Context(0)
                if kv: ctx.set_tuning(**kv)
                ss = bm.SeqSet(ctx, pk)
                em = bm.EM(ctx, ss, K, W, vbg, A, v0, 0.3, max_iterations=100, n_seqs_bound=N)
                em.iterate(12); ctx.sync()
                t = time.perf_counter(); em.iterate(10); ctx.sync(); ms = (time.perf_counter() - t) / 10 * 1e3
                res.append((ms, em.plan()))
                em.close(); ss.close(); ctx.close()
            flag = "  <-- planner slower" if res[0][0] > 1.03 * res[1][0] and res[0][1][0] > 0 else ""
            print("%d %2d %3d %7d | %.3f (%d, %d) | %.3f | %.2f%s" % (K, W, L0, N, res[0][0], res[0][1][0], res[0][1][1], res[1][0], res[0][0] / res[1][0], flag), flush=True)

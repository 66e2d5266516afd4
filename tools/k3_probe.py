"""K = 3 through the grouped kernel: pass time by width and table layout (probe for the W-dependence)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bammmotif2_amd as bm
from bammmotif2_amd import synth
N, L0, K = 1000000, 200, 3
widths = [int(x) for x in sys.argv[1].split(",")]
variants = [("default", {})]
for W in widths:
    pwm = synth.make_pwm(W, 1234); codes, off = synth.make_sequences(N, L0, pwm, 1234, plant_frac=0.5)
    pk = bm.PackedSeqs.from_codes(codes, off, False, seed=42)
    vbg = pk.bg_model(2, np.array([1, 10, 10], np.float32))
    A = synth.alpha_matrix(synth.default_alpha(K), W)
    v0 = synth.bamm_from_pwm((0.7 * pwm + 0.075).astype(np.float32), K)
    for name, kv in variants:
        ctx = bm.Context(0)
        if kv: ctx.set_tuning(**kv)
        ss = bm.SeqSet(ctx, pk)
        try:
            em = bm.EM(ctx, ss, K, W, vbg, A, v0, 0.3, max_iterations=100, n_seqs_bound=N)
        except Exception as e:
            print("W %d %s: %s" % (W, name, e)); ss.close(); ctx.close(); continue
        em.iterate(25); ctx.sync()
        t = time.perf_counter(); em.iterate(20); ctx.sync(); f_ms = (time.perf_counter() - t) / 20 * 1e3
        print("W %d %-10s fused %.3f ms  llh %.3f  plan %s" % (W, name, f_ms, em.trace()[0][-1], em.plan()), flush=True)
        em.close(); ss.close(); ctx.close()

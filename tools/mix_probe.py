"""Mixed rows (group_layout 8) against the uniform grouped kernel (group_layout 3): ms per pass, K = 2."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bammmotif2_amd as bm
from bammmotif2_amd import synth
N, L0, K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000, int(sys.argv[3]) if len(sys.argv) > 3 else 200, 2
widths = [int(x) for x in sys.argv[1].split(",")]
for W in widths:
    pwm = synth.make_pwm(W, 1234); codes, off = synth.make_sequences(N, L0, pwm, 1234, plant_frac=0.5)
    pk = bm.PackedSeqs.from_codes(codes, off, False, seed=42)
    vbg = pk.bg_model(2, np.array([1, 10, 10], np.float32))
    A = synth.alpha_matrix(synth.default_alpha(K), W)
    v0 = synth.bamm_from_pwm((0.7 * pwm + 0.075).astype(np.float32), K)
    out = {}
    for name, kv in (("uniform", {"group_layout": 3}), ("mixed", {"group_layout": 8}), ("planner", {})):
        ctx = bm.Context(0)
        if kv: ctx.set_tuning(**kv)
        ss = bm.SeqSet(ctx, pk)
        em = bm.EM(ctx, ss, K, W, vbg, A, v0, 0.3, max_iterations=100, n_seqs_bound=N)
        em.iterate(25); ctx.sync()
        t = time.perf_counter(); em.iterate(20); ctx.sync(); ms = (time.perf_counter() - t) / 20 * 1e3
        out[name] = (ms, em.trace()[0][-1], em.getV().copy(), em.plan())
        em.close(); ss.close(); ctx.close()
    dv = float(np.max(np.abs(out["uniform"][2] - out["mixed"][2]) / np.maximum(np.abs(out["uniform"][2]), 1e-30)))
    print("W %2d: uniform %.3f ms, mixed %.3f ms, planner %.3f ms | llh %.3f / %.3f | max rel dv %.2e | plans %s %s" %
          (W, out["uniform"][0], out["mixed"][0], out["planner"][0], out["uniform"][1], out["mixed"][1], dv, out["uniform"][3], out["mixed"][3]), flush=True)

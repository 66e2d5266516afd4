#!/usr/bin/env python3
"""What the FIRST passes of a process cost, call by call (host clock around every ABI call + stream sync), on the bench
workload (1M x 200 bp both strands, W = 20, k = 2) -- nothing warmed up beforehand.
    python tools/first_passes.py MODE [nseq] [passes]
MODE  plain     iterate(1) x passes
      getr      the same with getR(0, 2000) after passes 1, 2, 3 (what tools/pass_times.py does)
      getr_big  getR(0, 20000) (a block above the scratch pool's 4 MB threshold) instead
      fused     ONE iterate(passes) call (updates fused into the next pass's prologue), then a second one
      prewarm_getr       a first handle (one pass, closed), then `getr` on a second one, the results post-processed and dropped
      prewarm_getr_keep  the same, every getR result kept alive
      optimize  optimize() with epsilon 0 and max_iterations = passes (the CLI's call)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

T0 = time.perf_counter()
import bammmotif2_amd as bm
from bammmotif2_amd import synth

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
P = int(sys.argv[3]) if len(sys.argv) > 3 else 8
L0, W, K = 200, 20, 2
pwm = synth.make_pwm(W, 1234)
codes, off = synth.make_sequences(N, L0, pwm, 1234)
pk = bm.PackedSeqs.from_codes(codes, off, False, seed=42)
vbg = pk.bg_model(2, np.array([1, 10, 10], np.float32))
A = synth.alpha_matrix(synth.default_alpha(K), W)
v0 = synth.bamm_from_pwm((0.7 * pwm + 0.3 * 0.25).astype(np.float32), K)
rows = []


def timed(what, f, sync=None):
    t = time.perf_counter()
    r = f()
    if sync:
        sync()
    rows.append((what, (time.perf_counter() - t) * 1e3))
    return r


ctx = timed("Context(0)", lambda: bm.Context(0))
seqs = timed("SeqSet (upload)", lambda: bm.SeqSet(ctx, pk), ctx.sync)
em = timed("EM create", lambda: bm.EM(ctx, seqs, K, W, vbg, A, v0, 0.3, max_iterations=max(P, 1) + 8, epsilon=0.0), ctx.sync)
keep = []
if mode.startswith("prewarm"):                       # what tools/pass_times.py did up to round 4: a first handle, one pass, closed
    timed("iterate(1) of the pre-warm handle", lambda: em.iterate(1), ctx.sync)
    timed("close of the pre-warm handle", lambda: em.close())
    em = timed("EM create (second handle)", lambda: bm.EM(ctx, seqs, K, W, vbg, A, v0, 0.3, max_iterations=max(P, 1) + 8, epsilon=0.0), ctx.sync)
    for p in range(P):
        timed(f"iterate(1) pass {p + 1}", lambda: em.iterate(1), ctx.sync)
        if p < 3:
            r = timed(f"getR(0, 2000) after pass {p + 1}", lambda: em.getR(0, 2000))
            if mode == "prewarm_getr_keep":
                keep.append(r)
            elif mode == "prewarm_getr":
                r = np.concatenate([np.asarray(x, np.float64).ravel() for x in r]) if isinstance(r, (list, tuple)) else np.asarray(r, np.float64).ravel()
                _ = float((r * 2.0 ** 40 >= 1.0).mean())
if mode.startswith("prewarm"):
    pass
elif mode in ("plain", "getr", "getr_big"):
    for p in range(P):
        timed(f"iterate(1) pass {p + 1}", lambda: em.iterate(1), ctx.sync)
        if mode != "plain" and p < 3:
            n = 2000 if mode == "getr" else 20000
            timed(f"getR(0, {n}) after pass {p + 1}", lambda: em.getR(0, n))
elif mode == "fused":
    timed(f"iterate({P}) first call", lambda: em.iterate(P), ctx.sync)
    timed(f"iterate({P}) second call", lambda: em.iterate(P), ctx.sync)
elif mode == "optimize":
    em.close()
    em = timed("EM create (max_iterations = passes)", lambda: bm.EM(ctx, seqs, K, W, vbg, A, v0, 0.3, max_iterations=P, epsilon=0.0), ctx.sync)
    it = timed("optimize() first call", lambda: em.optimize(), ctx.sync)
    rows.append((f"  ... {it} passes", 0.0))
    em.close()
    em = timed("EM create (second handle)", lambda: bm.EM(ctx, seqs, K, W, vbg, A, v0, 0.3, max_iterations=P, epsilon=0.0), ctx.sync)
    it = timed("optimize() of a second handle", lambda: em.optimize(), ctx.sync)
else:
    raise SystemExit(__doc__)

print(f"# mode {mode}: {N} x {L0} bp both strands, W = {W}, k = {K}; ms per call (host clock, stream synchronised after each)")
for what, ms in rows:
    print(f"{ms:10.3f}  {what}")
em.close(); seqs.close(); ctx.close()

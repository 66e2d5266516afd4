#!/usr/bin/env python3
"""Config 4 (1M x 500 bp both strands, W = 30, k = 4) from the seed: milliseconds of every one of the first passes,
with the E pass handing the M slices compacted lists (default) and dense r (e_list = 0).  Round 3 measured a per-pass
choice between the two made on the device from the previous pass's count of non-zero windows: in pass 1 both cost 15 ms
(every window has a non-zero addend: the LDS adds, not the lists, are the bound), so there is nothing to choose.
    python tools/c4_cold_passes.py [nseq]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bammmotif2_amd as bm
from bammmotif2_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L0, W, K = 500, 30, 4
pwm = synth.make_pwm(W, 1234)
codes, off = synth.make_sequences(N, L0, pwm, 1234)
pk = bm.PackedSeqs.from_codes(codes, off, False, seed=42)
vbg = pk.bg_model(2, np.array([1, 10, 10], np.float32))
A = synth.alpha_matrix(synth.default_alpha(K), W)
v0 = synth.bamm_from_pwm((0.7 * pwm + 0.3 * 0.25).astype(np.float32), K)
ctx = bm.Context(0)
seqs = bm.SeqSet(ctx, pk)
for label, tune in (("lists", dict(e_list=1)), ("dense r", dict(e_list=0))):
    ctx.set_tuning(**tune)
    em = bm.EM(ctx, seqs, K, W, vbg, A, v0, 0.3, max_iterations=40)
    ctx.set_tuning(e_list=1)
    em.iterate(1); ctx.sync()            # allocations of the first pass
    em.close()
    ctx.set_tuning(**tune)
    em = bm.EM(ctx, seqs, K, W, vbg, A, v0, 0.3, max_iterations=40)
    ctx.set_tuning(e_list=1)
    ms = []
    for p in range(20):
        ctx.sync(); t0 = time.perf_counter(); em.iterate(1); ctx.sync(); ms.append((time.perf_counter() - t0) * 1e3)
    print(f"{label:14s} mean {np.mean(ms):6.2f} ms  passes:", " ".join(f"{x:.1f}" for x in ms), flush=True)
    em.close()

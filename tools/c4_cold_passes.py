#!/usr/bin/env python3
"""Config 4 (1M x 500 bp both strands, W = 30, k = 4) from the seed: milliseconds of every one of the first passes,
with the E pass handing the M slices compacted lists in every pass, dense r in every pass, and the per-pass choice
between the two made on the device from the previous pass's count of non-zero windows (the default), at several thresholds.
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
for label, tune in (("lists always", dict(adaptive_lists=0)), ("dense r always", dict(e_list=0)), ("chosen per pass, 30 %", dict(list_threshold_pct=30)),
                    ("chosen per pass, 45 %", dict(list_threshold_pct=45)), ("chosen per pass, 60 %", dict(list_threshold_pct=60))):
    ctx.set_tuning(**tune)
    em = bm.EM(ctx, seqs, K, W, vbg, A, v0, 0.3, max_iterations=40)
    ctx.set_tuning(adaptive_lists=1, e_list=1, list_threshold_pct=45)
    em.iterate(1); ctx.sync()            # allocations of the first pass
    em.close()
    ctx.set_tuning(**tune)
    em = bm.EM(ctx, seqs, K, W, vbg, A, v0, 0.3, max_iterations=40)
    ctx.set_tuning(adaptive_lists=1, e_list=1, list_threshold_pct=45)
    ms = []
    for p in range(20):
        ctx.sync(); t0 = time.perf_counter(); em.iterate(1); ctx.sync(); ms.append((time.perf_counter() - t0) * 1e3)
    print(f"{label:22s} mean {np.mean(ms):6.2f} ms  passes:", " ".join(f"{x:.1f}" for x in ms), flush=True)
    em.close()

#!/usr/bin/env python3
"""The fixed cost of one pass: the bench model on sets too small to matter (one sequence per wave and less), the
kernel each planner choice runs (rocprofv3 --kernel-trace --stats around this script names them and their durations).
    python tools/fixed_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bammmotif2_amd as bm
from bammmotif2_amd import synth

L0, W, K = 200, 20, 2
pwm = synth.make_pwm(W, 1234)
ctx = bm.Context(0)
for N, layout, fused in ((256, 8, 1), (4096, 8, 1), (4096, 8, 0), (4096, 3, 1), (16384, 8, 1), (16384, 3, 1)):
    codes, off = synth.make_sequences(N, L0, pwm, 1234)
    pk = bm.PackedSeqs.from_codes(codes, off, False, seed=42)
    vbg = pk.bg_model(2, np.array([1, 10, 10], np.float32))
    A = synth.alpha_matrix(synth.default_alpha(K), W)
    v0 = synth.bamm_from_pwm((0.7 * pwm + 0.3 * 0.25).astype(np.float32), K)
    seqs = bm.SeqSet(ctx, pk)
    ctx.set_tuning(group_layout=layout, fused_update=fused)
    em = bm.EM(ctx, seqs, K, W, vbg, A, v0, 0.3, max_iterations=400)
    ctx.set_tuning(group_layout=-1, fused_update=1)
    em.iterate(40); ctx.sync()
    t0 = time.perf_counter(); em.iterate(300); ctx.sync(); dt = (time.perf_counter() - t0) / 300 * 1e6
    print(f"N={N:6d} layout={layout} fused={fused}: {dt:7.2f} us per pass", flush=True)
    em.close(); seqs.close()

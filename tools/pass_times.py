#!/usr/bin/env python3
"""Milliseconds of every one of the first passes of the bench workload (1M x 200 bp both strands, W = 20, k = 2) from the
seed model, on the first handle of a fresh process (up to round 4 a first handle was run and closed beforehand), and the share of exactly-zero responsibilities (r < 2^-40 adds nothing to the integer counts) on a sample.
    python tools/pass_times.py [nseq] [passes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bammmotif2_amd as bm
from bammmotif2_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 60
L0, W, K = 200, 20, 2
pwm = synth.make_pwm(W, 1234)
codes, off = synth.make_sequences(N, L0, pwm, 1234)
pk = bm.PackedSeqs.from_codes(codes, off, False, seed=42)
vbg = pk.bg_model(2, np.array([1, 10, 10], np.float32))
A = synth.alpha_matrix(synth.default_alpha(K), W)
v0 = synth.bamm_from_pwm((0.7 * pwm + 0.3 * 0.25).astype(np.float32), K)
ctx = bm.Context(0)
seqs = bm.SeqSet(ctx, pk)
em = bm.EM(ctx, seqs, K, W, vbg, A, v0, 0.3, max_iterations=P + 8)      # the process's FIRST handle: nothing was warmed up
ms, nz = [], []
for p in range(P):
    ctx.sync(); t0 = time.perf_counter(); em.iterate(1); ctx.sync(); ms.append((time.perf_counter() - t0) * 1e3)
    if p in (0, 1, 2, 4, 7, 11, 15, 19, 24, 39, 59, 99, 199):
        r = em.getR(0, 2000)
        r = np.concatenate([np.asarray(x, np.float64).ravel() for x in r]) if isinstance(r, (list, tuple)) else np.asarray(r, np.float64).ravel()
        nz.append((p + 1, float((r * 2.0 ** 40 >= 1.0).mean())))
print("ms per pass (host clock around iterate(1) + sync, ~10 us of it the sync):")
for i in range(0, P, 10):
    print(f"  passes {i + 1:3d}-{min(i + 10, P):3d}:", " ".join(f"{x:.3f}" for x in ms[i:i + 10]))
print("share of windows with a non-zero integer addend after pass p (first 2000 sequences):", nz)

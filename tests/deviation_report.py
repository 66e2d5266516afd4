#!/usr/bin/env python3
"""SURVEY H4 / BASELINE.md: beyond ~10k sequences the reference's fp32 accumulation drifts away
from exact arithmetic.  One E+M step from the same seed on N sequences: max relative deviation
of v between (a) the faithful fp32 oracle (= the reference, 1 thread), (b) the HIP path and
(c) the fp64 oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import bammmotif2_amd as bm  # noqa: E402
import oracle  # noqa: E402
from bammmotif2_amd import synth  # noqa: E402

import argparse  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("sizes", nargs="*", type=int, default=[1000, 10000, 50000])
ap.add_argument("--width", type=int, default=20)
ap.add_argument("--order", type=int, default=2)
ap.add_argument("--len", type=int, default=200)
args = ap.parse_args()
W, K = args.width, args.order
O = oracle.Oracle()
O.set_threads(1)
ctx = bm.Context(0)
pwm = synth.make_pwm(W, 1234)
A = synth.alpha_matrix(synth.default_alpha(K), W)
v0 = synth.bamm_from_pwm((0.7 * pwm + 0.075).astype(np.float32), K)
print(f"N x {args.len} bp, W = {W}, k = {K}:  |gpu-f64|  |ref32-f64|  |gpu-ref32|   (max relative over v)", flush=True)
for N in args.sizes:
    codes, off = synth.make_sequences(N, args.len, pwm, 1234)
    _, kmer, o = O.encode_set(codes, off, False, 42)
    vbg = O.bg_model(kmer, o, 2, np.array([1, 10, 10], np.float32))
    v64, _, _, _ = O.em_step_f64(kmer, o, K, W, 2, vbg, A, v0, 0.3)
    res = O.optimize(kmer, o, K, W, 2, vbg, A, v0, 0.3, epsilon=0.0, max_iter=1)
    pk = bm.PackedSeqs.from_kmers(kmer, o)
    ss = bm.SeqSet(ctx, pk)
    em = bm.EM(ctx, ss, K, W, vbg, A, v0, 0.3)
    em.iterate(1)
    vg = em.getV()
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.abs(b)))
    print(f"{N}  {rel(vg, v64):.2e}  {rel(res['v'], v64):.2e}  {rel(vg, res['v']):.2e}", flush=True)
    em.close(); ss.close()

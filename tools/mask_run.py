#!/usr/bin/env python3
"""Timing of EM::mask (--advanceEM) on the device: N x 200 bp double strand, W=20, k=2, f=0.05."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import bammmotif2_amd as bm
from bammmotif2_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
oq = len(sys.argv) > 2 and sys.argv[2] == "optq"
W, K = 20, 2
ctx = bm.Context(0)
pwm = synth.make_pwm(W, 1234)
codes, off = synth.make_sequences(N, 200, pwm, 1234)
pk = bm.PackedSeqs.from_codes(codes, off, False, seed=42)
vbg = pk.bg_model(2, np.array([1, 10, 10], np.float32))
A = synth.alpha_matrix(synth.default_alpha(K), W)
v0 = synth.bamm_from_pwm((0.7 * pwm + 0.075).astype(np.float32), K)
ss = bm.SeqSet(ctx, pk)
for rep in range(2):
    em = bm.EM(ctx, ss, K, W, vbg, A, v0, 0.3, optimizeQ=oq, epsilon=0.0, max_iterations=20)
    ctx.sync(); t0 = time.perf_counter()
    it = em.mask(0.05)
    ctx.sync(); t1 = time.perf_counter()
    ms, launches = em.kernel_time()
    print(f"N={N} optq={oq} rep={rep}: mask() {1e3*(t1-t0):.1f} ms total, {it} masked iterations, "
          f"E+M kernels {ms/launches:.3f} ms/iteration, listed={em.last_mask['listed']} cutoff={em.last_mask['cutoff']:.3g} q={em.getQ():.4f}")
    em.close()

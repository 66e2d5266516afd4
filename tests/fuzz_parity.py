"""Randomised parity sweep (GPU): random shapes through one E+M step against the oracle's fp64
restatement (1e-6 on v, the bar of test_grouped_kernel_matches_exact_arithmetic; every third case with a
CV-fold mask) and the scorer bit-exact against the oracle.  Not collected by pytest (no test_ prefix); run on the GPU box:

    python -m tests.fuzz_parity --n 300 --seed 1

Prints one line per failing configuration and a summary; exit code 1 if anything failed."""
from __future__ import annotations

import argparse
import sys

import numpy as np

import bammmotif2_amd as bm
from oracle import Oracle
from tests.cases import Case


def random_spec(rng, i, budget):
    if rng.random() < 0.12:                                  # a shape the mixed-row kernel (csrc/mixed_kernel.h) takes
        W = int(rng.choice([13, 14, 16, 17, 20]))
        Lmax = int(rng.integers(200, 641))                   # 4..10 positions per lane
        rag = int(rng.integers(0, 40))
        L0 = max((Lmax - 1) // 2, W + rag + 1)
        N = int(np.clip(budget // (2 * L0), 6, 20000))
        return dict(name=f"f{i}", N=N, L0=L0, W=W, K=2, ss=False, ragged=rag, n_frac=float(rng.choice([0.0, 0.001, 0.004])),
                    bg_order=int(rng.integers(0, 4)), seed=int(rng.integers(1, 1 << 30)), mix=True)
    K = int(rng.choice([0, 1, 2, 2, 2, 3, 4]))
    ss = bool(rng.integers(0, 2))
    W = int(rng.integers(1, 41 if K <= 2 else 25))
    # positions per lane from 1 to 128: log-uniform length
    Lmax = int(2 ** rng.uniform(np.log2(max(W + 2, 20)), np.log2(4300 if K <= 3 else 2600)))
    rag = int(rng.integers(0, max(1, Lmax // 3)))
    L0 = Lmax if ss else max((Lmax - 1) // 2, W + rag + 1)
    L0 = max(L0, W + rag + 1)
    N = int(np.clip(budget // (L0 * (1 if ss else 2)), 6, 20000))
    n_frac = float(rng.choice([0.0, 0.0, 0.001, 0.004, 0.02]))
    return dict(name=f"f{i}", N=N, L0=L0, W=W, K=K, ss=ss, ragged=rag, n_frac=n_frac,
                bg_order=int(rng.integers(0, 4)), seed=int(rng.integers(1, 1 << 30)))


def run(n, seed, budget, ctx=None, orc=None, verbose=True):
    """(failures, kernels used) over n random cases."""
    rng = np.random.default_rng(seed)
    orc = orc or Oracle()
    ctx = ctx or bm.Context(0)
    bad = 0
    kernels = {}
    for i in range(n):
        spec = random_spec(rng, i, budget)
        try:
            mix_shape = spec.pop("mix", False)
            c = Case(**spec)
            seq, kmer, off, vbg = c.encode(orc)
            pk = bm.PackedSeqs.from_kmers(kmer, off)
            ss = bm.SeqSet(ctx, pk)
            mask = None
            if rng.random() < 0.3:                           # a CV fold over the shared resident set (FDR.cpp:49-57)
                mask = (rng.random(c.N) < 0.7).astype(np.uint8)
                mask[int(rng.integers(0, c.N))] = 1
            # half of the K = 2 cases ask for the mixed-row kernel (the planner keeps it for large launches); where it
            # does not apply the sequences go one column at a time
            want_mix = mix_shape or (c.K == 2 and rng.random() < 0.3)
            if want_mix:
                ctx.set_tuning(group_layout=8)
            try:
                em = bm.EM(ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q, bg_order=c.bg_order, mask=mask)
            finally:
                if want_mix:
                    ctx.set_tuning(group_layout=-1)
            g, o, _ = em.plan()
            kind = "grouped" if g and not o else "per-column/sliced" if o and not g else "mixed"
            kernels[kind] = kernels.get(kind, 0) + 1
            if em.plan_mixed():
                kernels["of these, mixed rows (k_em_mix)"] = kernels.get("of these, mixed rows (k_em_mix)", 0) + 1
            if mask is None:
                km_e, off_e = kmer, off
            else:
                keep = np.nonzero(mask)[0]
                lens = np.diff(off.astype(np.int64))
                km_e = np.concatenate([kmer[int(off[n]):int(off[n + 1])] for n in keep])
                off_e = np.concatenate([[0], np.cumsum(lens[keep])]).astype(np.uint64)
            v64, n64, llh64, _ = orc.em_step_f64(km_e, off_e, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q)
            # the E-only and r-writing flavours of the same kernels (EStep, getR) against the fp32 oracle
            em.EStep()
            Kb0 = min(c.bg_order, c.K)
            r_o, _ = orc.estep(km_e, off_e, c.K, c.W, orc.linear_s(c.v0, vbg, c.K, c.W, Kb0), c.q)
            Lmax = int(np.diff(off_e.astype(np.int64)).max())
            if mask is None:
                # W <= 2: the reference's sequential fp32 Z sum over near-equal terms rounds one-sidedly (3e-5 seen at
                # 1000 positions, 6.2e-5 at 3100: it grows with the length, and it is the oracle's side of the comparison --
                # v against fp64 below is unaffected)
                np.testing.assert_allclose(em.getR(), r_o, rtol=(7e-5 if c.W <= 2 else 2e-5) * max(1.0, 4e-4 * Lmax), atol=1e-12)
            np.testing.assert_allclose(em.getLLH(), llh64, rtol=2e-6, atol=1e-5 + 1e-7 * len(off_e))
            em.iterate(1)
            np.testing.assert_allclose(em.getV(), v64, rtol=1e-6, atol=1e-9)
            np.testing.assert_allclose(em.getCounts(), n64, rtol=3e-6, atol=2e-7)
            # log Z per sequence in fp32 (v_log_f32): ~1e-8 absolute per term, the terms may cancel in the sum
            np.testing.assert_allclose(em.trace()[0][-1], llh64, rtol=2e-6, atol=1e-5 + 1e-7 * len(off_e))
            v = em.getV()
            Kb = min(c.bg_order, c.K)
            mops_o, zoops_o, z_o = orc.logodds(kmer, off, c.K, c.W, orc.log_s(v, vbg, c.K, c.W, Kb))
            mops, zoops, z = bm.logodds(ctx, ss, c.K, c.W, c.bg_order, v, vbg)
            assert np.array_equal(mops, mops_o) and np.array_equal(zoops, zoops_o) and np.array_equal(z, z_o), "scorer"
            em.close(); ss.close()
        except Exception as e:  # noqa: BLE001 -- report and go on
            bad += 1
            print("FAIL", spec, "->", type(e).__name__, str(e).replace("\n", " ")[:300], flush=True)
        if verbose and (i + 1) % 25 == 0:
            print(f"[{i + 1}/{n}] failures so far: {bad}", flush=True)
    return bad, kernels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--budget", type=int, default=60000, help="positions per case (sets the number of sequences)")
    args = ap.parse_args()
    bad, kernels = run(args.n, args.seed, args.budget)
    print("kernels:", kernels, "failures:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

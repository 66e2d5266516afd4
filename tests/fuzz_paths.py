"""Randomised shapes through the GPU test bodies of the seeding, --advanceEM mask and shapes tests
(their own tolerances).  Not collected by pytest; run on the GPU box:

    python -m tests.fuzz_paths --n 200 --seed 1
"""
from __future__ import annotations

import argparse
import sys

import numpy as np

import bammmotif2_amd as bm
from oracle import Oracle
from tests import test_mask_gpu, test_parity_gpu, test_seed_gpu


def random_spec(rng, i):
    K = int(rng.choice([0, 1, 2, 2, 3]))
    ss = bool(rng.integers(0, 2))
    W = int(rng.integers(3, 31))
    Lmax = int(2 ** rng.uniform(np.log2(W + 12), np.log2(1000)))   # the fp32 oracle's own noise outgrows these bodies' tolerances beyond ~2000 positions
    rag = int(rng.integers(0, max(1, Lmax // 4)))
    L0 = Lmax if ss else max((Lmax - 1) // 2, 1)
    L0 = max(L0, W + rag + 2)
    N = int(np.clip(40000 // (L0 * (1 if ss else 2)), 8, 300))
    return dict(name=f"p{i}", N=N, L0=L0, W=W, K=K, ss=ss, ragged=rag,
                n_frac=float(rng.choice([0.0, 0.001, 0.01])), seed=int(rng.integers(1, 1 << 30)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    orc = Oracle()
    orc.set_threads(1)
    ctx = bm.Context(0)
    bad = {}
    for i in range(args.n):
        spec = random_spec(rng, i)
        bodies = [("shapes", lambda: test_parity_gpu.test_shapes_and_length_buckets.__wrapped__(spec, ctx, orc)
                   if hasattr(test_parity_gpu.test_shapes_and_length_buckets, "__wrapped__")
                   else test_parity_gpu.test_shapes_and_length_buckets(spec, ctx, orc)),
                  ("seed", lambda: test_seed_gpu.test_seed_from_pwm_matches_oracle(spec, ctx, orc)),
                  ("mask", lambda: test_mask_gpu.test_mask_three_passes_match_oracle(
                      spec, float(rng.choice([0.05, 0.2])), bool(rng.integers(0, 2)), ctx, orc))]
        for name, body in bodies:
            try:
                body()
            except Exception as e:  # noqa: BLE001
                bad[name] = bad.get(name, 0) + 1
                print("FAIL", name, spec, "->", type(e).__name__, str(e).replace("\n", " ")[:260], flush=True)
        if (i + 1) % 20 == 0:
            print(f"[{i + 1}/{args.n}] failures so far: {bad}", flush=True)
    print("failures:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

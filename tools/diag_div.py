import sys, numpy as np
sys.path.insert(0, '.')
import torch
import bammmotif2_amd as bm
import oracle
from tests.cases import Case, SMALL_CASES
rng = np.random.RandomState(0)
a = rng.random_sample(1 << 20).astype(np.float32); b = (rng.random_sample(1 << 20).astype(np.float32) + 0.01)
g = (torch.from_numpy(a).cuda() / torch.from_numpy(b).cuda()).cpu().numpy()
print("torch div mismatches vs numpy:", int((g != a / b).sum()))
o = oracle.Oracle(); o.set_threads(1)
c = Case(**SMALL_CASES[0])
seq, kmer, off, vbg = c.encode(o)
ctx = bm.Context(0)
pk = bm.PackedSeqs.from_kmers(kmer, off); ss = bm.SeqSet(ctx, pk)
em = bm.EM(ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q)
print("v roundtrip equal:", np.array_equal(em.getV(), c.v0))
s_g = em.getS().reshape(64, c.W)
vK = c.v0[bm.v_offset(c.K, c.W):].reshape(64, c.W); bb = vbg[bm.bg_offset(2):]
s_np = vK / bb[:, None]
bad = np.argwhere(s_g != s_np)
print("mismatches", len(bad), "of", s_g.size)
for y, j in bad[:5]:
    print(y, j, vK[y, j], bb[y], s_g[y, j], s_np[y, j], (s_g[y, j] - s_np[y, j]) / np.spacing(s_np[y, j]))
# is it the divisor index?  try all bg entries
for y, j in bad[:3]:
    cand = [i for i in range(len(vbg)) if np.float32(vK[y, j] / vbg[i]) == s_g[y, j]]
    print("divisor candidates for", y, j, cand, "expected", 20 + y)

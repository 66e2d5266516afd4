import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bammmotif2_amd as bm, oracle
from tests.cases import Case
O = oracle.Oracle(); O.set_threads(1)
ctx = bm.Context(0)
for spec in [dict(name="a", N=8, L0=720, W=17, K=1, ragged=60), dict(name="f", N=8, L0=1545, W=17, K=1, ss=True), dict(name="g", N=8, L0=1700, W=17, K=1, ss=True), dict(name="h", N=8, L0=1900, W=17, K=1, ss=True), dict(name="i", N=8, L0=772, W=17, K=1), dict(name="j", N=8, L0=1545, W=17, K=1, ss=True, n_frac=0.01)]:
    c = Case(**spec)
    seq, kmer, off, vbg = c.encode(O)
    pk = bm.PackedSeqs.from_kmers(kmer, off); ss = bm.SeqSet(ctx, pk)
    em = bm.EM(ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q)
    em.EStep(); r = em.getR()
    s_o = O.linear_s(c.v0, vbg, c.K, c.W, 1); r_o, _ = O.estep(kmer, off, c.K, c.W, s_o, c.q)
    em.MStep()
    n = em.getCounts()[bm.v_offset(c.K, c.W):].reshape(16, c.W)
    n_o = O.mstep_counts(kmer, off, c.K, c.W, r_o)[bm.v_offset(c.K, c.W):].reshape(16, c.W)
    print(spec["name"], "L", int(off[1]-off[0]), "r ok", np.allclose(r, r_o, rtol=1e-4), "col sums gpu", np.round(n.sum(0), 3)[:8], "ref", np.round(n_o.sum(0), 3)[:8], "total", n.sum(), n_o.sum())

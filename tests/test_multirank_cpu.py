"""World-size-2 gloo test of the N>1 plan on CPU.

The HIP kernels need a GPU, so here the per-rank local accumulation is done by the oracle; what
is under test is the host logic that has to be right by construction on the 8-GPU box: the shard
ranges partition the set, every rank sums the same fused buffer [n_K | llh | sum_r | N], and the
redundant update gives every rank the single-process model."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    import bammmotif2_amd as bm
    from tests.cases import Case
    O = oracle.Oracle()
    O.set_threads(1)
    c = Case("mr", N=300, L0=80, W=10, K=2, ragged=20, n_frac=0.01)
    _, kmer, off, vbg = c.encode(O)
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    b, e = pk.shard_range(c.W, rank, world)
    lens = np.diff(off.astype(np.int64))
    sub_off = np.concatenate([[0], np.cumsum(lens[b:e])]).astype(np.uint64)
    sub_kmer = kmer[int(off[b]):int(off[e])]
    v, q = c.v0.copy(), c.q
    cells = 4 ** (c.K + 1) * c.W
    for it in range(3):
        s = O.linear_s(v, vbg, c.K, c.W, 2)
        r, llh = O.estep(sub_kmer, sub_off, c.K, c.W, s, q)
        n = O.mstep_counts(sub_kmer, sub_off, c.K, c.W, r)
        # the layout and number format bamm_em_reduce_buffer exposes: 64-bit integers, counts in units of
        # 2^-40, llh 2^-24, sum_r 2^-30, the sequence count as is -- summed across ranks as int64
        buf = np.zeros(cells + 3, np.int64)
        buf[:cells] = np.rint(n[bm.v_offset(c.K, c.W):].astype(np.float64) * 2.0 ** 40)
        buf[cells + 0] = np.rint(llh * 2.0 ** 24)
        buf[cells + 1] = np.rint(sum(float(r[int(sub_off[i]):int(sub_off[i + 1])].sum()) for i in range(e - b)) * 2.0 ** 30)
        buf[cells + 2] = e - b
        t = torch.from_numpy(buf)
        dist.all_reduce(t)                                   # one small collective per iteration
        nK = (t.numpy()[:cells] * 2.0 ** -40).astype(np.float32)
        n_all = np.zeros(bm.v_size(c.K, c.W), np.float32)
        n_all[bm.v_offset(c.K, c.W):] = nK
        for k in range(c.K, 0, -1):                          # EM.cpp:247-254
            hi = n_all[bm.v_offset(k, c.W):bm.v_offset(k + 1, c.W)].reshape(4, 4 ** k, c.W)
            n_all[bm.v_offset(k - 1, c.W):bm.v_offset(k, c.W)] = hi.sum(axis=0).ravel()
        v = O.update_v(n_all, c.A, vbg, c.K, c.W)
        N_glob, sum_r = float(t.numpy()[cells + 2]), t.numpy()[cells + 1] * 2.0 ** -30
        q = float(np.float32((N_glob - sum_r + 1.0) / (N_glob + 2.0)))
    np.save(os.path.join(out_dir, f"v_{rank}.npy"), v)
    np.save(os.path.join(out_dir, f"range_{rank}.npy"), np.array([b, e, q]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_equal_single_process(tmp_path, orc):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    v0, v1 = np.load(tmp_path / "v_0.npy"), np.load(tmp_path / "v_1.npy")
    r0, r1 = np.load(tmp_path / "range_0.npy"), np.load(tmp_path / "range_1.npy")
    assert r0[0] == 0 and r0[1] == r1[0] and r1[1] == 300          # shards partition the set
    assert np.array_equal(v0, v1)                                  # redundant update is identical
    from tests.cases import Case
    c = Case("mr", N=300, L0=80, W=10, K=2, ragged=20, n_frac=0.01)
    _, kmer, off, vbg = c.encode(orc)
    res = orc.optimize(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q, optimizeQ=True,
                       epsilon=0.0, max_iter=3)
    np.testing.assert_allclose(v0, res["v"], rtol=1e-5, atol=1e-9)   # fp32 summation-order noise
    np.testing.assert_allclose(r0[2], res["q"], rtol=2e-5)

"""Host-side mirror of the reference's EM / ScoreSeqSet interface over the C ABI.

Method names follow the reference (`EStep`, `MStep`, `optimize`, `optimize_q`, `getR`, `getQ`;
/root/reference/src/refinement/EM.h:20-36, seq_scoring/ScoreSeqSet.h:26-36) so that the parity
tests read like calls into the reference.  Everything numeric happens in libbamm_em.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import numpy as np

from . import abi
from .abi import check


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def v_size(K: int, W: int) -> int:
    return W * ((4 ** (K + 2) - 4) // 3)


def v_offset(k: int, W: int) -> int:
    return W * ((4 ** (k + 1) - 4) // 3)


def bg_size(K: int) -> int:
    return (4 ** (K + 2) - 4) // 3


def bg_offset(k: int) -> int:
    return (4 ** (k + 1) - 4) // 3


def libc_srand(seed: int = 42) -> None:
    """mainBaMM.cpp:22 -- the reference seeds libc's rand() once before reading sequences."""
    C.CDLL(None).srand(C.c_uint(seed))


class PackedSeqs:
    """Host-resident 2-bit packed sequence set (+ N exceptions)."""

    def __init__(self, ptr):
        self._p = ptr
        self.lib = abi.load()

    @classmethod
    def from_kmers(cls, kmer, off) -> "PackedSeqs":
        lib = abi.load()
        off = _u64(off)
        out = C.POINTER(abi.Packed)()
        check(lib.bamm_pack_kmers(_u64(kmer), off, len(off) - 1, C.byref(out)))
        return cls(out)

    @classmethod
    def from_codes(cls, codes, off, single_strand: bool = False, seed: Optional[int] = 42) -> "PackedSeqs":
        lib = abi.load()
        off = _u64(off)
        out = C.POINTER(abi.Packed)()
        if seed is not None:      # the libc stream starts at srand(seed): the draws may be taken on all host threads
            check(lib.bamm_pack_codes_seeded(np.ascontiguousarray(codes, np.uint8), off, len(off) - 1,
                                             int(single_strand), seed, C.byref(out)))
        else:                     # wherever the caller's libc stream stands
            check(lib.bamm_pack_codes(np.ascontiguousarray(codes, np.uint8), off, len(off) - 1,
                                      int(single_strand), C.byref(out)))
        return cls(out)

    @property
    def c(self):
        return self._p.contents

    @property
    def n_seqs(self) -> int:
        return int(self.c.n_seqs)

    @property
    def total_len(self) -> int:
        return int(self.c.total_len)

    @property
    def lengths(self) -> np.ndarray:
        return np.ctypeslib.as_array(self.c.len, (max(self.n_seqs, 1),))[: self.n_seqs].copy()

    @property
    def n_exceptions(self) -> int:
        return int(self.c.n_exc)

    @property
    def words(self) -> np.ndarray:
        return np.ctypeslib.as_array(self.c.words, (max(int(self.c.n_words), 1),))[: int(self.c.n_words)].copy()

    def arrays(self) -> dict:
        """Every array of the packed set (copies): words, word_off, len, exc_off, exc_pos, exc_kmer, exc_clean."""
        c, n, ne = self.c, self.n_seqs, int(self.c.n_exc)
        view = lambda ptr, count: np.ctypeslib.as_array(ptr, (max(count, 1),))[:count].copy()
        return dict(words=view(c.words, int(c.n_words)), word_off=view(c.word_off, n + 1), len=view(c.len, n),
                    exc_off=view(c.exc_off, n + 1), exc_pos=view(c.exc_pos, ne), exc_kmer=view(c.exc_kmer, ne),
                    exc_clean=view(c.exc_clean, ne), total_len=int(c.total_len), max_len=int(c.max_len), min_len=int(c.min_len))

    def offsets(self) -> np.ndarray:
        return np.concatenate([[0], np.cumsum(self.lengths.astype(np.int64))]).astype(np.uint64)

    def unpack_y(self, K: int) -> np.ndarray:
        out = np.zeros(max(self.total_len, 1), np.uint32)
        check(self.lib.bamm_unpack_y(self._p, K, out))
        return out[: self.total_len]

    def bg_model(self, K: int, alpha) -> np.ndarray:
        """BackgroundModel (BackgroundModel.cpp:3-46,441-473) learned from this set."""
        out = np.zeros(bg_size(K), np.float32)
        alpha = _f32(alpha)
        assert len(alpha) >= K + 1
        check(self.lib.bamm_bg_model(self._p, K, alpha, out))
        return out

    def shard_range(self, W: int, rank: int, world: int):
        b, e = C.c_uint64(0), C.c_uint64(0)
        lens = np.ascontiguousarray(self.lengths, np.uint32)
        if len(lens) == 0:
            lens = np.zeros(1, np.uint32)
        check(self.lib.bamm_shard_range(lens, self.n_seqs, W, rank, world, C.byref(b), C.byref(e)))
        return int(b.value), int(e.value)

    def free(self):
        if self._p:
            self.lib.bamm_packed_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def device_count() -> int:
    """HIP devices visible to this process (raises without one: there is no CPU fallback)."""
    n = C.c_int()
    check(abi.load().bamm_device_count(C.byref(n)))
    return int(n.value)


def device_pci_bus_id(device: int) -> str:
    buf = C.create_string_buffer(32)
    check(abi.load().bamm_device_pci_bus_id(int(device), buf, 32))
    return buf.value.decode()


def device_can_access_peer(device: int, peer: int) -> bool:
    can = C.c_int(0)
    check(abi.load().bamm_device_can_access_peer(int(device), int(peer), C.byref(can)))
    return bool(can.value)


def peer_access_matrix():
    """[d][p] = device d can address device p's memory (hipDeviceCanAccessPeer), over every visible device."""
    n = device_count()
    return [[device_can_access_peer(d, p) for p in range(n)] for d in range(n)]


class Context:
    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.lib = abi.load()
        h = C.c_void_p()
        check(self.lib.bamm_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h)))
        self.h = h
        self.device = device

    def sync(self):
        check(self.lib.bamm_ctx_sync(self.h))

    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        check(self.lib.bamm_ctx_device_name(self.h, buf, 256))
        return buf.value.decode()

    def set_launch(self, blocks: int = 0, threads: int = 0):
        check(self.lib.bamm_ctx_set_launch(self.h, blocks, threads))

    def set_tuning(self, **kv):
        """Kernel-selection switches for EM handles created afterwards (include/bamm_em.h:
        grouped, group_size, group_layout, sparse, e_fused)."""
        for k, v in kv.items():
            check(self.lib.bamm_ctx_set_tuning(self.h, k.encode(), int(v)))

    def close(self):
        if self.h:
            self.lib.bamm_ctx_destroy(self.h)
            self.h = None


COMM_ID_BYTES = 128


class Comm:
    """One rank of an RCCL communicator bound to a Context (include/bamm_em.h: bamm_comm_*)."""

    def __init__(self, ctx: Context, handle):
        self.ctx, self.lib, self.h = ctx, ctx.lib, handle

    @classmethod
    def init_all(cls, ctxs):
        """One process driving len(ctxs) distinct devices: ncclCommInitAll."""
        lib = ctxs[0].lib
        arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        out = (C.c_void_p * len(ctxs))()
        check(lib.bamm_comm_init_all(arr, len(ctxs), out))
        return [cls(c, C.c_void_p(h)) for c, h in zip(ctxs, out)]

    @classmethod
    def init_local(cls, ctxs, max_words: int):
        """One process, len(ctxs) contexts on any devices (even one): host-staged sum, no RCCL (self-tests)."""
        lib = ctxs[0].lib
        arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        out = (C.c_void_p * len(ctxs))()
        check(lib.bamm_comm_init_local(arr, len(ctxs), max_words, out))
        return [cls(c, C.c_void_p(h)) for c, h in zip(ctxs, out)]

    @classmethod
    def init_shm(cls, ctx: Context, name: str, rank: int, world: int, max_words: int):
        """One PROCESS per rank on one host, the sum staged through the POSIX shared-memory segment `name` ("/..."): no RCCL
        (self-tests of the cross-process paths on a 1-GPU box)."""
        h = C.c_void_p()
        check(ctx.lib.bamm_comm_init_shm(ctx.h, name.encode(), rank, world, max_words, C.byref(h)))
        return cls(ctx, h)

    def abort(self):
        """Wake the peers blocked in a collective with this rank (they get BAMM_ERR_COMM)."""
        if self.h:
            self.lib.bamm_comm_abort(self.h)

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        check(abi.load().bamm_comm_unique_id(buf, COMM_ID_BYTES))
        return buf.raw

    @classmethod
    def init_rank(cls, ctx: Context, uid: bytes, rank: int, world: int):
        """One process per device: `uid` from rank 0's unique_id(), carried by the launcher."""
        assert len(uid) == COMM_ID_BYTES
        h = C.c_void_p()
        check(ctx.lib.bamm_comm_init_rank(ctx.h, C.c_char_p(uid), rank, world, C.byref(h)))
        return cls(ctx, h)

    def info(self):
        r, w, v = C.c_uint32(), C.c_uint32(), C.c_int()
        check(self.lib.bamm_comm_info(self.h, C.byref(r), C.byref(w), C.byref(v)))
        return dict(rank=r.value, world=w.value, rccl_version=v.value)

    def time_allreduce(self, n_words: int, iters: int = 200) -> float:
        """Microseconds per bare all-reduce of n_words int64 words (collective: every rank calls it)."""
        us = C.c_float()
        check(self.lib.bamm_comm_time_allreduce(self.h, n_words, iters, C.byref(us)))
        return float(us.value)

    def close(self):
        if self.h:
            self.lib.bamm_comm_destroy(self.h)
            self.h = None


class SeqSet:
    """Sequences resident in HBM (a shard [begin, end) of a PackedSeqs)."""

    def __init__(self, ctx: Context, packed: PackedSeqs, begin: int = 0, end: Optional[int] = None):
        self.ctx, self.lib = ctx, ctx.lib
        end = packed.n_seqs if end is None else end
        h = C.c_void_p()
        check(self.lib.bamm_seqs_upload(ctx.h, packed._p, begin, end, C.byref(h)))
        self.h = h
        self.n_seqs = end - begin
        self.lengths = packed.lengths[begin:end]
        self.off = np.concatenate([[0], np.cumsum(self.lengths.astype(np.int64))]).astype(np.uint64)

    @classmethod
    def from_codes(cls, ctx: "Context", codes, off, single_strand: bool = False, seed: int = 42, resident: bool = True):
        """Sequence::Sequence on the device (include/bamm_em.h: bamm_seqs_from_codes): returns (PackedSeqs, SeqSet) -- the
        packed set bamm_pack_codes_seeded would have built on the host, and the resident set (None with resident=False)."""
        off = _u64(off)
        pk, h = C.POINTER(abi.Packed)(), C.c_void_p()
        check(ctx.lib.bamm_seqs_from_codes(ctx.h, np.ascontiguousarray(codes, np.uint8), off, len(off) - 1, int(single_strand), seed,
                                           C.byref(pk), C.byref(h) if resident else None))
        packed = PackedSeqs(pk)
        if not resident:
            return packed, None
        self = cls.__new__(cls)
        self.ctx, self.lib, self.h = ctx, ctx.lib, h
        self.n_seqs = packed.n_seqs
        self.lengths = packed.lengths
        self.off = np.concatenate([[0], np.cumsum(self.lengths.astype(np.int64))]).astype(np.uint64)
        return packed, self

    def bg_model(self, K: int, alpha) -> np.ndarray:
        """BackgroundModel learned from the resident set, its counting pass on the device (bamm_seqs_bg_model)."""
        out = np.zeros(bg_size(K), np.float32)
        alpha = _f32(alpha)
        assert len(alpha) >= K + 1
        check(self.lib.bamm_seqs_bg_model(self.ctx.h, self.h, K, alpha, out))
        return out

    def info(self):
        n, t, m, b = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint64()
        check(self.lib.bamm_seqs_info(self.h, C.byref(n), C.byref(t), C.byref(m), C.byref(b)))
        return dict(n_seqs=n.value, total_len=t.value, max_len=m.value, hbm_bytes=b.value)

    def close(self):
        if self.h:
            self.lib.bamm_seqs_destroy(self.h)
            self.h = None


class EM:
    """Drop-in for the reference's `EM` (EM.h:11-69) on one GPU shard."""

    def __init__(self, ctx: Context, seqs: SeqSet, K: int, W: int, vbg, A, v_init, q: float,
                 bg_order: int = 2, optimizeQ: bool = False, mask=None, epsilon: float = 0.01,
                 max_iterations: int = 1000, n_seqs_global: int = 0, n_seqs_bound: int = 0):
        self.ctx, self.lib, self.seqs = ctx, ctx.lib, seqs
        self.K, self.W = K, W
        prm = abi.EmParams()
        self.lib.bamm_em_default_params(C.byref(prm))
        prm.K, prm.W, prm.bg_order, prm.q = K, W, bg_order, q
        prm.optimize_q = int(optimizeQ)
        prm.epsilon, prm.max_iterations, prm.n_seqs_global = epsilon, max_iterations, n_seqs_global
        prm.n_seqs_bound = n_seqs_bound
        self.max_iterations = max_iterations
        vbg, A, v_init = _f32(vbg), _f32(A), _f32(v_init)
        assert len(vbg) >= bg_size(bg_order) and len(A) == (K + 1) * W and len(v_init) == v_size(K, W)
        mptr = None
        if mask is not None:
            self._mask = np.ascontiguousarray(mask, np.uint8)
            assert len(self._mask) == seqs.n_seqs
            mptr = self._mask.ctypes.data_as(C.c_void_p)
        h = C.c_void_p()
        check(self.lib.bamm_em_create(ctx.h, seqs.h, C.byref(prm), vbg, A, v_init, mptr, C.byref(h)))
        self.h = h
        self._cb = None

    # -- the reference's public surface ---------------------------------------------------
    def EStep(self):
        check(self.lib.bamm_em_estep(self.h))

    def MStep(self):
        check(self.lib.bamm_em_mstep(self.h))

    def optimize_q(self):
        check(self.lib.bamm_em_optimize_q(self.h))

    def optimize(self) -> int:
        it = C.c_uint32()
        check(self.lib.bamm_em_optimize(self.h, C.byref(it)))
        return int(it.value)

    def mask(self, f: float = 0.05) -> int:
        """EM::mask (EM.cpp:261-503, --advanceEM); the cut-off and list size land in last_mask."""
        it, cut, listed = C.c_uint32(), C.c_float(), C.c_uint64()
        check(self.lib.bamm_em_mask(self.h, f, C.byref(it), C.byref(cut), C.byref(listed)))
        self.last_mask = dict(cutoff=float(cut.value), listed=int(listed.value))
        return int(it.value)

    def getQ(self) -> float:
        q = C.c_float()
        check(self.lib.bamm_em_get_q(self.h, C.byref(q)))
        return float(q.value)

    def getR(self, begin: int = 0, end: Optional[int] = None) -> np.ndarray:
        end = self.seqs.n_seqs if end is None else end
        total = int(self.seqs.off[end] - self.seqs.off[begin])
        out = np.zeros(max(total, 1), np.float32)
        check(self.lib.bamm_em_get_r(self.h, begin, end, out, total))
        return out[:total]

    # -- extensions -------------------------------------------------------------------------
    def iterate(self, n: int = 1):
        check(self.lib.bamm_em_iterate(self.h, n))

    def accumulate(self):
        check(self.lib.bamm_em_accumulate(self.h))

    def update(self):
        check(self.lib.bamm_em_update(self.h))

    def reduce_buffer(self):
        p, n = C.c_void_p(), C.c_uint64()
        check(self.lib.bamm_em_reduce_buffer(self.h, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    def set_reduce_buffer(self, dev_ptr: int, n_words: int):
        """Use caller-owned device memory (e.g. a torch int64 tensor) as the fused accumulator."""
        check(self.lib.bamm_em_set_reduce_buffer(self.h, C.c_void_p(dev_ptr), n_words))

    def set_allreduce(self, fn: Optional[Callable[[int, int, int], int]]):
        """fn(dev_ptr, n_words, hip_stream) -> 0 on success: sum n_words int64 words across ranks;
        called between the local accumulation and the model update of every pass."""
        if fn is None:
            self._cb = abi.ALLREDUCE_FN(0)
        else:
            def tramp(_user, ptr, n, stream):
                try:
                    return int(fn(int(ptr or 0), int(n), int(stream or 0)) or 0)
                except Exception:  # never unwind through C
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = abi.ALLREDUCE_FN(tramp)
        check(self.lib.bamm_em_set_allreduce(self.h, self._cb, None))

    def set_comm(self, comm: Optional["Comm"]):
        """Native path: one ncclAllReduce(int64, sum) of the accumulator per pass, on the context's stream."""
        check(self.lib.bamm_em_set_comm(self.h, comm.h if comm is not None else None))
        self._comm = comm

    def comm_mode(self):
        """(mode, note): 0 = no reduction over ranks, 1 = one collective per pass, 2 = inside the sequence kernels over
        peer-mapped inboxes (Context.set_tuning(peer_allreduce=1) on every rank); collective on first use."""
        m = C.c_int()
        buf = C.create_string_buffer(512)
        check(self.lib.bamm_em_comm_mode(self.h, C.byref(m), buf, 512))
        return int(m.value), buf.value.decode()

    def getV(self) -> np.ndarray:
        out = np.zeros(v_size(self.K, self.W), np.float32)
        check(self.lib.bamm_em_get_v(self.h, out))
        return out

    def getCounts(self) -> np.ndarray:
        out = np.zeros(v_size(self.K, self.W), np.float32)
        check(self.lib.bamm_em_get_counts(self.h, out))
        return out

    def getS(self) -> np.ndarray:
        out = np.zeros(4 ** (self.K + 1) * self.W, np.float32)
        check(self.lib.bamm_em_get_s(self.h, out))
        return out

    def getLLH(self) -> float:
        x = C.c_float()
        check(self.lib.bamm_em_get_llh(self.h, C.byref(x)))
        return float(x.value)

    def getVdiff(self) -> float:
        x = C.c_float()
        check(self.lib.bamm_em_get_vdiff(self.h, C.byref(x)))
        return float(x.value)

    def iteration(self) -> int:
        x = C.c_uint32()
        check(self.lib.bamm_em_get_iteration(self.h, C.byref(x)))
        return int(x.value)

    def trace(self):
        cap = self.max_iterations
        llh, vd, q = (np.zeros(cap, np.float32) for _ in range(3))
        n = C.c_uint32()
        check(self.lib.bamm_em_get_trace(self.h, llh, vd, q, cap, C.byref(n)))
        m = min(int(n.value), cap)
        return llh[:m].copy(), vd[:m].copy(), q[:m].copy()

    TIMING_WHOLE_CALL = 0xFFFFFFFF                           # BAMM_TIMING_WHOLE_CALL

    def set_kernel_timing(self, every: int):
        """Time passes 0, every, 2*every, ... of each call with a pair of HIP events each (0 = none, 1 = all; a pair costs
        7-8 us of stream time); -1 / TIMING_WHOLE_CALL: one pair around all passes of a call (gaps included, no cost per pass)."""
        check(self.lib.bamm_em_set_kernel_timing(self.h, self.TIMING_WHOLE_CALL if every < 0 else every))

    def kernel_time(self):
        ms, n = C.c_float(), C.c_uint32()
        check(self.lib.bamm_em_kernel_time(self.h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def plan(self):
        """(sequences through the grouped-column kernel, through the per-column kernel, launches per pass)."""
        g, o, n = C.c_uint64(), C.c_uint64(), C.c_uint32()
        check(self.lib.bamm_em_plan(self.h, C.byref(g), C.byref(o), C.byref(n)))
        return int(g.value), int(o.value), int(n.value)

    def plan_mixed(self) -> int:
        """Sequences (of the grouped ones) that go through the mixed-row kernel (csrc/mixed_kernel.h)."""
        m = C.c_uint64()
        check(self.lib.bamm_em_plan_mixed(self.h, C.byref(m)))
        return int(m.value)

    def close(self):
        if self.h:
            self.lib.bamm_em_destroy(self.h)
            self.h = None


def sample_negatives(ctx: Context, positives: SeqSet, s_order: int = 2, m_fold: int = 1, generic: bool = False,
                     keep_stride: int = 0, resident: bool = True):
    """SeqGenerator's negative sampler on the device (include/bamm_em.h: bamm_sample_negatives): (PackedSeqs, SeqSet or
    None) -- m_fold negatives per resident positive, the reference's own, base for base."""
    pk, h = C.POINTER(abi.Packed)(), C.c_void_p()
    check(ctx.lib.bamm_sample_negatives(ctx.h, positives.h, s_order, m_fold, int(generic), keep_stride, C.byref(pk),
                                        C.byref(h) if resident else None))
    packed = PackedSeqs(pk)
    if not resident:
        return packed, None
    neg = SeqSet.__new__(SeqSet)
    neg.ctx, neg.lib, neg.h = ctx, ctx.lib, h
    neg.n_seqs = packed.n_seqs
    neg.lengths = packed.lengths
    neg.off = np.concatenate([[0], np.cumsum(neg.lengths.astype(np.int64))]).astype(np.uint64)
    return packed, neg


def seed_from_pwm(ctx: Context, seqs: SeqSet, K: int, W: int, score, q: float, u):
    """The pass over the sequences of Motif::initFromPWM (Motif.cpp:228-311) on the device: returns
    (counts[v_size(K,W)] int32, z[n_seqs] uint32).  score: [4][W] floored PWM / 0th-order background;
    u[n]: the uniform variate of sequence n's draw (see include/bamm_em.h)."""
    score = _f32(score)
    u = np.ascontiguousarray(u, np.float64)
    assert len(score) == 4 * W and len(u) == seqs.n_seqs
    counts = np.zeros(v_size(K, W), np.int32)
    z = np.zeros(seqs.n_seqs, np.uint32)
    check(ctx.lib.bamm_seed_from_pwm(ctx.h, seqs.h, K, W, score, q, u.ctypes.data_as(C.c_void_p),
                                     counts.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p)))
    return counts, z


def logodds(ctx: Context, seqs: SeqSet, K: int, W: int, bg_order: int, v, vbg, want_mops: bool = True, mask=None):
    """ScoreSeqSet::calcLogOdds (ScoreSeqSet.cpp:25-67): returns (mops or None, zoops, z).

    mask: optional per-sequence bytes; sequences with 0 are skipped and report zeros."""
    lib = ctx.lib
    N = seqs.n_seqs
    total = int((seqs.lengths.astype(np.int64) - W + 1).sum()) if N else 0
    mops = np.zeros(max(total, 1), np.float32) if want_mops else None
    zoops = np.zeros(max(N, 1), np.float32)
    z = np.zeros(max(N, 1), np.uint64)
    mptr = mops.ctypes.data_as(C.c_void_p) if want_mops else None
    if mask is None:
        check(lib.bamm_logodds(ctx.h, seqs.h, K, W, bg_order, _f32(v), _f32(vbg), mptr, total, zoops, z))
    else:
        mk = np.ascontiguousarray(mask, np.uint8)
        assert len(mk) == N
        check(lib.bamm_logodds_subset(ctx.h, seqs.h, mk.ctypes.data_as(C.c_void_p), K, W, bg_order, _f32(v), _f32(vbg),
                                      mptr, total, zoops, z))
    return (mops[:total] if want_mops else None), zoops[:N], z[:N]


def calculate_p(v, vbg, bg_order: int, K: int, W: int) -> np.ndarray:
    lib = abi.load()
    p = np.zeros(v_size(K, W), np.float32)
    check(lib.bamm_calculate_p(_f32(v), _f32(vbg), bg_order, K, W, p))
    return p

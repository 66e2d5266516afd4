"""TEST INFRASTRUCTURE ONLY.

ctypes loaders + numpy wrappers for

* ``oracle/libbamm_oracle.so``  -- our plain-C restatement (``bamm_oracle.c``), class ``Oracle``
* ``oracle/_ref/libbammref.so`` -- the real reference translation units driven by
  ``ref_harness.cpp``, class ``Reference`` (only present where it was built).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker.  Nothing under ``bammmotif2_amd/`` imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libbamm_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libbammref.so")

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(ref: bool = True) -> None:
    """Compile the C restatement (always) and the reference build (when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])
        pkg_lib = os.path.join(os.path.dirname(HERE), "bammmotif2_amd", "libbamm_em.so")
        if os.path.exists(pkg_lib):                       # the reference's classes over libbamm_em (integration/)
            subprocess.check_call(["make", "-s", "-C", HERE, "ref_hip"])


def have_reference() -> bool:
    return os.path.exists(REF_SO)


def v_offset(k: int, W: int) -> int:
    return W * ((4 ** (k + 1) - 4) // 3)


def v_size(K: int, W: int) -> int:
    return v_offset(K + 1, W)


def bg_offset(k: int) -> int:
    return (4 ** (k + 1) - 4) // 3


def bg_size(K: int) -> int:
    return bg_offset(K + 1)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class Oracle:
    """numpy front-end of bamm_oracle.h (see that header for the reference citations)."""

    def __init__(self, path: str = ORACLE_SO):
        if not os.path.exists(path):
            build(ref=False)
        L = self.L = C.CDLL(path)
        sz, f, i = C.c_size_t, C.c_float, C.c_int
        L.orc_set_threads.argtypes = [i]
        L.orc_encode_set.argtypes = [_u8p, _u64p, sz, i, i, C.c_uint, _u8p, _u64p, _u64p]
        L.orc_bg_model.argtypes = [_u64p, _u64p, sz, sz, _f32p, _f32p]
        L.orc_linear_s.argtypes = [_f32p, _f32p, sz, sz, sz, _f32p]
        L.orc_log_s.argtypes = [_f32p, _f32p, sz, sz, sz, _f32p]
        L.orc_estep.argtypes = [_u64p, _u64p, sz, sz, sz, _f32p, f, _f32p]
        L.orc_estep.restype = f
        L.orc_mstep_counts.argtypes = [_u64p, _u64p, sz, sz, sz, _f32p, _f32p]
        L.orc_update_v.argtypes = [_f32p, _f32p, _f32p, sz, sz, _f32p]
        L.orc_optimize_q.argtypes = [_f32p, _u64p, sz, sz]
        L.orc_optimize_q.restype = f
        L.orc_calculate_p.argtypes = [_f32p, _f32p, sz, sz, sz, _f32p]
        L.orc_optimize.argtypes = [_u64p, _u64p, sz, sz, sz, sz, _f32p, _f32p, _f32p,
                                   C.POINTER(f), i, f, sz, _f32p, _f32p, _f32p, _f32p, C.POINTER(f)]
        L.orc_optimize.restype = sz
        L.orc_mask.argtypes = [_u64p, _u64p, sz, sz, sz, sz, _f32p, _f32p, _f32p, C.POINTER(f), i, f, f, sz,
                               _f32p, _f32p, _f32p, _f32p, C.POINTER(f), C.POINTER(f), C.POINTER(C.c_uint64)]
        L.orc_mask.restype = sz
        L.orc_logodds.argtypes = [_u64p, _u64p, sz, sz, sz, _f32p, _f32p, _f32p, _u64p]
        L.orc_init_from_pwm.argtypes = [_f32p, sz, sz, _f32p, _f32p, _u64p, _u64p, sz, f, _f32p]
        L.orc_init_from_pwm_sites.argtypes = [_f32p, sz, sz, _f32p, _f32p, _u64p, _u64p, sz, f, _f32p, C.c_void_p, C.c_void_p]
        L.orc_em_step_f64.argtypes = [_u64p, _u64p, sz, sz, sz, sz, _f32p, _f32p, _f32p, f,
                                      _f32p, _f32p, C.POINTER(C.c_double), C.POINTER(C.c_double)]

    def set_threads(self, n: int) -> None:
        self.L.orc_set_threads(int(n))

    def encode_set(self, codes, in_off, single_strand=False, seed=42, do_srand=True):
        codes = np.ascontiguousarray(codes, np.uint8)
        in_off = _u64(in_off)
        N = len(in_off) - 1
        lens = np.diff(in_off.astype(np.int64))
        out_len = lens if single_strand else 2 * lens + 1
        total = int(out_len.sum())
        seq = np.zeros(total, np.uint8)
        kmer = np.zeros(total, np.uint64)
        off = np.zeros(N + 1, np.uint64)
        self.L.orc_encode_set(codes, in_off, N, int(single_strand), int(do_srand), seed, seq, kmer, off)
        return seq, kmer, off

    def bg_model(self, kmer, off, K, alpha):
        out = np.zeros(bg_size(K), np.float32)
        self.L.orc_bg_model(_u64(kmer), _u64(off), len(off) - 1, K, _f32(alpha), out)
        return out

    def linear_s(self, v, vbg, K, W, K_bg):
        s = np.zeros(4 ** (K + 1) * W, np.float32)
        self.L.orc_linear_s(_f32(v), _f32(vbg), K, W, K_bg, s)
        return s

    def log_s(self, v, vbg, K, W, K_bg):
        s = np.zeros(4 ** (K + 1) * W, np.float32)
        self.L.orc_log_s(_f32(v), _f32(vbg), K, W, K_bg, s)
        return s

    def estep(self, kmer, off, K, W, s, q):
        off = _u64(off)
        r = np.zeros(int(off[-1]), np.float32)
        llh = self.L.orc_estep(_u64(kmer), off, len(off) - 1, K, W, _f32(s), q, r)
        return r, float(llh)

    def mstep_counts(self, kmer, off, K, W, r):
        n = np.zeros(v_size(K, W), np.float32)
        self.L.orc_mstep_counts(_u64(kmer), _u64(off), len(off) - 1, K, W, _f32(r), n)
        return n

    def update_v(self, n, A, vbg, K, W):
        v = np.zeros(v_size(K, W), np.float32)
        self.L.orc_update_v(_f32(n), _f32(A), _f32(vbg), K, W, v)
        return v

    def optimize_q(self, r, off, W):
        return float(self.L.orc_optimize_q(_f32(r), _u64(off), len(off) - 1, W))

    def calculate_p(self, v, vbg, k_bg, K, W):
        p = np.zeros(v_size(K, W), np.float32)
        self.L.orc_calculate_p(_f32(v), _f32(vbg), k_bg, K, W, p)
        return p

    def optimize(self, kmer, off, K, W, bg_order, vbg, A, v0, q, optimizeQ=False,
                 epsilon=0.01, max_iter=1000):
        off = _u64(off)
        v = _f32(v0).copy()
        qv = C.c_float(q)
        llh = C.c_float(0)
        r = np.zeros(int(off[-1]), np.float32)
        n = np.zeros(v_size(K, W), np.float32)
        tl = np.zeros(max_iter, np.float32)
        tv = np.zeros(max_iter, np.float32)
        it = self.L.orc_optimize(_u64(kmer), off, len(off) - 1, K, W, bg_order, _f32(vbg), _f32(A), v,
                                 C.byref(qv), int(optimizeQ), epsilon, max_iter, r, n, tl, tv, C.byref(llh))
        return dict(iterations=int(it), v=v, q=float(qv.value), llh=float(llh.value), r=r, n=n,
                    trace_llh=tl[:it].copy(), trace_vdiff=tv[:it].copy())

    def mask(self, kmer, off, K, W, bg_order, vbg, A, v0, q, optimizeQ=False, f=0.05,
             epsilon=0.01, max_iter=1000):
        off = _u64(off)
        v = _f32(v0).copy()
        qv, llh, cut = C.c_float(q), C.c_float(0), C.c_float(0)
        listed = C.c_uint64(0)
        r = np.zeros(int(off[-1]), np.float32)
        n = np.zeros(v_size(K, W), np.float32)
        tl = np.zeros(max_iter, np.float32)
        tv = np.zeros(max_iter, np.float32)
        it = self.L.orc_mask(_u64(kmer), off, len(off) - 1, K, W, bg_order, _f32(vbg), _f32(A), v,
                             C.byref(qv), int(optimizeQ), f, epsilon, max_iter, r, n, tl, tv, C.byref(llh),
                             C.byref(cut), C.byref(listed))
        return dict(iterations=int(it), v=v, q=float(qv.value), llh=float(llh.value), r=r, n=n,
                    trace_llh=tl[:it].copy(), trace_vdiff=tv[:it].copy(), cutoff=float(cut.value),
                    listed=int(listed.value))

    def logodds(self, kmer, off, K, W, s_log):
        off = _u64(off)
        N = len(off) - 1
        lens = np.diff(off.astype(np.int64))
        mops = np.zeros(int((lens - W + 1).sum()), np.float32)
        zoops = np.zeros(N, np.float32)
        z = np.zeros(N, np.uint64)
        self.L.orc_logodds(_u64(kmer), off, N, K, W, _f32(s_log), mops, zoops, z)
        return mops, zoops, z

    def init_from_pwm(self, pwm, W, K, A, vbg, kmer, off, q):
        v = np.zeros(v_size(K, W), np.float32)
        self.L.orc_init_from_pwm(_f32(pwm), W, K, _f32(A), _f32(vbg), _u64(kmer), _u64(off), len(off) - 1, q, v)
        return v

    def init_from_pwm_sites(self, pwm, W, K, A, vbg, kmer, off, q):
        """(v, z, counts): the seed model, the sampled site per sequence (0 = none, i = window i-1) and the
        integer site counts of all orders."""
        v = np.zeros(v_size(K, W), np.float32)
        z = np.zeros(max(len(off) - 1, 1), np.uint32)
        cnt = np.zeros(v_size(K, W), np.int32)
        self.L.orc_init_from_pwm_sites(_f32(pwm), W, K, _f32(A), _f32(vbg), _u64(kmer), _u64(off), len(off) - 1, q, v,
                                       z.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p))
        return v, z[: len(off) - 1], cnt

    def em_step_f64(self, kmer, off, K, W, bg_order, vbg, A, v, q):
        v_out = np.zeros(v_size(K, W), np.float32)
        n_out = np.zeros(v_size(K, W), np.float32)
        llh, sr = C.c_double(0), C.c_double(0)
        self.L.orc_em_step_f64(_u64(kmer), _u64(off), len(off) - 1, K, W, bg_order, _f32(vbg), _f32(A),
                               _f32(v), q, v_out, n_out, C.byref(llh), C.byref(sr))
        return v_out, n_out, llh.value, sr.value


class Reference:
    """Thin front-end of ref_harness.cpp: the reference's own classes, run in process."""

    def __init__(self, path: str = REF_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        # ctypes forces RTLD_NOW, which would fail on the one symbol the partial reference build
        # leaves undefined (SequenceSet::getSequences, only reachable from Motif::initFromPWM,
        # which the harness never calls).  dlopen lazily ourselves and hand ctypes the handle.
        libc = C.CDLL(None)
        libc.dlopen.restype = C.c_void_p
        libc.dlopen.argtypes = [C.c_char_p, C.c_int]
        handle = libc.dlopen(path.encode(), os.RTLD_LAZY | os.RTLD_LOCAL)
        if not handle:
            raise OSError("dlopen(RTLD_LAZY) failed for " + path)
        R = self.R = C.CDLL(path, handle=handle)
        vp, u64, f, i = C.c_void_p, C.c_uint64, C.c_float, C.c_int
        R.ref_set_threads.argtypes = [i]
        R.ref_session_create.argtypes = [_u8p, _u64p, u64, i, i, C.c_uint]
        R.ref_session_create.restype = vp
        R.ref_session_destroy.argtypes = [vp]
        R.ref_seq_L.argtypes = [vp, u64]
        R.ref_seq_L.restype = u64
        R.ref_seq_kmer.argtypes = [vp, u64]
        R.ref_seq_kmer.restype = C.POINTER(C.c_uint64)
        R.ref_seq_codes.argtypes = [vp, u64]
        R.ref_seq_codes.restype = C.POINTER(C.c_uint8)
        R.ref_bg_create.argtypes = [vp, u64, _f32p]
        R.ref_bg_create.restype = vp
        R.ref_bg_destroy.argtypes = [vp]
        R.ref_bg_v.argtypes = [vp, u64]
        R.ref_bg_v.restype = C.POINTER(C.c_float)
        R.ref_bg_write.argtypes = [vp, C.c_char_p, C.c_char_p]
        R.ref_motif_create.argtypes = [u64, u64, _f32p, vp, f, _f32p]
        R.ref_motif_create.restype = vp
        R.ref_motif_from_bamm_file.argtypes = [u64, u64, _f32p, vp, f, C.c_char_p]
        R.ref_motif_from_bamm_file.restype = vp
        R.ref_motif_destroy.argtypes = [vp]
        R.ref_motif_flat_size.argtypes = [vp]
        R.ref_motif_flat_size.restype = u64
        for name in ("ref_motif_get_v", "ref_motif_get_p", "ref_motif_get_s"):
            getattr(R, name).argtypes = [vp, _f32p]
        R.ref_motif_linear_s.argtypes = [vp, vp, u64]
        R.ref_motif_log_s.argtypes = [vp, vp, u64]
        R.ref_motif_write.argtypes = [vp, C.c_char_p, C.c_char_p]
        R.ref_em_create.argtypes = [vp, vp, vp, i, i, f]
        R.ref_em_create.restype = vp
        for name in ("ref_em_destroy", "ref_em_estep", "ref_em_mstep", "ref_em_optimize_q"):
            getattr(R, name).argtypes = [vp]
        R.ref_em_optimize.argtypes = [vp]
        R.ref_em_optimize.restype = i
        R.ref_em_mask.argtypes = [vp]
        R.ref_em_mask.restype = i
        R.ref_em_q.argtypes = [vp]
        R.ref_em_q.restype = f
        R.ref_em_llh.argtypes = [vp]
        R.ref_em_llh.restype = f
        R.ref_em_r.argtypes = [vp, u64]
        R.ref_em_r.restype = C.POINTER(C.c_float)
        R.ref_em_get_n.argtypes = [vp, _f32p]
        R.ref_em_write.argtypes = [vp, C.c_char_p, C.c_char_p, i]
        R.ref_logodds.argtypes = [vp, vp, vp, _f32p, _f32p, _u64p]
        R.ref_write_logodds.argtypes = [vp, vp, vp, C.c_char_p, C.c_char_p, i]
        R.ref_negset_create.argtypes = [vp, u64, u64, i]
        R.ref_negset_create.restype = vp
        R.ref_session_size.argtypes = [vp]
        R.ref_session_size.restype = u64
        R.ref_occurrence.argtypes = [vp, vp, vp, _f32p, u64, f, i, C.c_char_p, C.c_char_p, _f32p]
        R.ref_fdr_create.argtypes = [vp, vp, vp, vp, u64, i, i, i, i, i]
        R.ref_fdr_create.restype = vp
        R.ref_fdr_destroy.argtypes = [vp]
        R.ref_fdr_evaluate.argtypes = [vp, i, i, f, u64]
        R.ref_fdr_write.argtypes = [vp, C.c_char_p, C.c_char_p]
        R.ref_fdr_scores.argtypes = [vp, i, _f32p, u64]
        R.ref_fdr_scores.restype = u64
        R.ref_fdr_q.argtypes = [vp]
        R.ref_fdr_q.restype = f
        R.ref_fdr_stats_only.argtypes = [vp, _f32p, u64, _f32p, u64, _f32p, u64, _f32p, u64, i]

    def set_threads(self, n):
        self.R.ref_set_threads(int(n))

    # ---- sessions -------------------------------------------------------------------
    def session(self, codes, in_off, single_strand=False, seed=42, do_srand=True):
        return RefSession(self, codes, in_off, single_strand, seed, do_srand)


class RefSession:
    def __init__(self, ref: Reference, codes, in_off, single_strand, seed, do_srand, handle=None):
        self.ref, self.R = ref, ref.R
        if handle is None:
            in_off = _u64(in_off)
            self.N = len(in_off) - 1
            self.h = self.R.ref_session_create(np.ascontiguousarray(codes, np.uint8), in_off, self.N,
                                               int(single_strand), int(do_srand), seed)
        else:
            self.h = handle
            self.N = int(self.R.ref_session_size(handle))
        self.L = np.array([self.R.ref_seq_L(self.h, n) for n in range(self.N)], np.int64)
        self.off = np.concatenate([[0], np.cumsum(self.L)]).astype(np.uint64)
        self._keep = []

    def kmers(self):
        out = np.zeros(int(self.off[-1]), np.uint64)
        for n in range(self.N):
            p = self.R.ref_seq_kmer(self.h, n)
            out[int(self.off[n]):int(self.off[n + 1])] = np.ctypeslib.as_array(p, (int(self.L[n]),))
        return out

    def seq_codes(self):
        out = np.zeros(int(self.off[-1]), np.uint8)
        for n in range(self.N):
            p = self.R.ref_seq_codes(self.h, n)
            out[int(self.off[n]):int(self.off[n + 1])] = np.ctypeslib.as_array(p, (int(self.L[n]),))
        return out

    def bg(self, order, alpha):
        b = self.R.ref_bg_create(self.h, order, _f32(alpha))
        v = np.concatenate([np.ctypeslib.as_array(self.R.ref_bg_v(b, k), (4 ** (k + 1),)).copy()
                            for k in range(order + 1)])
        return b, v

    def motif(self, W, K, alpha, bg, q, v_flat):
        return self.R.ref_motif_create(W, K, _f32(alpha), bg, q, _f32(v_flat))

    def motif_v(self, m):
        out = np.zeros(int(self.R.ref_motif_flat_size(m)), np.float32)
        self.R.ref_motif_get_v(m, out)
        return out

    def motif_p(self, m):
        out = np.zeros(int(self.R.ref_motif_flat_size(m)), np.float32)
        self.R.ref_motif_get_p(m, out)
        return out

    def motif_s(self, m, K, W):
        out = np.zeros(4 ** (K + 1) * W, np.float32)
        self.R.ref_motif_get_s(m, out)
        return out

    def em(self, m, bg, optimizeQ=False, verbose=False, f=0.05):
        return self.R.ref_em_create(m, bg, self.h, int(optimizeQ), int(verbose), f)

    def em_r(self, e):
        out = np.zeros(int(self.off[-1]), np.float32)
        for n in range(self.N):
            p = self.R.ref_em_r(e, n)
            out[int(self.off[n]):int(self.off[n + 1])] = np.ctypeslib.as_array(p, (int(self.L[n]),))
        return out

    def em_n(self, e, K, W):
        out = np.zeros(v_size(K, W), np.float32)
        self.R.ref_em_get_n(e, out)
        return out

    def logodds(self, m, bg, W):
        mops = np.zeros(int((self.L - W + 1).sum()), np.float32)
        zoops = np.zeros(self.N, np.float32)
        z = np.zeros(self.N, np.uint64)
        self.R.ref_logodds(m, bg, self.h, mops, zoops, z)
        return mops, zoops, z

    def negset(self, s_order=2, m_fold=1, generic=False):
        """SeqGenerator(posSet, NULL, sOrder, 1, genericNeg).sample_bgseqset_by_fold(mFold)"""
        h = self.R.ref_negset_create(self.h, s_order, m_fold, int(generic))
        return RefSession(self.ref, None, None, True, 0, False, handle=h)

    def write_logodds(self, m, bg, ss, base="x"):
        """ScoreSeqSet::writeLogOdds -> bytes of <base>.logOddsZoops"""
        with tempfile.TemporaryDirectory() as d:
            self.R.ref_write_logodds(m, bg, self.h, d.encode(), base.encode(), int(ss))
            return open(os.path.join(d, base + ".logOddsZoops"), "rb").read()

    def fdr(self, neg, m, bg, cv_fold, mops, zoops, em=True, optimizeQ=False, frac=0.05, threads=1,
            save_pvalues=True, base="x", save_logodds=False):
        f = self.R.ref_fdr_create(self.h, neg.h, m, bg, cv_fold, int(mops), int(zoops), 1, int(save_pvalues), int(save_logodds))
        self.R.ref_fdr_evaluate(f, int(em), int(optimizeQ), frac, threads)
        files = {}
        with tempfile.TemporaryDirectory() as d:
            self.R.ref_fdr_write(f, d.encode(), base.encode())
            for name in os.listdir(d):
                files[name[len(base) + 1:]] = open(os.path.join(d, name), "rb").read()
        scores = []
        for which in range(4):
            n = int(self.R.ref_fdr_scores(f, which, np.zeros(1, np.float32), 0))
            buf = np.zeros(max(n, 1), np.float32)
            self.R.ref_fdr_scores(f, which, buf, n)
            scores.append(buf[:n].copy())
        q = float(self.R.ref_fdr_q(f))
        self.R.ref_fdr_destroy(f)
        return files, scores, q

    def occurrence(self, m, bg, W, neg_all, pval_cutoff, ss, base="x"):
        pv = np.zeros(int((self.L - W + 1).sum()), np.float32)
        with tempfile.TemporaryDirectory() as d:
            self.R.ref_occurrence(m, bg, self.h, _f32(neg_all), len(neg_all), pval_cutoff, int(ss), d.encode(),
                                  base.encode(), pv)
            return open(os.path.join(d, base + ".occurrence"), "rb").read(), pv

    def write_motif(self, m, base="m"):
        with tempfile.TemporaryDirectory() as d:
            self.R.ref_motif_write(m, d.encode(), base.encode())
            return (open(os.path.join(d, base + ".ihbcp"), "rb").read(),
                    open(os.path.join(d, base + ".ihbp"), "rb").read())

    def write_bg(self, b, base="bg"):
        with tempfile.TemporaryDirectory() as d:
            self.R.ref_bg_write(b, d.encode(), base.encode())
            return (open(os.path.join(d, base + ".hbcp"), "rb").read(),
                    open(os.path.join(d, base + ".hbp"), "rb").read())

    def close(self):
        if self.h:
            self.R.ref_session_destroy(self.h)
            self.h = None

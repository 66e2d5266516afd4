import ctypes as C, numpy as np
L = C.CDLL('/tmp/libpack_asan.so')
class P(C.Structure):
    _fields_ = [("n_seqs", C.c_uint64), ("n_words", C.c_uint64), ("n_exc", C.c_uint64), ("total_len", C.c_uint64), ("max_len", C.c_uint32), ("min_len", C.c_uint32),
                ("words", C.c_void_p), ("word_off", C.c_void_p), ("len", C.c_void_p), ("exc_off", C.c_void_p), ("exc_pos", C.c_void_p), ("exc_kmer", C.c_void_p), ("exc_clean", C.c_void_p)]
rng = np.random.default_rng(2)
for trial, (N, Lmax, nfrac, ss) in enumerate([(1, 1, 0.0, 0), (7, 30, 0.3, 0), (300, 200, 0.02, 0), (300, 200, 0.02, 1), (5, 9000, 0.001, 0), (50000, 120, 0.01, 0), (3, 2, 1.0, 0)]):
    lens = rng.integers(1, Lmax + 1, N)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    codes = rng.integers(1, 5, int(off[-1])).astype(np.uint8)
    codes[rng.random(len(codes)) < nfrac] = 0
    for th in (1, 5):
        L.bamm_set_host_threads(th)
        p = C.POINTER(P)()
        rc = L.bamm_pack_codes_seeded(codes.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), C.c_uint64(N), ss, 42, C.byref(p))
        assert rc == 0
        tot = p.contents.total_len
        for K in (0, 2, 5, 10):
            y = np.zeros(tot, np.uint32)
            assert L.bamm_unpack_y(p, K, y.ctypes.data_as(C.c_void_p)) == 0
            vbg = np.zeros(4 ** (min(K, 4) + 2), np.float32)
        alpha = np.array([1, 10, 10], np.float32); vb = np.zeros(4 + 16 + 64, np.float32)
        assert L.bamm_bg_model(p, 2, alpha.ctypes.data_as(C.c_void_p), vb.ctypes.data_as(C.c_void_p)) == 0
        print(trial, th, N, tot, p.contents.n_exc, float(vb[:4].sum()))
        L.bamm_packed_free(p)
out = np.zeros(64, np.int32); m = C.c_int(0)
for skip in (0, 1, 33, 34, 35, 1000, 10**6, 2**33 + 5):
    assert L.bamm_rand_stream_draws(7, C.c_uint64(skip), 1, 64, out.ctypes.data_as(C.c_void_p), C.byref(m)) == 0
    print("jump", skip, m.value, out[:3])
print("done")

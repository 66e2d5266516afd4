import ctypes as C, numpy as np, os, sys, tempfile
H = C.CDLL('/tmp/libbamm_host_asan.so')
H.bh_last_error.restype = C.c_char_p
def fasta(path):
    n, m = C.c_uint64(), C.c_uint64()
    rc = H.bh_read_fasta(path.encode(), C.byref(n), C.byref(m), None, None, None)
    if rc: return rc, None, None
    codes = np.zeros(max(m.value,1), np.uint8); off = np.zeros(n.value + 1, np.uint64); bf = np.zeros(4, np.float32)
    rc = H.bh_read_fasta(path.encode(), C.byref(n), C.byref(m), codes.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), bf.ctypes.data_as(C.c_void_p))
    return rc, codes[:m.value], off
d = tempfile.mkdtemp()
rng = np.random.default_rng(1)
cases = {
 "plain": ">a\nACGT\n>b\nGG\n",
 "noeol": ">a\nACGT\n>b\nGG",
 "crlf": ">a\tx\r\nACGT\r\n>b\r\nGG\r\n",
 "empties": ">a\n>b\n>c\nAC\n>d\n",
 "blank": "\n\n>a\n\nAC\n\nGT\n\n",
 "only_gt": ">\nAC\n>\nGT\n",
 "space": ">a\nAC GT\n",
 "nohdr": "ACGT\n>a\nAC\n",
 "one": ">a\nA",
}
for name, txt in cases.items():
    p = os.path.join(d, name + ".fa"); open(p, "w").write(txt)
    for th in (1, 3, 8):
        H.bh_set_threads(th)
        rc, codes, off = fasta(p)
        print(name, th, rc, None if off is None else off.tolist(), None if codes is None else codes.tolist()[:12], H.bh_last_error()[:40] if rc else b"")
# big multi-part file
recs = []
for n in range(40000):
    L = int(rng.integers(1, 300)); seq = "".join(rng.choice(list("ACGTNacgt"), L))
    recs.append(f">s{n}\n{seq}\n" + ("\n" if n % 17 == 0 else "") + (">e\n" if n % 501 == 0 else ""))
p = os.path.join(d, "big.fa"); open(p, "w").write("".join(recs))
H.bh_auto_threads()
rc, codes, off = fasta(p); print("big", rc, len(off), int(off[-1]), os.path.getsize(p))
# formatter + stats writer
H.bh_format_g_check.restype = C.c_uint64
x = rng.integers(0, 2**32, 2_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32)
bits = C.c_uint32(0)
print("format_g bad:", H.bh_format_g_check(x.ctypes.data_as(C.c_void_p), C.c_uint64(len(x)), 6, C.byref(bits)))
pos = (rng.standard_normal(20000) * 3 + 2).astype(np.float32); neg = (rng.standard_normal(200000) * 2).astype(np.float32); z = np.zeros(1, np.float32)
f = lambda a: a.ctypes.data_as(C.c_void_p)
H.bh_fdr_stats.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
print("fdr", H.bh_fdr_stats(f(pos), 20000, f(neg), 200000, f(pos), 20000, f(neg), 200000, 20000, 200000, 0.9, 1, 1, 1, d.encode(), b"x"))
print("done")

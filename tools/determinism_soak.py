#!/usr/bin/env python3
"""The same command, again and again: the drop-in CLI on config 5's shape (200k x 200 bp, --FDR -n 5 -m 10) and on config 3's
(--EM, 1M sequences), R runs each -- every output file of every run must have the same bytes (integer accumulation: nothing
depends on the order in which blocks, waves or atomics happen to finish).

    determinism_soak.py [R]        (GPU box; prints one line per configuration)"""
import hashlib, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bammmotif2_amd import synth, build

R = int(sys.argv[1]) if len(sys.argv) > 1 else 20
build.build_host()
W = 20
pwm = synth.make_pwm(W, 1234)
lut = np.frombuffer(b"NACGT", np.uint8)
for name, N, extra in (("config 5", 200000, ["--FDR", "-n", "5", "-m", "10"]), ("config 3", 1000000, [])):
    out = f"/tmp/soak_{N}"
    os.makedirs(out, exist_ok=True)
    codes, off = synth.make_sequences(N, 200, pwm, 1234)
    seqs = lut[codes].reshape(N, 200)
    with open(os.path.join(out, "pos.fasta"), "wb") as f:
        for n in range(N):
            f.write(b">s%d\n" % n); f.write(seqs[n].tobytes()); f.write(b"\n")
    with open(os.path.join(out, "seed.meme"), "w") as f:
        f.write("MEME version 4\n\nALPHABET= ACGT\n\nMOTIF m\nletter-probability matrix: alength= 4 w= %d nsites= 100\n" % W)
        blur = 0.7 * pwm + 0.075
        for j in range(W):
            f.write(" ".join("%.6f" % blur[y, j] for y in range(4)) + "\n")
    digests, t0 = set(), time.time()
    for r in range(R):
        res = os.path.join(out, "res")
        subprocess.run(["rm", "-rf", res])
        p = subprocess.run([build.CLI, res, os.path.join(out, "pos.fasta"), "--PWMFile", os.path.join(out, "seed.meme"), "--EM", "-k", "2",
                            "--maxEMIterations", "60"] + extra, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-2000:]
        h = hashlib.sha256()
        for fn in sorted(os.listdir(res)):
            h.update(fn.encode()); h.update(open(os.path.join(res, fn), "rb").read())
        digests.add(h.hexdigest())
    print(f"{name}: {R} runs in {time.time() - t0:.1f} s, {len(os.listdir(res))} files each, distinct digests over all files: {len(digests)}"
          f" ({'identical bytes every run' if len(digests) == 1 else 'RUNS DIFFER'})")
    assert len(digests) == 1

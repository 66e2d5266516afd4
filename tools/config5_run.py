#!/usr/bin/env python3
"""BASELINE config 5 through the drop-in CLI: 200k x 200 bp, -k 2, --FDR -n 5 -m 10 (5-fold CV,
10x sampled negatives).  Writes a FASTA + MEME seed, runs BaMMmotif, prints wall times (per stage
with --timing).  `config5_run.py N OUT em` runs the plain --EM line instead (config 3 at N=1M)."""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bammmotif2_amd import synth, build
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/c5"
os.makedirs(out, exist_ok=True)
W = 20
pwm = synth.make_pwm(W, 1234)
codes, off = synth.make_sequences(N, 200, pwm, 1234)
t = time.time()
lut = np.frombuffer(b"NACGT", np.uint8)
seqs = lut[codes].reshape(N, 200)
with open(os.path.join(out, "pos.fasta"), "wb") as f:
    for n in range(N):
        f.write(b">s%d\n" % n); f.write(seqs[n].tobytes()); f.write(b"\n")
with open(os.path.join(out, "seed.meme"), "w") as f:
    f.write("MEME version 4\n\nALPHABET= ACGT\n\nMOTIF m\nletter-probability matrix: alength= 4 w= %d nsites= 100\n" % W)
    blur = 0.7 * pwm + 0.075
    for j in range(W):
        f.write(" ".join("%.6f" % blur[y, j] for y in range(4)) + "\n")
print("inputs written in %.1f s" % (time.time() - t))
build.build_host()
t = time.time()
em_only = len(sys.argv) > 3 and sys.argv[3] == "em"
extra = [] if em_only else ["--FDR", "-n", "5", "-m", "10"]
r = subprocess.run([build.CLI, os.path.join(out, "res"), os.path.join(out, "pos.fasta"), "--PWMFile", os.path.join(out, "seed.meme"),
                    "--EM", "-k", "2", "--maxEMIterations", "60", "--timing"] + extra, capture_output=True, text=True)
dt = time.time() - t
print("BaMMmotif exit", r.returncode, "wall %.1f s" % dt)
print("\n".join(l for l in r.stdout.splitlines() if "Runtime" in l))
print(r.stderr[-2500:])
print(sorted(os.listdir(os.path.join(out, "res"))))
if not em_only:
    print(open(os.path.join(out, "res", "pos_motif_1.zoops.stats")).read()[:300])

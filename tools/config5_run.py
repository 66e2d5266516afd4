#!/usr/bin/env python3
"""BASELINE config 5 through the drop-in CLI: 200k x 200 bp, -k 2, --FDR -n 5 -m 10 (5-fold CV, 10x
sampled negatives).  Writes a FASTA + MEME seed, runs BaMMmotif with --timing and prints the wall time per
stage: the GPU part (full EM, fold EMs + scoring) apart from the host work the reference also does (FASTA,
rand()-driven negative sampler, sorting / statistics / writers).

    config5_run.py [N] [OUT] [em] [--gpus G | --deviceList a,b,..]

`em` runs the plain --EM line instead (config 3 at N = 1M)."""
import os, re, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bammmotif2_amd import synth, build
argv = sys.argv[1:]
extra_dev = []
for flag in ("--gpus", "--deviceList"):
    if flag in argv:
        i = argv.index(flag); extra_dev += argv[i:i + 2]; del argv[i:i + 2]
for flag in ("--hostSampler", "--hostPacking", "--debug"):                # the host flavours of the stages that run on the device by default
    if flag in argv:
        argv.remove(flag); extra_dev.append(flag)
N = int(argv[0]) if len(argv) > 0 else 200000
out = argv[1] if len(argv) > 1 else "/tmp/c5"
em_only = len(argv) > 2 and argv[2] == "em"
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
extra = [] if em_only else ["--FDR", "-n", "5", "-m", "10"]
r = subprocess.run([build.CLI, os.path.join(out, "res"), os.path.join(out, "pos.fasta"), "--PWMFile", os.path.join(out, "seed.meme"),
                    "--EM", "-k", "2", "--maxEMIterations", "60", "--timing"] + extra + extra_dev, capture_output=True, text=True)
dt = time.time() - t
print("BaMMmotif", " ".join(extra + extra_dev), "exit", r.returncode, "wall %.2f s" % dt)
stages = re.findall(r"\[timing\] (.*): ([\d.e+-]+) s", r.stderr)
gpu = sum(float(s) for n, s in stages if "EM" in n or "GPU" in n)
print("  stage                                                          seconds")
for n, s in stages:
    print("  %-62s %8.3f" % (n, float(s)))
for l in r.stderr.splitlines():
    if l.startswith("[timing-beside]"):
        print("  beside the stages above:", l[len("[timing-beside] "):])
ab = dict(re.findall(r"\[timing-abs\] main (entered|left) at ([\d.]+)", r.stderr))
td = re.search(r"\[timing-abs\] teardown done at ([\d.]+)", r.stderr)
if td:
    print("  --debug: orderly teardown %.3f s, then %.3f s to the reaped process" % (float(td.group(1)) - float(ab["left"]), t + dt - float(td.group(1))))
if len(ab) == 2:
    print("  outside main: %.3f s from the spawn to main, %.3f s from main's last line to the reaped process" % (float(ab["entered"]) - t, t + dt - float(ab["left"])))
print("  GPU stages (EM runs, fold EMs + scoring) %.3f s of %.3f s in all; the rest is host work" % (gpu, sum(float(s) for _, s in stages)))
print("\n".join(l for l in r.stdout.splitlines() if "Runtime" in l))
if r.returncode:
    print(r.stderr[-2500:])
print(sorted(os.listdir(os.path.join(out, "res"))))
if not em_only:
    print(open(os.path.join(out, "res", "pos_motif_1.zoops.stats")).read()[:300])

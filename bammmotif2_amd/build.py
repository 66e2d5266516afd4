"""Build the gfx950 shared library in-tree (``bammmotif2_amd/libbamm_em.so``).

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import contextlib
import fcntl
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbamm_em.so")
SOURCES = ["kernels.hip", "grouped.hip", "grouped_long.hip", "grouped_xl.hip", "mask.hip", "seed.hip", "abi.cpp", "pack.cpp"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "device_utils.h"), os.path.join(CSRC, "grouped_kernel.h"),
           os.path.join(HERE, "..", "include", "bamm_em.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-result", "-Wno-pass-failed"]
OBJDIR = os.path.join(HERE, "build")


def _obj(src: str) -> str:
    return os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def is_stale() -> bool:
    return _stale(LIB, [os.path.join(CSRC, s) for s in SOURCES] + HEADERS)


@contextlib.contextmanager
def _build_lock():
    """One builder at a time: the ranks of a multi-GPU run all pass through here at start-up."""
    os.makedirs(OBJDIR, exist_ok=True)
    with open(os.path.join(OBJDIR, ".lock"), "w") as fh:
        fcntl.flock(fh, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(fh, fcntl.LOCK_UN)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """One object per source (rebuilt only when it or a header changed, in parallel), then link."""
    if not force and not is_stale():
        return LIB
    with _build_lock():
        if not force and not is_stale():                     # another rank built it while this one waited
            return LIB
        return _build_library_locked(force, verbose)


def _build_library_locked(force: bool, verbose: bool) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the gfx950 extension")
    os.makedirs(OBJDIR, exist_ok=True)
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), _obj(s)
        if force or _stale(obj, [src] + HEADERS):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, proc in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s) for s in SOURCES] + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


HOST = os.path.join(HERE, "host")
HOST_LIB = os.path.join(HERE, "libbamm_host.so")
CLI = os.path.join(HERE, "BaMMmotif")
HOST_SOURCES = ["io.cpp", "fdr.cpp", "hooks.cpp"]


def build_host(force: bool = False, verbose: bool = False):
    """C++17 host code: libbamm_host.so (test hooks) and the `BaMMmotif` drop-in CLI."""
    build_library(force=False, verbose=verbose)
    deps = [os.path.join(HOST, f) for f in HOST_SOURCES + ["main.cpp", "bamm_host.h"]] + [LIB]
    outs = [HOST_LIB, CLI]
    if not force and all(os.path.exists(o) for o in outs) and \
            min(os.path.getmtime(o) for o in outs) >= max(os.path.getmtime(d) for d in deps):
        return outs
    cxx = shutil.which("g++") or "g++"
    common = [cxx, "-std=c++17", "-O2", "-fopenmp", "-Wall", "-fPIC", "-L" + HERE, "-Wl,-rpath,$ORIGIN"]
    cmd = common + ["-shared"] + [os.path.join(HOST, f) for f in HOST_SOURCES] + ["-lbamm_em", "-o", HOST_LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    cmd = common + [os.path.join(HOST, "main.cpp"), os.path.join(HOST, "io.cpp"), os.path.join(HOST, "fdr.cpp"), "-lbamm_em", "-o", CLI]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return outs


if __name__ == "__main__":
    build_host(force=True, verbose=True)
    print(build_library(force=True, verbose=True))

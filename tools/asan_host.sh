#!/bin/bash
# The host side (FASTA reader, FDR statistics and writers, the %g formatter) under AddressSanitizer + UBSan on the CPU: edge-case
# FASTA files at 1 / 3 / 8 threads, a 6 MB file cut over all cores, 2 M floats through format_g, the statistics + writers.
#   bash tools/asan_host.sh        (no GPU needed; prints the cases and "done", and nothing from the sanitizers)
set -e
cd "$(dirname "$0")/../bammmotif2_amd"
g++ -std=c++17 -O1 -g -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -Wall -fPIC -L. -Wl,-rpath,$PWD -shared \
    host/io.cpp host/fdr.cpp host/hooks.cpp -lbamm_em -o /tmp/libbamm_host_asan.so
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python3 ../tools/asan_host.py 2>&1 | grep -v "^Warning: Ignore FASTA"
# ... and the host restatements of the input side (csrc/pack.cpp: Sequence::Sequence, BackgroundModel, the rand() stream's jump-ahead)
g++ -std=c++17 -O1 -g -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -shared \
    csrc/pack.cpp -o /tmp/libpack_asan.so
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python3 ../tools/asan_pack.py

// Sequence::Sequence and the counting pass of BackgroundModel where the data will live (csrc/prep.hip): from the alphabet
// codes of the FASTA records to the 2-bit stream, the exception list and the k-mer counts, on the device.
// Reference (file:line relative to /root/reference/src): init/Sequence.cpp:4-43 (reverse complement behind an N, kmer_[i]
// from up to 11 bases, rand() % 4 per (position, digit) term for an unknown base), init/Alphabet.cpp:46-55 (the complement
// table maps code 0 to the BYTE 'N' = 78), init/BackgroundModel.cpp:26-42 (every position counts once per order).
#pragma once
#include "common.h"

namespace bamm {

// pack.cpp (host): the first D draws rand() % 4 of the stream srand(seed) starts, on all host threads (jump-ahead)
void rand_draws_mod4(uint32_t seed, uint64_t D, uint8_t* out);
// pack.cpp (host): the host threads the process was granted (bamm_set_host_threads; a modest default when unset)
uint32_t host_threads_hint();
// pack.cpp (host): BackgroundModel::calculateV from the counts of the highest order
void bg_from_top_counts(const uint64_t* top_counts, uint32_t K, const float* alpha, float* vbg_out);

struct PrepArgs {
    const uint8_t* codes;        // device: 0 = unknown, 1..4 = A, C, G, T (anything else enters the arithmetic as code - 1)
    const uint64_t* off;         // device [n + 1]: first code of every record
    uint64_t n;
    int single_strand;
    // per sequence, filled by k_prep_count and scanned in place (exclusive; [n] = total):
    uint32_t* len;               // [n]     L = L0, or 2 L0 + 1
    uint64_t* word_off;          // [n + 1] 32-bit words of the stream
    uint64_t* pos_off;           // [n + 1]
    uint64_t* zero_off;          // [n + 1] positions that hold code 0 (the strand separator included)
    uint64_t* draw_off;          // [n + 1] rand() draws: min(11, L - z) per such position z
    uint64_t* exc_off;           // [n + 1] exceptions (filled by the counting run of k_prep_pack)
    uint32_t* zero_pos;          // [zero_off[n]] ascending within a sequence
    const uint8_t* draws;        // [draw_off[n]] rand() % 4, in the reference's order
    uint32_t* words;             // [word_off[n]] zeroed by the caller
    uint32_t* exc_pos;           // [exc_off[n]]
    uint32_t* exc_kmer;
    uint32_t* exc_clean;
};

int launch_prep_count(const PrepArgs& a, hipStream_t st);                  // len, and the per-sequence counts behind word/pos/zero/draw_off[i + 1]
int launch_prep_zeros(const PrepArgs& a, hipStream_t st);                  // zero_pos
int launch_prep_pack(const PrepArgs& a, bool write, hipStream_t st);       // write = false: exception counts into exc_off[i + 1] only
int launch_scan_u64(uint64_t* data, uint64_t n_plus_1, hipStream_t st);    // in place: data[0] = 0 on entry, data[i + 1] = count of i  ->  exclusive prefix sums, data[n] = total
// k-mer counts of order K over a resident set (the stream + the set's exception list of that order: (position, kmer_ mod
// 4^(K+1)) where the stream implies something else), BackgroundModel.cpp:26-42: counts[4^(K+1)] += 1 per position
int launch_bg_counts(const uint32_t* words, const uint64_t* word_off, const uint32_t* len, const uint64_t* exc_off,
                     const uint2* exc, uint64_t n, uint32_t K, unsigned long long* counts, uint32_t num_cus, hipStream_t st);

}  // namespace bamm

// SeqGenerator's negative sampler on the device (csrc/negs.hip); reference: /root/reference/src/seq_generator/SeqGenerator.cpp:63-348.
#pragma once
#include "common.h"

namespace bamm {

constexpr uint32_t kNegMaxOrder = 2;                          // the sequence-specific rescaling is written for s = 2 (:112-186)
constexpr uint32_t kNegTable = 84;                            // 4 + 16 + 64 cells of the orders 0..2

struct NegArgs {
    // the resident positives (bamm_seqs) and their exception list of order s
    const uint32_t* words; const uint64_t* word_off; const uint32_t* len; const uint64_t* exc_off; const uint2* exc;
    uint64_t n;
    uint32_t s;                  // order of the sampler's conditionals (-s, 2)
    int generic;                 // --genericNeg: the set's own bars for every positive
    uint64_t m_fold, keep_stride;
    unsigned long long* total_counts;   // [kNegTable] k_neg_counts adds into it
    const float* v;              // [kNegTable] the set's conditionals (host: kmer_frequency's tail)
    const float* bar;            // [kNegTable] their cumulative bars
    float A[kNegMaxOrder + 1];   // pseudo-counts (20, SeqGenerator.cpp:29-32)
    const uint64_t* draw0;       // [n] first draw of positive i: sum over the positives before it of L * m_fold
    const uint32_t* seed_state;  // [34] the generator as srand(42) leaves it
    const uint32_t* pow2;        // [48][31] t^(2^b) mod (t^31 - t^28 - 1)
    const uint64_t* out_word_off;// [n] first output word of positive i's kept negatives
    uint32_t* out_words;
    uint32_t* bad;               // counts the draws the reference leaves undefined (rand() == RAND_MAX at a first base)
};

int launch_neg_counts(const NegArgs& a, hipStream_t st);
int launch_neg_sample(const NegArgs& a, hipStream_t st);

}  // namespace bamm

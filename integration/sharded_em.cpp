// sharded_em.cpp -- the multi-GPU form of the EM path in a caller's own program: one process, one host
// thread and one bamm_ctx per GPU, the sequences sharded by bamm_shard_range, ONE RCCL all-reduce of the
// integer count accumulator per EM iteration issued by libbamm_em itself (bamm_em_set_comm), after which
// every GPU holds the identical model.  This is what `BaMMmotif --gpus N --EM` does
// (bammmotif2_amd/host/main.cpp); tests/test_host_cpu.py compiles this file against include/bamm_em.h.
//
//   g++ -std=c++17 -pthread -I include integration/sharded_em.cpp -L bammmotif2_amd -lbamm_em
//
// Reference reduction points this replaces: the OpenMP `reduction(+:llikelihood)` of EM::EStep
// (src/refinement/EM.cpp:148), the CAS float adds into n_[K] of EM::MStep (EM.cpp:203-215,240) and the
// serial sum over r_ of EM::optimize_q (EM.cpp:509-513).
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "bamm_em.h"

// packed: the whole set (bamm_pack_codes / bamm_pack_kmer_ptrs); vbg, A, v: flat models as in bamm_em.h.
// On return v holds the refined model.  Returns the number of EM iterations, or -1 (message on stderr).
int sharded_em(const bamm_packed* packed, const std::vector<int>& devices, const bamm_em_params& params,
               const std::vector<float>& vbg, const std::vector<float>& A, std::vector<float>& v) {
    const uint32_t R = (uint32_t)devices.size();
    std::vector<bamm_ctx*> ctx(R, nullptr);
    std::vector<bamm_seqs*> shard(R, nullptr);
    std::vector<bamm_comm*> comm(R, nullptr);
    std::vector<bamm_em*> em(R, nullptr);
    std::vector<std::string> err(R);
    std::vector<uint32_t> iterations(R, 0);
    auto fail = [&](const char* what) { fprintf(stderr, "%s: %s\n", what, bamm_last_error()); return -1; };

    for (uint32_t r = 0; r < R; r++) {
        if (bamm_ctx_create(devices[r], nullptr, &ctx[r])) return fail("context");
        uint64_t b = 0, e = 0;                               // contiguous range balanced by sum(L - W + 1)
        if (bamm_shard_range(packed->len, packed->n_seqs, params.W, r, R, &b, &e)) return fail("shard range");
        if (bamm_seqs_upload(ctx[r], packed, b, e, &shard[r])) return fail("upload");
    }
    if (bamm_comm_init_all(ctx.data(), R, comm.data())) return fail("RCCL communicator");   // ncclCommInitAll
    bamm_em_params p = params;
    p.n_seqs_bound = packed->n_seqs;                          // every rank: the same unit for the int64 accumulator
    for (uint32_t r = 0; r < R; r++) {
        if (bamm_em_create(ctx[r], shard[r], &p, vbg.data(), A.data(), v.data(), nullptr, &em[r])) return fail("EM");
        if (bamm_em_set_comm(em[r], comm[r])) return fail("EM communicator");
    }
    // One std::thread per rank (an OpenMP team may come back smaller than asked for -- OMP_THREAD_LIMIT, dynamic
    // teams -- and the missing ranks would leave the others waiting in the collective for ever).
    // EM::optimize (EM.cpp:62-137): every pass = local E+M over the shard, ncclAllReduce(int64, sum) of
    // [n_K | llh | sum_r | N] on this GPU's stream, the update.  The stopping rule reads the same numbers
    // on every rank, so all of them leave the loop in the same pass.  A rank that fails alone aborts every
    // communicator: its peers' collectives return BAMM_ERR_COMM instead of blocking.
    std::vector<std::thread> team;
    for (uint32_t r = 0; r < R; r++)
        team.emplace_back([&, r] {
            if (bamm_em_optimize(em[r], &iterations[r])) {
                err[r] = bamm_last_error();
                for (uint32_t o = 0; o < R; o++) bamm_comm_abort(comm[o]);
            }
        });
    for (auto& t : team) t.join();
    int rc = (int)iterations[0];
    for (uint32_t r = 0; r < R; r++)
        if (!err[r].empty()) { fprintf(stderr, "GPU %d: %s\n", devices[r], err[r].c_str()); rc = -1; }
    if (rc >= 0 && bamm_em_get_v(em[0], v.data())) rc = fail("read-back");
    for (uint32_t r = 0; r < R; r++) {
        bamm_em_destroy(em[r]);
        bamm_comm_destroy(comm[r]);
        bamm_seqs_destroy(shard[r]);
        bamm_ctx_destroy(ctx[r]);
    }
    return rc;
}

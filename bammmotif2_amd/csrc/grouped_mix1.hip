// Mixed-row kernel (mixed_kernel.h) with ONE wide group (W mod 3 == 1, e.g. W = 19 = 3+3+3+3+3+4).
#include "mixed_kernel.h"

namespace bamm {

// (the planner keeps the narrow groups within one quad -- mix_geometry: more of them lost to the uniform rows --,
// so only NQ = 1 is instantiated)
#define BAMM_MIX_CASE(idx, M, T, A)                                                                       \
    case idx * 4 + 1: rc = launch_mix_variant<M, A, 1, T>(accum, write_r, a, blocks, st); break;

int launch_em_mix1(int mclass, bool accum, bool write_r, const GrpKernelArgs& a, uint32_t blocks, uint32_t threads, hipStream_t st) {
    if (threads != grp_max_threads(kMClasses[mclass]) || a.g.mixA != 1u || a.g.Tq != 1u) {
        set_error("mixed-row kernel: bad launch (%u threads, A=%u, %u quads)", threads, a.g.mixA, a.g.Tq);
        return BAMM_ERR_ARG;
    }
    int rc = BAMM_ERR_UNSUPPORTED;
    switch (mclass * 4 + (int)a.g.Tq) {
        BAMM_MIX_CASE(3, 4, 1024, 1) BAMM_MIX_CASE(4, 5, 1024, 1) BAMM_MIX_CASE(5, 6, 1024, 1) BAMM_MIX_CASE(6, 7, 1024, 1)
        BAMM_MIX_CASE(7, 8, 1024, 1) BAMM_MIX_CASE(8, 10, 768, 1)
        default: set_error("no mixed-row kernel for M class %d", mclass);
    }
    if (rc) return rc;
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

}  // namespace bamm

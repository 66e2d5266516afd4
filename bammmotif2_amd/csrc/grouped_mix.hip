// Mixed-row kernel (mixed_kernel.h) with TWO wide groups (W mod 3 == 2, e.g. W = 20 = 3+3+3+3+4+4): the
// instantiations for 4..10 positions per lane.  A translation unit of its own: it compiles next to the others.
#include "mixed_kernel.h"

namespace bamm {

// (the planner keeps the narrow groups within one quad -- mix_geometry: more of them lost to the uniform rows --,
// so only NQ = 1 is instantiated)
#define BAMM_MIX_CASE(idx, M, T, A)                                                                       \
    case idx * 4 + 1: rc = launch_mix_variant<M, A, 1, T>(accum, write_r, a, blocks, st); break;

// arguments checked by launch_em_grp
int launch_em_mix(int mclass, bool accum, bool write_r, const GrpKernelArgs& a, uint32_t blocks, uint32_t threads, hipStream_t st) {
    if (threads != grp_max_threads(kMClasses[mclass]) || a.g.mixA != 2u || a.g.Tq != 1u) {
        set_error("mixed-row kernel: bad launch (%u threads, A=%u, %u quads)", threads, a.g.mixA, a.g.Tq);
        return BAMM_ERR_ARG;
    }
    int rc = BAMM_ERR_UNSUPPORTED;
    switch (mclass * 4 + (int)a.g.Tq) {
        BAMM_MIX_CASE(3, 4, 1024, 2) BAMM_MIX_CASE(4, 5, 1024, 2) BAMM_MIX_CASE(5, 6, 1024, 2) BAMM_MIX_CASE(6, 7, 1024, 2)
        BAMM_MIX_CASE(7, 8, 1024, 2) BAMM_MIX_CASE(8, 10, 768, 2)
        default: set_error("no mixed-row kernel for M class %d", mclass);
    }
    if (rc) return rc;
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

}  // namespace bamm

#ifdef BAMM_PHASE_CLOCK
// debug builds only: the phase clocks of the last k_em_mix launch of this translation unit ([256 blocks][16], 100 MHz ticks)
extern "C" int bamm_debug_phase_clock(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(bamm::g_phase_clock), 256 * 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

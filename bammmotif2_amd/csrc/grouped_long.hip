// The grouped-column kernel (grouped_kernel.h, grouped.hip) for 20..32 positions per lane: sequences of
// 1025..2048 positions (513..1023 bp double-stranded), blocks of 512 threads.  A translation unit of its
// own so that the instantiations compile next to those of the shorter classes.

#include "grouped_kernel.h"

namespace bamm {

#define BAMM_FOR_EACH_GCLASS_LONG(X) X(12, 20, 512) X(13, 24, 512) X(14, 28, 512) X(15, 32, 512)

// arguments checked by launch_em_grp
int launch_em_grp_long(int mclass, bool accum, bool write_r, const GrpKernelArgs& a, uint32_t blocks, uint32_t threads,
                       hipStream_t st) {
    if (kMClasses[mclass] > 32) return launch_em_grp_xl(mclass, accum, write_r, a, blocks, threads, st);
    const uint32_t KG = a.e.K + a.g.G;
    switch (mclass * 64 + (int)a.g.G * 8 + (int)KG) {
        BAMM_FOR_EACH_GCLASS_LONG(BAMM_GRP_CASES)
        default:
            set_error("no grouped kernel for M class %d, G=%u", mclass, a.g.G);
            return BAMM_ERR_UNSUPPORTED;
    }
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

}  // namespace bamm

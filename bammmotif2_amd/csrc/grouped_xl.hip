// The grouped-column kernel (grouped_kernel.h, grouped.hip) for 40 .. 64 positions per lane: sequences of
// 2049..4096 positions (1024..2047 bp double-stranded), blocks of 256 threads -- one wave per SIMD, which
// owns all 512 registers.  A third translation unit: these instantiations are the slowest to compile.
// At 56 and 64 positions per lane the E-chain reads its table one slot at a time (grp_chain, M > 48): the
// 4*M destination registers of M hand-issued ds_read_b128 no longer fit the register file there (round 1: the
// compiler moved destinations it believed ready, wrong results); M ds_read_b32 in flight do.

#include "grouped_kernel.h"

namespace bamm {

#define BAMM_FOR_EACH_GCLASS_XL(X) X(16, 40, 256) X(17, 48, 256) X(18, 56, 256) X(19, 64, 256)

// arguments checked by launch_em_grp
int launch_em_grp_xl(int mclass, bool accum, bool write_r, const GrpKernelArgs& a, uint32_t blocks, uint32_t threads,
                     hipStream_t st) {
    const uint32_t KG = a.e.K + a.g.G;
    switch (mclass * 64 + (int)a.g.G * 8 + (int)KG) {
        BAMM_FOR_EACH_GCLASS_XL(BAMM_GRP_CASES)
        default:
            set_error("no grouped kernel for M class %d, G=%u", mclass, a.g.G);
            return BAMM_ERR_UNSUPPORTED;
    }
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

}  // namespace bamm

// The grouped-column kernel (grouped_kernel.h, grouped.hip) for 40 and 48 positions per lane: sequences of
// 2049..3072 positions (1024..1535 bp double-stranded), blocks of 256 threads -- one wave per SIMD, which
// owns all 512 registers.  A third translation unit: these instantiations are the slowest to compile.
// 56 and 64 positions per lane are NOT instantiated: there the E-chain's M hand-issued ds_read_b128 (4*M
// destination registers in flight until lds_wait) no longer fit the register file, the compiler spills
// destinations it believes ready, and the results are wrong (caught by test_grouped_gpu.py at those
// lengths during round 1).  Those sequences stay on k_em_seq.

#include "grouped_kernel.h"

namespace bamm {

#define BAMM_FOR_EACH_GCLASS_XL(X) X(16, 40, 256) X(17, 48, 256)

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

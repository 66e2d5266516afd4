// Debug builds only (-DBAMM_PHASE_CLOCK, tools/phase_clock.py): thread 0 of every block of the bench class's kernel
// leaves the 100 MHz wall clock at the phase boundaries of its last launch; bamm_debug_phase_clock (grouped_mix.hip)
// reads them back.  Release builds compile none of it.
#pragma once
#include <hip/hip_runtime.h>

#ifdef BAMM_PHASE_CLOCK
namespace bamm {
namespace {
__device__ unsigned long long g_phase_clock[256 * 16];
}
}
#define BAMM_PHASE(i) do { if (threadIdx.x == 0 && blockIdx.x < 256u) bamm::g_phase_clock[blockIdx.x * 16u + (i)] = wall_clock64(); } while (0)
#else
#define BAMM_PHASE(i) do { } while (0)
#endif

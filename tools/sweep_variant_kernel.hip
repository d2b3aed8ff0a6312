// The shipped d = 32 sweep kernel for tools/sweep_variants.hip: the library's own kernel body (abcdez.jl_amd/csrc/abz_kernels.h),
// optionally with a forced occupancy or other -D knobs of this translation unit.
//   hipcc -c -DVNAME=cur [-DVWAVES=4] tools/sweep_variant_kernel.hip
#include "abz_kernels.h"

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#ifdef VWAVES
__global__ __launch_bounds__(ABZ_BLOCK) __attribute__((amdgpu_waves_per_eu(VWAVES, VWAVES)))
#else
__global__ __launch_bounds__(ABZ_BLOCK)
#endif
void CAT(sweep_kernel_, VNAME)(const SmcPackedArgs a) {
  smc_swarm_packed_body<ABZ_SIM_MVN, 4, 8, true>(a);
}

extern "C" int CAT(sweep_occ_, VNAME)() {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, CAT(sweep_kernel_, VNAME), ABZ_BLOCK, 0) != hipSuccess) return -1;
  return nb;
}
extern "C" void CAT(sweep_launch_, VNAME)(const SmcPackedArgs* a, unsigned grid, hipStream_t st) {
  hipLaunchKernelGGL(CAT(sweep_kernel_, VNAME), dim3(grid), dim3(ABZ_BLOCK), 0, st, *a);
}

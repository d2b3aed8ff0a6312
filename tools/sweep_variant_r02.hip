// Round 2's sweep kernel (Box-Muller sampler, Philox4x32-10, one tile per workgroup, tables staged per workgroup) as a variant of
// tools/sweep_variants.hip: compiled against the round-2 headers of a git worktree (`git worktree add _r02 <round-2 commit>`),
// with its own argument struct and tables -- the tool hands over plain pointers and scalars.
#include <string.h>
#include "abz_dispatch.h"
#include "abz_kernels.h"

__global__ __launch_bounds__(ABZ_BLOCK) void sweep_kernel_r02(const SmcPackedArgs a) {
  smc_swarm_packed_body<ABZ_SIM_MVN, 4, 8, true>(a);
}

extern "C" int sweep_occ_r02() {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sweep_kernel_r02, ABZ_BLOCK, 0) != hipSuccess) return -1;
  return nb;
}
// generic arguments: (bits, bits_out, slot0, slot1, logpi, delta, cslots, prior, data, eps, gamma0, gsig, n_alive, sweep)
extern "C" void sweep_launch_r02_raw(const uint32_t* bits, uint32_t* bits_out, double* slot0, double* slot1, double* logpi, double* delta,
                                     unsigned long long* cslots, const void* prior, const double* data, double eps, double gamma0,
                                     double gsig, uint32_t n_alive, uint32_t sweep, hipStream_t st) {
  static abz_tables* d_tab = nullptr;
  if (!d_tab) {
    (void)hipMalloc((void**)&d_tab, sizeof(abz_tables));
    (void)hipMemcpy(d_tab, &abz_tables_host, sizeof(abz_tables), hipMemcpyHostToDevice);
  }
  SmcPackedArgs a;
  memset(&a, 0, sizeof(a));
  a.hm.seed = 1; a.hm.prior = (const abz_prior_dim*)prior; a.hm.data = data; a.hm.tables = d_tab;
  a.hm.sim_p[0] = 1.0; a.hm.d = 32; a.hm.abck = ABZ_K_INDICATOR_STRICT; a.hm.n_data = 32; a.hm.n_blob = 0;
  a.bits = bits; a.bits_out = bits_out; a.slot0 = slot0; a.slot1 = slot1; a.logpi = logpi; a.delta = delta; a.cslots = cslots;
  a.eps = eps; a.gamma0 = gamma0; a.gsig = gsig; a.n_alive = n_alive; a.r_lo = 0; a.n_work = n_alive; a.sweep = sweep; a.c_cls = ABZ_C_NACC;
  const unsigned grid = (unsigned)(((uint64_t)n_alive * 4 + ABZ_BLOCK - 1) / ABZ_BLOCK);
  hipLaunchKernelGGL(sweep_kernel_r02, dim3(grid), dim3(ABZ_BLOCK), 0, st, a);
}

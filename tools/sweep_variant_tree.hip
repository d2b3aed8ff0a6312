// The d = 32 sweep kernel of ANOTHER COMMIT of this repository as a variant of tools/sweep_variants.hip: compiled against the
// headers of a git worktree (-I<tree>/abcdez.jl_amd/csrc -I<tree>/include), with that tree's argument struct and tables -- the
// tool hands over plain pointers and scalars.
//   -DTREE=r02   round 2 (commit 79040f7): Box-Muller + Philox4x32-10, one tile per workgroup, 5 waves per SIMD
//   -DTREE=icdf -DTREE_ICDF   round 3's experiment (commit 4d39988): inverse-CDF normal + Philox4x32-7, workgroups that loop
//                             over wave-tiles, tables staged once per workgroup, next-round prefetch, 4 waves per SIMD
#include <string.h>
#include "abz_dispatch.h"
#include "abz_kernels.h"
#ifdef TREE_ICDF
#include "abcdez_tables_data.h"
#endif

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#ifdef TREE_ICDF
__global__ __launch_bounds__(ABZ_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
__global__ __launch_bounds__(ABZ_BLOCK)
#endif
void CAT(sweep_kernel_, TREE)(const SmcPackedArgs a) {
  smc_swarm_packed_body<ABZ_SIM_MVN, 4, 8, true>(a);
}

extern "C" int CAT(sweep_occ_, TREE)() {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, CAT(sweep_kernel_, TREE), ABZ_BLOCK, 0) != hipSuccess) return -1;
  return nb;
}
extern "C" void CAT(CAT(sweep_launch_, TREE), _raw)(const uint32_t* bits, uint32_t* bits_out, double* slot0, double* slot1, double* logpi,
                                                   double* delta, unsigned long long* cslots, const void* prior, const double* data, double eps,
                                                   double gamma0, double gsig, uint32_t n_alive, uint32_t sweep, int ncu, hipStream_t st) {
  static abz_tables* d_tab = nullptr;
  SmcPackedArgs a;
  memset(&a, 0, sizeof(a));
#ifdef TREE_ICDF
  static abz_f64x2* d_all = nullptr;
  if (!d_tab) {
    (void)hipMalloc((void**)&d_all, sizeof(abz_icdf_all_data));
    (void)hipMemcpy(d_all, abz_icdf_all_data, sizeof(abz_icdf_all_data), hipMemcpyHostToDevice);
    abz_tables* ht = new abz_tables(abz_tables_host);
    for (int q = 0; q < ABZ_ICDF_PIECES; ++q) for (int r = 0; r < ABZ_ICDF_HOT_ROWS; ++r) ht->icdf_hot[q][r] = abz_icdf_all_data[q][r];
    ht->icdf_all = d_all;
    (void)hipMalloc((void**)&d_tab, sizeof(abz_tables));
    (void)hipMemcpy(d_tab, ht, sizeof(abz_tables), hipMemcpyHostToDevice);
    delete ht;
  }
  a.hm.icdf_all = d_all;
#else
  if (!d_tab) {
    (void)hipMalloc((void**)&d_tab, sizeof(abz_tables));
    (void)hipMemcpy(d_tab, &abz_tables_host, sizeof(abz_tables), hipMemcpyHostToDevice);
  }
#endif
  a.hm.seed = 1; a.hm.prior = (const abz_prior_dim*)prior; a.hm.data = data; a.hm.tables = d_tab;
  a.hm.sim_p[0] = 1.0; a.hm.d = 32; a.hm.abck = ABZ_K_INDICATOR_STRICT; a.hm.n_data = 32; a.hm.n_blob = 0;
  a.bits = bits; a.bits_out = bits_out; a.slot0 = slot0; a.slot1 = slot1; a.logpi = logpi; a.delta = delta; a.cslots = cslots;
  a.eps = eps; a.gamma0 = gamma0; a.gsig = gsig; a.n_alive = n_alive; a.r_lo = 0; a.n_work = n_alive; a.sweep = sweep; a.c_cls = ABZ_C_NACC;
#ifdef TREE_ICDF
  const unsigned tiles = (n_alive + 127u) / 128u;        // positions per workgroup and loop trip
  const unsigned res = (unsigned)ncu * (unsigned)CAT(sweep_occ_, TREE)();
  const unsigned per = (tiles + res - 1) / res;
  const unsigned grid = tiles <= res ? tiles : (tiles + per - 1) / per;
#else
  (void)ncu;
  const unsigned grid = (unsigned)(((uint64_t)n_alive * 4 + ABZ_BLOCK - 1) / ABZ_BLOCK);
#endif
  hipLaunchKernelGGL(CAT(sweep_kernel_, TREE), dim3(grid), dim3(ABZ_BLOCK), 0, st, a);
}

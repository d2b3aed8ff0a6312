/*
 * abz_jit.hip -- user-supplied simulators.
 *
 * The reference calls an arbitrary closure dist!(theta, ve) (src/abcdez_smc.jl:137).  The closest
 * a GPU library can offer is a device function supplied as source text:
 *
 *     __device__ double abz_user_dist(const double* theta, int d, const double* data, int n_data,
 *                                     const double* sim_p, abz_user_rng& rng);
 *
 * compiled here with hiprtc together with the very same kernel bodies the built-in simulators use
 * (abz_kernels.h), -ffp-contract=off like the rest of the library.  theta arrives push_p-cast.
 */
#include <hip/hiprtc.h>

#include <string.h>

#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "abz_ctx.h"
#include "abz_jit_sources.h"
#include "abz_kernels.h"

/* compiled code objects by (architecture, options, translation unit) */
static std::mutex g_code_mutex;
static std::unordered_map<std::string, std::vector<char>> g_code_cache;

struct AbzUserModule {
  hipModule_t mod = nullptr;
  hipFunction_t f_init = nullptr, f_smcp = nullptr, f_mc = nullptr, f_blob = nullptr;
  hipFunction_t f_p1 = nullptr, f_p2 = nullptr;     /* the sweep as two launches (rows of 4, 8 or 16 doubles; abz_kernels.h, smc_split_phase1_body) */
  hipFunction_t f_replay = nullptr;                 /* replay of a sharded sweep on the replicas (it evaluates the accepted rows' log-priors) */
  unsigned block_smc = ABZ_BLOCK, block_p2 = ABZ_BLOCK;
  bool rounds = false;                              /* the second launch runs the staged form round by round (abz_user_rounds.h) */
};

/* the user kernels loop over tiles like the built-in ones (ABZ_TILE_LOOP): grid = what is resident at once */
static unsigned jit_grid(abcdez_ctx* ctx, hipFunction_t f, uint64_t ntiles) {
  const void* key = reinterpret_cast<const void*>(f);
  auto it = ctx->occ.find(key);
  int per_cu;
  if (it != ctx->occ.end()) per_cu = it->second;
  else {
    int nb = 0;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, ABZ_BLOCK, 0) != hipSuccess || nb < 1) nb = 1;
    per_cu = nb;
    ctx->occ.emplace(key, per_cu);
  }
  return abz_tiles_to_grid(ntiles, (uint64_t)ctx->n_cu * (uint64_t)per_cu);
}

#define ABZ_RTC_CHECK(expr)                                                                \
  do {                                                                                     \
    hiprtcResult _r = (expr);                                                              \
    if (_r != HIPRTC_SUCCESS) {                                                            \
      abz_set_error(std::string(#expr) + ": " + hiprtcGetErrorString(_r));                 \
      return -4;                                                                           \
    }                                                                                      \
  } while (0)

/* The translation unit of a user simulator and the options it is compiled with: lane-group shape (L lanes x C components; the whole
 * row in one thread up to 16 parameters -- abz_user_dist --, 8 components per lane beyond -- abz_user_dist_lanes), PLAIN (every dimension
 * a continuous Normal: the two-instruction log-density of the built-in kernels), blobs.  A function of the model alone -- no device is
 * touched --, so the CPU tests compile the very same text with hipcc (abcdez_user_translation_unit, tests/test_user_simulator_sources.py). */
/* the shapes the two-launch sweep exists for: one lane per particle; a user-supplied simulator on rows of 4, 8 or 16 doubles (3 to 16
 * parameters), the built-in Lotka-Volterra simulator on rows of 4 */
static bool abz_jit_split_shape(int sim_id, int L, int C) {
  return L == 1 && ((sim_id == ABZ_SIM_USER && (C == 4 || C == 8 || C == 16)) || (sim_id == ABZ_SIM_LV && C == 4));
}
/* threads per workgroup of a user simulator's second launch: the staged form keeps a workgroup's proposals and their carried state in
 * LDS (abz_user_rounds.h: 8 C + 8 ABZ_USER_STATE + 44 bytes per proposal), which at 16 doubles per row fits for 128 of them */
static unsigned abz_jit_user_p2_block(int C) { return C == 16 ? 128u : (unsigned)ABZ_BLOCK; }

static int abz_jit_make_tu(int sim_id, int L, int C, bool plain, bool has_blob, bool wrap, const char* user_source, std::string& tu,
                           std::vector<std::string>& defs) {
  const bool user = sim_id == ABZ_SIM_USER;
  if (user && !user_source) { abz_set_error("user simulator: no source"); return -1; }
  if (user && L > 1 && !strstr(user_source, "abz_user_dist_lanes")) {
    abz_set_error("user simulator: rows of more than 16 parameters are spread over the lanes of a wavefront -- the source must define the "
                  "cooperative form abz_user_dist_lanes (include/abcdez_hip.h)");
    return -1;
  }
  if (user && L == 1 && !strstr(user_source, "abz_user_dist") && !strstr(user_source, "abz_user_round")) {
    abz_set_error("user simulator: the source must define abz_user_dist (or the staged abz_user_round)");
    return -1;
  }
  tu = "#include \"abz_kernels.h\"\n";
  if (user) {
    tu += "#line 1 \"user_simulator\"\n";
    tu += user_source;
    /* behind the user's text, so that it sees the ABZ_USER_ROUNDS / ABZ_USER_STATE the text may define: the staged form's abz_user_dist
     * and its round-by-round second launch */
    tu += "\n#line 1 \"abz_user_entry\"\n#include \"abz_user_rounds.h\"\n";
  }
  /* ABZ_JIT_SIM: the simulator these kernels are made for -- ABZ_SIM_USER, or a built-in one whose model has prior factors of the
   * wrapper families (the statically compiled sweeps do not carry those: include/abcdez_spec.h, ABZ_PRIOR_WRAP) */
  if (user)
    tu += "extern \"C\" __global__ __launch_bounds__(ABZ_BLOCK) void abz_user_init(const HotModel M, double* theta, "
          "double* logpi, double* delta, uint32_t i0, uint32_t n, unsigned long long* bad, uint64_t* stamp) {\n"
          "  init_kernel_body<ABZ_JIT_SIM, ABZ_USER_L, ABZ_USER_C>(M, theta, logpi, delta, i0, n, bad, stamp);\n}\n";
  tu += "extern \"C\" __global__ __launch_bounds__((abz_sweep_block<ABZ_JIT_SIM, ABZ_USER_L, ABZ_USER_C>())) void abz_user_smc_packed(const SmcPackedArgs a) {\n"
        "  smc_swarm_packed_body<ABZ_JIT_SIM, ABZ_USER_L, ABZ_USER_C, ABZ_USER_PLAIN != 0>(a);\n}\n"
        "extern \"C\" __global__ __launch_bounds__(ABZ_BLOCK) void abz_user_mc(const McSwarmArgs a) {\n"
        "  mc_swarm_kernel_body<ABZ_JIT_SIM, ABZ_USER_L, ABZ_USER_C, ABZ_USER_PLAIN != 0>(a);\n}\n"
        "extern \"C\" __global__ __launch_bounds__(ABZ_BLOCK) void abz_user_replay(const SmcReplayPackedArgs a) {\n"
        "  smc_replay_packed_body<ABZ_USER_L, ABZ_USER_C, ABZ_USER_PLAIN != 0>(a);\n}\n";
  const bool split = abz_jit_split_shape(sim_id, L, C);
  if (split && user)
    tu += "extern \"C\" __global__ __launch_bounds__(ABZ_BLOCK) void abz_user_smc_p1(const SmcPackedArgs a, const LvHandList h) {\n"
          "  smc_split_phase1_body<ABZ_SIM_USER, ABZ_USER_C, ABZ_USER_PLAIN != 0>(a, h);\n}\n"
          "#ifdef ABZ_USER_ROUNDS\n"      /* the staged form: round by round, leavers dropped (abz_user_rounds.h) */
          "extern \"C\" __global__ __launch_bounds__(ABZ_USER_P2_BLOCK) void abz_user_smc_p2(const SmcPackedArgs a, const LvHandList h) {\n"
          "  smc_user_rounds_phase2_body<ABZ_USER_C, ABZ_USER_PLAIN != 0, ABZ_USER_P2_BLOCK>(a, h);\n}\n"
          "extern \"C\" __global__ void abz_user_has_rounds() {}\n"
          "#else\n"
          "extern \"C\" __global__ __launch_bounds__(ABZ_USER_P2_BLOCK) void abz_user_smc_p2(const SmcPackedArgs a, const LvHandList h) {\n"
          "  smc_split_phase2_body<ABZ_SIM_USER, ABZ_USER_C, ABZ_USER_PLAIN != 0, ABZ_USER_P2_BLOCK>(a, h);\n}\n"
          "#endif\n";
  if (split && !user)                    /* the built-in Lotka-Volterra simulator: its own two launches (rounds with early exit) */
    tu += "extern \"C\" __global__ __launch_bounds__(ABZ_BLOCK) void abz_user_smc_p1(const SmcPackedArgs a, const LvHandList h) {\n"
          "  smc_lv_phase1_body<ABZ_USER_PLAIN != 0, ABZ_BLOCK>(a, h);\n}\n"
          "extern \"C\" __global__ __launch_bounds__(ABZ_LV_BLOCK2) void abz_user_smc_p2(const SmcPackedArgs a, const LvHandList h) {\n"
          "  smc_lv_phase2_body<ABZ_USER_PLAIN != 0, ABZ_LV_BLOCK2>(a, h);\n}\n";
  if (has_blob && L != 1 && user) { abz_set_error("user simulator: blobs need the whole row in one thread (length(prior) <= 16)"); return -1; }
  if (has_blob && user)
    tu += "extern \"C\" __global__ __launch_bounds__(ABZ_BLOCK) void abz_user_blob_eval(const HotModel M, const double* theta, "
          "const uint64_t* stamp, uint32_t n, double* blob, double* delta_out, uint32_t nbw) {\n"
          "  blob_eval_kernel_body<ABZ_SIM_USER, 1, ABZ_USER_C>(M, theta, stamp, n, blob, delta_out, nbw);\n}\n";
  defs = {"-DABZ_USER_C=" + std::to_string(C), "-DABZ_USER_L=" + std::to_string(L), std::string("-DABZ_USER_PLAIN=") + (plain ? "1" : "0"),
          "-DABZ_JIT_SIM=" + std::to_string(sim_id)};
  if (has_blob && user) defs.push_back("-DABZ_USER_HAS_BLOB=1");
  if (wrap) defs.push_back("-DABZ_PRIOR_WRAP=1");
  if (split && user) defs.push_back("-DABZ_USER_P2_BLOCK=" + std::to_string(abz_jit_user_p2_block(C)));
  return 0;
}

int abz_jit_build(abcdez_ctx* ctx, const char* user_source) {
  const int L = ctx->L, C = ctx->C, sim_id = ctx->h_model.sim_id;
  const bool user = sim_id == ABZ_SIM_USER;
  const bool split = abz_jit_split_shape(sim_id, L, C);
  const bool has_blob = ctx->h_model.n_blob > 0;     /* then a user source must also define abz_user_blob */
  std::string tu;
  std::vector<std::string> defs;
  if (int rc = abz_jit_make_tu(sim_id, L, C, ctx->prior_plain, has_blob, ctx->h_model.n_ext > 0, user_source, tu, defs)) return rc;
  hipDeviceProp_t prop;
  ABZ_HIP_CHECK(hipGetDeviceProperties(&prop, ctx->device));
  std::string arch = std::string("--offload-arch=") + prop.gcnArchName;
  const size_t colon = arch.find(':');                 /* "gfx950:sramecc+:xnack-" -> "gfx950" */
  if (colon != std::string::npos) arch = arch.substr(0, colon);
  /* the same text for the same architecture and options is the same code object: a process that creates many contexts of one model
   * (one per run, as both hosts do) compiles it once -- hiprtc takes 0.5-0.7 s per call (profiles/r06_user_lv_ab.jsonl) */
  std::string key = arch;
  for (const std::string& d : defs) key += " " + d;
  key += "\n" + tu;
  std::vector<char> code;
  {
    std::lock_guard<std::mutex> lock(g_code_mutex);
    auto it = g_code_cache.find(key);
    if (it != g_code_cache.end()) code = it->second;
  }
  if (code.empty()) {
  hiprtcProgram prog;
  ABZ_RTC_CHECK(hiprtcCreateProgram(&prog, tu.c_str(), "abz_user.hip", abz_jit_n_headers, (const char**)abz_jit_header_sources,
                                    (const char**)abz_jit_header_names));
  std::vector<const char*> opts = {arch.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math"};
  for (const std::string& d : defs) opts.push_back(d.c_str());
  const hiprtcResult cr = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
  if (cr != HIPRTC_SUCCESS) {
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    std::string log(ls, '\0');
    if (ls) hiprtcGetProgramLog(prog, &log[0]);
    hiprtcDestroyProgram(&prog);
    abz_set_error("user simulator does not compile:\n" + log);
    return -4;
  }
  size_t cs = 0;
  ABZ_RTC_CHECK(hiprtcGetCodeSize(prog, &cs));
  code.resize(cs);
  ABZ_RTC_CHECK(hiprtcGetCode(prog, code.data()));
  hiprtcDestroyProgram(&prog);
  std::lock_guard<std::mutex> lock(g_code_mutex);
  if (g_code_cache.size() >= 32) g_code_cache.clear();          /* a host that keeps generating new sources: start over */
  g_code_cache.emplace(key, code);
  }
  AbzUserModule* um = new AbzUserModule();
  ctx->user_module = um;            /* owned by the context from here on: abz_jit_destroy releases it on any failure below */
  ABZ_HIP_CHECK(hipModuleLoadData(&um->mod, code.data()));
  if (user) ABZ_HIP_CHECK(hipModuleGetFunction(&um->f_init, um->mod, "abz_user_init"));
  ABZ_HIP_CHECK(hipModuleGetFunction(&um->f_mc, um->mod, "abz_user_mc"));
  ABZ_HIP_CHECK(hipModuleGetFunction(&um->f_smcp, um->mod, "abz_user_smc_packed"));
  ABZ_HIP_CHECK(hipModuleGetFunction(&um->f_replay, um->mod, "abz_user_replay"));
  if (has_blob && user) ABZ_HIP_CHECK(hipModuleGetFunction(&um->f_blob, um->mod, "abz_user_blob_eval"));
  um->block_smc = (sim_id == ABZ_SIM_LV && L == 1 && C == 4) ? ABZ_LV_BLOCK : ABZ_BLOCK;      /* abz_sweep_block of the one-kernel body */
  um->block_p2 = user ? abz_jit_user_p2_block(C) : (unsigned)ABZ_LV_BLOCK2;
  if (split) {
    ABZ_HIP_CHECK(hipModuleGetFunction(&um->f_p1, um->mod, "abz_user_smc_p1"));
    ABZ_HIP_CHECK(hipModuleGetFunction(&um->f_p2, um->mod, "abz_user_smc_p2"));
    hipFunction_t marker = nullptr;       /* present iff the source defined ABZ_USER_ROUNDS (the staged form) */
    um->rounds = user && hipModuleGetFunction(&marker, um->mod, "abz_user_has_rounds") == hipSuccess;
    (void)hipGetLastError();
  }
  return 0;
}

void abz_jit_destroy(abcdez_ctx* ctx) {
  AbzUserModule* um = (AbzUserModule*)ctx->user_module;
  if (!um) return;
  if (um->mod) (void)hipModuleUnload(um->mod);
  delete um;
  ctx->user_module = nullptr;
}

int abz_jit_launch_init(abcdez_ctx* ctx, double* theta, double* logpi, double* delta, uint32_t i0, uint32_t n,
                        unsigned long long* bad, uint64_t* stamp) {
  AbzUserModule* um = (AbzUserModule*)ctx->user_module;
  HotModel M = ctx->hot;
  void* params[] = {&M, &theta, &logpi, &delta, &i0, &n, &bad, &stamp};
  const uint64_t tiles = ((uint64_t)n * (uint64_t)ctx->L + ABZ_BLOCK - 1) / ABZ_BLOCK;       /* L lanes per particle */
  ABZ_HIP_CHECK(hipModuleLaunchKernel(um->f_init, jit_grid(ctx, um->f_init, tiles), 1, 1, ABZ_BLOCK, 1, 1,
                                      0, ctx->stream, params, nullptr));
  return 0;
}
int abz_jit_launch_blob(abcdez_ctx* ctx, const double* theta, const uint64_t* stamp, uint32_t n, double* blob,
                        double* delta_out, uint32_t nbw) {
  AbzUserModule* um = (AbzUserModule*)ctx->user_module;
  if (!um || !um->f_blob) { abz_set_error("blob_eval: the user simulator was built without blobs (n_blob = 0)"); return -3; }
  HotModel M = ctx->hot;
  void* params[] = {&M, &theta, &stamp, &n, &blob, &delta_out, &nbw};
  ABZ_HIP_CHECK(hipModuleLaunchKernel(um->f_blob, jit_grid(ctx, um->f_blob, (n + ABZ_BLOCK - 1) / ABZ_BLOCK), 1, 1, ABZ_BLOCK, 1, 1,
                                      0, ctx->stream, params, nullptr));
  return 0;
}
int abz_jit_launch_mc(abcdez_ctx* ctx, const void* args, unsigned ntiles) {
  AbzUserModule* um = (AbzUserModule*)ctx->user_module;
  void* params[] = {const_cast<void*>(args)};
  ABZ_HIP_CHECK(hipModuleLaunchKernel(um->f_mc, jit_grid(ctx, um->f_mc, ntiles), 1, 1, ABZ_BLOCK, 1, 1, 0, ctx->stream, params, nullptr));
  return 0;
}
int abz_jit_launch_smc_packed(abcdez_ctx* ctx, const void* args, unsigned nblocks) {
  AbzUserModule* um = (AbzUserModule*)ctx->user_module;
  void* params[] = {const_cast<void*>(args)};
  ABZ_HIP_CHECK(hipModuleLaunchKernel(um->f_smcp, nblocks, 1, 1, um->block_smc, 1, 1, 0, ctx->stream, params, nullptr));
  return 0;
}
unsigned abz_jit_smc_block(abcdez_ctx* ctx) { return ((AbzUserModule*)ctx->user_module)->block_smc; }
bool abz_jit_has_replay(abcdez_ctx* ctx) {
  AbzUserModule* um = (AbzUserModule*)ctx->user_module;
  return um && um->f_replay;
}
int abz_jit_launch_replay(abcdez_ctx* ctx, const void* args, unsigned nblocks) {
  AbzUserModule* um = (AbzUserModule*)ctx->user_module;
  void* params[] = {const_cast<void*>(args)};
  ABZ_HIP_CHECK(hipModuleLaunchKernel(um->f_replay, nblocks, 1, 1, ABZ_BLOCK, 1, 1, 0, ctx->stream, params, nullptr));
  return 0;
}
/* the sweep as two launches: phase 1 over the positions, phase 2 over the hand-over list (one workgroup per ABZ_BLOCK possible records) */
bool abz_jit_has_smc_split(abcdez_ctx* ctx) {
  AbzUserModule* um = (AbzUserModule*)ctx->user_module;
  return um && um->f_p1 && um->f_p2;
}
bool abz_jit_has_rounds(abcdez_ctx* ctx) {
  AbzUserModule* um = (AbzUserModule*)ctx->user_module;
  return um && um->rounds;
}
int abz_jit_launch_smc_split(abcdez_ctx* ctx, const void* args, const void* list, unsigned nblocks) {
  AbzUserModule* um = (AbzUserModule*)ctx->user_module;
  void* params[] = {const_cast<void*>(args), const_cast<void*>(list)};
  ABZ_HIP_CHECK(hipModuleLaunchKernel(um->f_p1, nblocks, 1, 1, ABZ_BLOCK, 1, 1, 0, ctx->stream, params, nullptr));
  const unsigned nb2 = (unsigned)(((uint64_t)nblocks * ABZ_BLOCK + um->block_p2 - 1) / um->block_p2);       /* one workgroup per block_p2 possible records */
  ABZ_HIP_CHECK(hipModuleLaunchKernel(um->f_p2, nb2, 1, 1, um->block_p2, 1, 1, 0, ctx->stream, params, nullptr));
  return 0;
}

/* Test hook (no device needed): the translation unit abcdez_ctx_create_user would hand to hiprtc for this model and source, and the -D
 * options that go with it (one per line).  Returns the text's length, or a negative status. */
extern "C" __attribute__((visibility("default"))) int abcdez_user_translation_unit(const abz_model* model, const char* user_source,
                                                                                     char* tu_out, size_t tu_cap, char* opts_out, size_t opts_cap) {
  if (!model || !tu_out || !opts_out || (model->sim_id == ABZ_SIM_USER && !user_source)) { abz_set_error("user_translation_unit: null argument"); return -1; }
  int L = 1, C = model->ld;                                   /* default_shape of abz_api.hip */
  if (model->sim_id == ABZ_SIM_USER && model->ld > 64) { L = 8; C = model->ld / 8; }
  else if (model->sim_id == ABZ_SIM_USER && model->ld > 16) { C = 8; L = model->ld / 8; }
  if (model->sim_id == ABZ_SIM_MVN && model->ld > 64) { L = 8; C = model->ld / 8; }
  else if (model->sim_id == ABZ_SIM_MVN && model->ld > 8) { C = 8; L = model->ld / 8; }
  bool plain = model->d == model->ld && !model->mv;
  for (int k = 0; k < model->d && plain; ++k) plain = model->prior[k].family == ABZ_PRIOR_NORMAL && !model->prior[k].discrete;
  std::string tu;
  std::vector<std::string> defs;
  if (int rc = abz_jit_make_tu(model->sim_id, L, C, plain, model->n_blob > 0, model->n_ext > 0, user_source, tu, defs)) return rc;
  std::string o;
  for (const std::string& d : defs) o += d + "\n";
  if (tu.size() + 1 > tu_cap || o.size() + 1 > opts_cap) { abz_set_error("user_translation_unit: buffer too small"); return -1; }
  memcpy(tu_out, tu.c_str(), tu.size() + 1);
  memcpy(opts_out, o.c_str(), o.size() + 1);
  return (int)tu.size();
}

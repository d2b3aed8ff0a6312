// What a small kernel between two dependent launches costs on this part, and whether a captured graph makes it cheaper:
// a chain of K dependent launches of a tiny kernel (G workgroups, each adds 1 to its word) timed (a) as plain stream launches,
// (b) as one hipGraph captured from the same stream calls and replayed.
//   hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip && ./launch_floor
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void touch(unsigned* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1u; }
int main() {
  unsigned* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int K = 16, REP = 200;
  for (int G : {1, 256, 4096}) {
    for (int w = 0; w < 50; ++w) hipLaunchKernelGGL(touch, dim3(G), dim3(256), 0, st, d);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < REP; ++r) for (int k = 0; k < K; ++k) hipLaunchKernelGGL(touch, dim3(G), dim3(256), 0, st, d);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_stream = ms * 1e3 / (REP * K);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int k = 0; k < K; ++k) hipLaunchKernelGGL(touch, dim3(G), dim3(256), 0, st, d);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 10; ++w) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_graph = ms * 1e3 / (REP * K);
    /* (c) every launch bracketed by a pair of hipEventRecord; (d) hipExtLaunchKernelGGL with the pair handed to the launch */
    static hipEvent_t ev[2 * 16];
    static bool have = false;
    if (!have) { for (int q = 0; q < 32; ++q) CK(hipEventCreate(&ev[q])); have = true; }
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < REP; ++r) for (int k = 0; k < K; ++k) {
      CK(hipEventRecord(ev[2 * k], st));
      hipLaunchKernelGGL(touch, dim3(G), dim3(256), 0, st, d);
      CK(hipEventRecord(ev[2 * k + 1], st));
    }
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_pairs = ms * 1e3 / (REP * K);
    float in_pair = 0; CK(hipEventElapsedTime(&in_pair, ev[0], ev[1]));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < REP; ++r) for (int k = 0; k < K; ++k)
      hipExtLaunchKernelGGL(touch, dim3(G), dim3(256), 0, st, ev[2 * k], ev[2 * k + 1], 0, d);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_ext = ms * 1e3 / (REP * K);
    float in_ext = 0; CK(hipEventElapsedTime(&in_ext, ev[0], ev[1]));
    printf("{\"workgroups\": %d, \"chain\": %d, \"us_per_kernel_stream\": %.3f, \"us_per_kernel_graph\": %.3f, "
           "\"us_per_kernel_with_event_pair\": %.3f, \"event_pair_reads_us\": %.3f, \"us_per_kernel_ext_launch_with_events\": %.3f, "
           "\"ext_events_read_us\": %.3f}\n", G, K, us_stream, us_graph, us_pairs, in_pair * 1e3, us_ext, in_ext * 1e3);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}

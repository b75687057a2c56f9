// launch_chain.hip — what a dependent kernel boundary costs on this box (no profiler attached): N trivial kernels back to back on
// one stream, eager vs captured hipGraph, for 1 / 16 / 256 workgroups; plus the same chain with a 5 us body.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/launch_chain tools/probe/launch_chain.hip && tools/probe/launch_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void tiny(float* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}
__global__ void body(float* p, int n, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = i < n ? p[i] : 0.f;
  for (int k = 0; k < iters; ++k) v = v * 1.0001f + 0.5f;
  if (i < n) p[i] = v;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  float* d;
  hipMalloc(&d, 1 << 24);
  hipMemset(d, 0, 1 << 24);
  hipStream_t s;
  hipStreamCreate(&s);
  const int N = 200;
  for (int wgs : {1, 16, 256, 1024}) {
    for (int iters : {0, 2000}) {
      auto chain = [&]() {
        for (int i = 0; i < N; ++i) {
          if (iters) hipLaunchKernelGGL(body, dim3(wgs), dim3(256), 0, s, d, wgs * 256, iters);
          else hipLaunchKernelGGL(tiny, dim3(wgs), dim3(256), 0, s, d, wgs * 256);
        }
      };
      chain();
      hipStreamSynchronize(s);
      double best = 1e9;
      for (int r = 0; r < 5; ++r) {
        const double t0 = now();
        chain();
        hipStreamSynchronize(s);
        best = std::min(best, now() - t0);
      }
      hipGraph_t g;
      hipGraphExec_t ge;
      hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
      chain();
      hipStreamEndCapture(s, &g);
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      hipGraphLaunch(ge, s);
      hipStreamSynchronize(s);
      double bestg = 1e9;
      for (int r = 0; r < 5; ++r) {
        const double t0 = now();
        hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        bestg = std::min(bestg, now() - t0);
      }
      // one kernel alone, for the body time
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0, s);
      if (iters) hipLaunchKernelGGL(body, dim3(wgs), dim3(256), 0, s, d, wgs * 256, iters);
      else hipLaunchKernelGGL(tiny, dim3(wgs), dim3(256), 0, s, d, wgs * 256);
      hipEventRecord(e1, s);
      hipStreamSynchronize(s);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      printf("wgs %4d body_iters %4d : eager %.2f us/kernel, graph %.2f us/kernel (chain of %d; single kernel event-timed %.2f us)\n", wgs,
             iters, best / N * 1e6, bestg / N * 1e6, N, ms * 1e3);
      hipGraphExecDestroy(ge);
      hipGraphDestroy(g);
    }
  }
  return 0;
}

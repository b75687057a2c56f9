// grid_barrier.hip — what an IN-KERNEL grid-wide hand-over costs on MI355X (8 XCDs, one L2 each), against the dependent kernel
// boundary it would replace.  A persistent kernel (one workgroup per CU) runs `iters` rounds of
//     write my slice of a [bytes] buffer  ->  make it visible to every XCD  ->  grid barrier  ->  read ANOTHER workgroup's slice
// (the reader sits on a different XCD by construction) and counts every value that is not the one the writer stored.
// Three ways to make the data visible:
//   FENCE : plain stores / loads, __threadfence()-style agent-scope release before the barrier and acquire after it
//           (buffer_wbl2 sc1 + buffer_inv sc1: what a kernel boundary does implicitly)
//   SC1   : every shared store / load is a relaxed agent-scope atomic (global_store/load ... sc1): the data itself bypasses the
//           non-coherent L2 lines, the barrier only waits for the stores (s_waitcnt) — no cache maintenance at all
//   NONE  : plain accesses, no fence (shows that the test can actually see stale data: errors expected)
// and the barrier alone (no data).  Printed: microseconds per round and the error count.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/grid_barrier tools/probe/grid_barrier.hip && tools/probe/grid_barrier
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

enum { MODE_FENCE = 0, MODE_SC1 = 1, MODE_NONE = 2, MODE_BARRIER_ONLY = 3 };

// one thread per workgroup arrives; monotone counter, round r waits for (r + 1) * G arrivals.  Bounded spin: a workgroup that
// would wait forever (not all co-resident) gives up and raises `*fail` instead of hanging the GPU.
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, unsigned* fail) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) { ok = false; __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
  return ok;
}

template <int MODE>
__global__ void __launch_bounds__(256) rounds_kernel(float* buf0, float* buf1, int per_wg, int iters, unsigned* counter,
                                                      unsigned* fail, unsigned* errors, int partner_shift) {
  const int G = gridDim.x, w = blockIdx.x;
  const int partner = (w + partner_shift) % G;       // consecutive workgroup ids go round-robin over the XCDs: shift not % 8 == another XCD
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    float* buf = (it & 1) ? buf1 : buf0;
    if (MODE != MODE_BARRIER_ONLY) {
      for (int i = threadIdx.x; i < per_wg; i += blockDim.x) {
        const float v = (float)(it * 7 + ((w * per_wg + i) & 1023));
        if (MODE == MODE_SC1) __hip_atomic_store(buf + (size_t)w * per_wg + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else buf[(size_t)w * per_wg + i] = v;
      }
      if (MODE == MODE_FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (MODE == MODE_SC1) __builtin_amdgcn_s_waitcnt(0);      // the stores have left the CU (vmcnt / vscnt 0)
    }
    if (!grid_barrier(counter, (unsigned)(it + 1) * G, fail)) return;
    if (MODE != MODE_BARRIER_ONLY) {
      if (MODE == MODE_FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      for (int i = threadIdx.x; i < per_wg; i += blockDim.x) {
        const float want = (float)(it * 7 + ((partner * per_wg + i) & 1023));
        float got;
        if (MODE == MODE_SC1) got = __hip_atomic_load(buf + (size_t)partner * per_wg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else got = buf[(size_t)partner * per_wg + i];
        bad += got != want;
      }
    }
  }
  if (bad) atomicAdd(errors, bad);
}

// the thing it would replace: the same write / read pair as two dependent kernels per round
__global__ void __launch_bounds__(256) write_kernel(float* buf, int per_wg, int it) {
  const int w = blockIdx.x;
  for (int i = threadIdx.x; i < per_wg; i += blockDim.x) buf[(size_t)w * per_wg + i] = (float)(it * 7 + ((w * per_wg + i) & 1023));
}
__global__ void __launch_bounds__(256) read_kernel(const float* buf, int per_wg, int it, unsigned* errors, int partner_shift) {
  const int G = gridDim.x, partner = (blockIdx.x + partner_shift) % G;
  unsigned bad = 0;
  for (int i = threadIdx.x; i < per_wg; i += blockDim.x) bad += buf[(size_t)partner * per_wg + i] != (float)(it * 7 + ((partner * per_wg + i) & 1023));
  if (bad) atomicAdd(errors, bad);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int MODE>
static void run(const char* name, int G, int per_wg, int iters, float* b0, float* b1, unsigned* ctl, hipStream_t s) {
  // ctl[0] counter, ctl[1] fail, ctl[2] errors
  auto once = [&]() {
    hipMemsetAsync(ctl, 0, 3 * sizeof(unsigned), s);
    hipLaunchKernelGGL(rounds_kernel<MODE>, dim3(G), dim3(256), 0, s, b0, b1, per_wg, iters, ctl, ctl + 1, ctl + 2, 37);
  };
  once();
  hipStreamSynchronize(s);
  double best = 1e9;
  unsigned h[3] = {0, 0, 0};
  for (int r = 0; r < 3; ++r) {
    const double t0 = now();
    once();
    hipStreamSynchronize(s);
    best = std::min(best, now() - t0);
    hipMemcpy(h, ctl, sizeof(h), hipMemcpyDeviceToHost);
    if (h[1]) break;
  }
  printf("  %-13s G %4d  %7.1f KB/round : %6.2f us/round   errors %u%s\n", name, G, G * per_wg * 4 / 1024.0, best / iters * 1e6, h[2],
         h[1] ? "   BARRIER TIMED OUT (workgroups not co-resident)" : "");
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("%s, %d CUs\n", prop.name, prop.multiProcessorCount);
  hipStream_t s;
  hipStreamCreate(&s);
  float *b0, *b1;
  unsigned* ctl;
  hipMalloc(&b0, 64 << 20);
  hipMalloc(&b1, 64 << 20);
  hipMalloc(&ctl, 256);
  hipMemset(b0, 0, 64 << 20);
  hipMemset(b1, 0, 64 << 20);
  const int iters = 400;
  for (int G : {128, 256, 512}) {
    printf("grid %d workgroups x 256 threads\n", G);
    run<MODE_BARRIER_ONLY>("barrier only", G, 0, iters, b0, b1, ctl, s);
    for (int kb : {288, 1152, 4608}) {           // a [192][384] fp32 activation, x4, x16
      const int per_wg = kb * 1024 / 4 / G;
      run<MODE_FENCE>("fence", G, per_wg, iters, b0, b1, ctl, s);
      run<MODE_SC1>("sc1 accesses", G, per_wg, iters, b0, b1, ctl, s);
      run<MODE_NONE>("no coherence", G, per_wg, iters, b0, b1, ctl, s);
      // two dependent kernels per round
      unsigned zero = 0;
      hipMemcpy(ctl + 2, &zero, 4, hipMemcpyHostToDevice);
      auto chain = [&]() {
        for (int it = 0; it < iters; ++it) {
          float* buf = (it & 1) ? b1 : b0;
          hipLaunchKernelGGL(write_kernel, dim3(G), dim3(256), 0, s, buf, per_wg, it);
          hipLaunchKernelGGL(read_kernel, dim3(G), dim3(256), 0, s, buf, per_wg, it, ctl + 2, 37);
        }
      };
      chain();
      hipStreamSynchronize(s);
      double best = 1e9;
      for (int r = 0; r < 3; ++r) {
        const double t0 = now();
        chain();
        hipStreamSynchronize(s);
        best = std::min(best, now() - t0);
      }
      unsigned e = 0;
      hipMemcpy(&e, ctl + 2, 4, hipMemcpyDeviceToHost);
      printf("  %-13s G %4d  %7.1f KB/round : %6.2f us/round   errors %u   (= 2 dependent launches)\n", "two kernels", G, G * per_wg * 4 / 1024.0,
             best / iters * 1e6, e);
    }
  }
  return 0;
}

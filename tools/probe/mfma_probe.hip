// mfma_probe.hip — microbenchmark behind the fp32 conv kernel's tiling decisions (tools/, not product code):
// how many cycles per v_mfma_f32_32x32x2_f32 does ONE SIMD sustain, as a function of (waves per SIMD, independent
// accumulators per wave, filler instructions between MFMAs)?   hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int FILL>
__global__ void __launch_bounds__(256) probe(float* out, const float* in, int iters, long long* cyc) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 256) lds[i] = in[i];
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float av = in[tid], bv = in[tid + 256];
  const float* lp = lds + (tid & 63);
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float b = bv;
      if (FILL >= 1) b = lp[(u * 67 + it) & 4031];              // one ds_read_b32 feeding the MFMA (like the conv's B operand)
      if (FILL >= 2) av = av * 1.0001f + 0.5f;                   // some VALU
      acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[u % NACC], 0, 0, 0);
    }
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int FILL>
void run(int wg_per_cu, float* out, float* in, long long* cyc) {
  const int iters = 2000, nwg = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  // dynamic LDS sized so that exactly wg_per_cu workgroups fit a CU (160 KB)
  size_t lds = wg_per_cu == 1 ? 96 * 1024 : (wg_per_cu == 2 ? 72 * 1024 : (wg_per_cu == 3 ? 50 * 1024 : 36 * 1024));
  auto k = probe<NACC, FILL>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, out, in, 10, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, out, in, iters, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(nwg);
  hipMemcpy(h.data(), cyc, sizeof(long long) * nwg, hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto v : h) avg += (double)v;
  avg /= nwg;
  const double mfma_per_simd = (double)iters * 8 * wg_per_cu;   // each WG puts one wave on each SIMD
  const double tf = 2.0 * 32 * 32 * 2 * (double)iters * 8 * 4 * nwg / (ms * 1e-3) / 1e12;
  printf("waves/SIMD %d  acc/wave %d  fill %d : %7.1f wave-cycles per MFMA per SIMD (clock64 ticks/%0.0f), %6.1f TF, %.3f ms\n",
         wg_per_cu, NACC, FILL, avg / mfma_per_simd, 1.0, tf, ms);
}

int main() {
  float *out, *in; long long* cyc;
  hipMalloc(&out, 4 << 20); hipMalloc(&in, 1 << 20); hipMalloc(&cyc, 1 << 16);
  hipMemset(in, 0, 1 << 20);
  for (int w = 1; w <= 4; ++w) {
    run<1, 0>(w, out, in, cyc); run<2, 0>(w, out, in, cyc); run<4, 0>(w, out, in, cyc);
    run<1, 1>(w, out, in, cyc); run<2, 1>(w, out, in, cyc); run<4, 1>(w, out, in, cyc);
    run<1, 2>(w, out, in, cyc); run<2, 2>(w, out, in, cyc);
  }
  return 0;
}

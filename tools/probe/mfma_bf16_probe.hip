// mfma_bf16_probe.hip — microbenchmark behind conv_x6.hip's tiling (tools/, not product code): what fraction of the 2.5 PF bf16 MFMA
// peak does the chip sustain on v_mfma_f32_32x32x16_bf16 as a function of (workgroups of 4 waves per CU, independent accumulators per
// wave, operand traffic per MFMA)?   hipcc --offload-arch=gfx950 -O3 mfma_bf16_probe.hip -o mfma_bf16_probe
//   FILL 0: registers only        FILL 1: + one ds_read_b128 per 2 MFMAs (the x6 unit: 6 reads per 12 MFMAs)
//   FILL 2: FILL 1 + one global_load_dwordx4 per 4 MFMAs (3 per unit, L2-resident 1.5 MB stream)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int FILL>
__global__ void __launch_bounds__(256) probe(float* out, const u32x4* in, int iters) {
  extern __shared__ u32x4 lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 2048; i += 256) lds[i] = in[i];
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  u32x4 av[3], bv[6];
  for (int i = 0; i < 3; ++i) av[i] = in[tid + 256 * i];
  for (int i = 0; i < 6; ++i) bv[i] = in[tid + 256 * (3 + i)];
  const u32x4* lp = lds + (tid & 63) * 5;           // 80-byte pitch like the x6 tile
  const u32x4* gp = in + (blockIdx.x % 64) * 1536 + (tid & 63);
  for (int it = 0; it < iters; ++it) {
    u32x4 bn[6], an[3];
    if (FILL >= 1) {
#pragma unroll
      for (int i = 0; i < 6; ++i) bn[i] = lp[(i * 331 + (it & 7) * 40) & 1023];
    }
#pragma unroll
    for (int u = 0; u < 12; ++u)
      acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[u % 3]), __builtin_bit_cast(bf16x8, bv[u % 6]),
                                                              acc[u % NACC], 0, 0, 0);
    if (FILL >= 2) {
#pragma unroll
      for (int i = 0; i < 3; ++i) an[i] = gp[((it & 255) * 3 + i) * 64];
    }
    __builtin_amdgcn_sched_group_barrier(0x100, FILL >= 1 ? 6 : 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, FILL >= 2 ? 3 : 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (FILL >= 1) {
#pragma unroll
      for (int i = 0; i < 6; ++i) bv[i] = bn[i];
    }
    if (FILL >= 2) {
#pragma unroll
      for (int i = 0; i < 3; ++i) av[i] = an[i];
    }
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int NACC, int FILL>
void run(int wg_per_cu, float* out, u32x4* in) {
  const int iters = 3000, nwg = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  size_t lds = wg_per_cu == 1 ? 96 * 1024 : (wg_per_cu == 2 ? 72 * 1024 : (wg_per_cu == 3 ? 50 * 1024 : 36 * 1024));
  auto k = probe<NACC, FILL>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, out, in, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, out, in, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)nwg * 4 * iters * 12 * 2.0 * 32 * 32 * 16;
  const double tf = flops / (ms * 1e-3) / 1e12;
  // cycles per MFMA per SIMD if the clock were the nominal 2.4 GHz
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 12 * wg_per_cu);
  printf("  waves/SIMD %d  acc %d  fill %d : %8.1f TF  (%.3f of 2.5 PF)  %.1f nominal cycles per MFMA per SIMD  [%.3f ms]\n", wg_per_cu, NACC, FILL,
         tf, tf / 2500.0, cyc, ms);
}

int main() {
  float* out; u32x4* in;
  hipMalloc(&out, 4 * 256 * 1024 * 4);
  hipMalloc(&in, 4 << 20);
  std::vector<unsigned> h((4 << 20) / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c003c00u + (unsigned)(i * 2654435761u & 0x007f007fu);   // small bf16 pairs
  hipMemcpy(in, h.data(), 4 << 20, hipMemcpyHostToDevice);
  for (int w = 1; w <= 3; ++w) {
    run<1, 0>(w, out, in); run<2, 0>(w, out, in); run<4, 0>(w, out, in);
    run<2, 1>(w, out, in); run<4, 1>(w, out, in);
    run<2, 2>(w, out, in); run<4, 2>(w, out, in);
  }
  return 0;
}

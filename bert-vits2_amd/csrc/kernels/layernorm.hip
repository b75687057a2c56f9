// layernorm.hip — LayerNorm over the CHANNEL dim of a [B,C,T] tensor (reference modules.py:26-29 / attentions.py:21-24:
// transpose -> F.layer_norm(eps=1e-5, biased variance) -> transpose), with the surrounding elementwise work fused:
//   * the partial slabs of a split-K convolution summed in front (conv_mfma.hip), plus the residual sum
//     (Encoder: norm(x + y), attentions.py:114,118)
//   * depthwise k=3 dilated conv in front (DDSConv.convs_sep, modules.py:122)
//   * exact-erf GELU, residual add, per-batch speaker vector add, and sequence mask behind
//     (modules.py:124-129, attentions.py:107-111,119).
// Layout: a workgroup owns 16 consecutive time steps x all C channels (small tiles: at batch 1 the tensor is ~300 KB and
// the kernel is pure latency, so it is cut into as many workgroups as keeps 64-byte row segments); thread (tx = t,
// ty = channel group of 16) keeps its C/16 values in registers, so the input is read once and the statistics are a
// two-pass (mean, then centred sum of squares) reduction over registers + one LDS exchange between the 16 groups.
#include <hip/hip_runtime.h>
#include "../bv2_kernels.h"

namespace bv2 {

constexpr int LN_TT = 16;      // time steps per workgroup
constexpr int LN_G = 16;       // channel groups (threads along C)
constexpr int LN_MAXCPT = 16;  // channels per thread (C <= 256)

__global__ void __launch_bounds__(256) layernorm_kernel(const LnArgs A) {
  __shared__ float red[LN_G][LN_TT + 1];
  __shared__ float stat[LN_TT];
  const int tx = threadIdx.x & (LN_TT - 1), ty = threadIdx.x >> 4;
  const int b = blockIdx.y;
  const int t = blockIdx.x * LN_TT + tx;
  const bool tok = t < A.T;
  const int64_t base = (int64_t)b * A.C * A.T;
  const int nslab = A.nslab < 1 ? 1 : A.nslab;

  float v[LN_MAXCPT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXCPT; ++i) {
    v[i] = 0.f;
    const int c = ty + i * LN_G;
    if (c < A.C && tok) {
      const int64_t off = base + (int64_t)c * A.T + t;
      float x;
      if (A.mode == 0) {
        x = A.a[off];
        for (int sl = 1; sl < nslab; ++sl) x += A.a[(int64_t)sl * A.slab_stride + off];
        if (A.add) x += A.add[off];
      } else {
        x = A.dwb[c];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int tt = t + (j - 1) * A.dil;
          if (tt >= 0 && tt < A.T) {
            float xv = A.a[base + (int64_t)c * A.T + tt];
            if (A.in_mask) xv *= A.in_mask[(int64_t)b * A.T + tt];
            x += A.dww[c * 3 + j] * xv;
          }
        }
      }
      v[i] = x;
      s += x;
    }
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0) {
    float m = 0.f;
#pragma unroll
    for (int g = 0; g < LN_G; ++g) m += red[g][tx];
    stat[tx] = m / (float)A.C;
  }
  __syncthreads();
  const float mean = stat[tx];
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXCPT; ++i) {
    const int c = ty + i * LN_G;
    if (c < A.C) {
      const float d = v[i] - mean;
      q += d * d;
    }
  }
  __syncthreads();
  red[ty][tx] = q;
  __syncthreads();
  if (ty == 0) {
    float m = 0.f;
#pragma unroll
    for (int g = 0; g < LN_G; ++g) m += red[g][tx];
    stat[tx] = 1.0f / sqrtf(m / (float)A.C + A.eps);
  }
  __syncthreads();
  const float rstd = stat[tx];
  if (!tok) return;
  const float mk = A.mask ? A.mask[(int64_t)b * A.T + t] : 1.f;
#pragma unroll
  for (int i = 0; i < LN_MAXCPT; ++i) {
    const int c = ty + i * LN_G;
    if (c < A.C) {
      const int64_t off = base + (int64_t)c * A.T + t;
      float y = (v[i] - mean) * rstd * A.gamma[c] + A.beta[c];
      if (A.post_gelu) y = 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f));
      if (A.res) y += A.res[off];
      if (A.vec) y += A.vec[(int64_t)b * A.vec_bstride + c];
      A.out[off] = y * mk;
    }
  }
}

int launch_layernorm(hipStream_t stream, const LnArgs& a) {
  if (a.C > LN_G * LN_MAXCPT || a.C < 1 || a.T < 1 || a.B < 1 || a.nslab > BV2_MAX_KSPLIT) return -1;
  dim3 grid((a.T + LN_TT - 1) / LN_TT, a.B);
  hipLaunchKernelGGL(layernorm_kernel, grid, dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace bv2

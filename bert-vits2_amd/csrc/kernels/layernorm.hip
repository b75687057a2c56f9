// layernorm.hip — LayerNorm over the CHANNEL dim of a [B,C,T] tensor (reference modules.py:26-29 / attentions.py:21-24:
// transpose -> F.layer_norm(eps=1e-5, biased variance) -> transpose), with the surrounding elementwise work fused:
//   * the partial slabs of a split-K convolution summed in front (conv_mfma.hip), plus the residual sum
//     (Encoder: norm(x + y), attentions.py:114,118)
//   * depthwise k=3 dilated conv in front (DDSConv.convs_sep, modules.py:122)
//   * exact-erf GELU, residual add, per-batch speaker vector add, and sequence mask behind
//     (modules.py:124-129, attentions.py:107-111,119).
// Latency-first layout (at batch 1 the tensor is ~300 KB): a workgroup owns only 8 consecutive time steps x all C
// channels; thread (tx = t, ty = channel group of 32) keeps its C/32 values in registers.  All loads are unconditional
// (clamped addresses) and the slab / channel counts are template parameters, so every load of a thread is in flight
// at once: ONE memory round trip.  The channel reduction is wavefront-level: a wave holds 8 time steps x 8 channel groups
// (lane = tx + 8*cg), so each of the two passes (mean, then the centred sum of squares — the exact two-pass variance) is
// three `__shfl_xor` butterflies (lane ^ 8, ^ 16, ^ 32) plus ONE 4-entry exchange between the workgroup's four waves
// through LDS: two barriers per kernel, no serial sums.
#include <hip/hip_runtime.h>
#include "../bv2_kernels.h"

namespace bv2 {

constexpr int LN_TT = 8;       // time steps per workgroup
constexpr int LN_G = 32;       // channel groups (threads along C)

template <int CPT, int NSLAB, int MODE, bool GUARD, int KS = 0>   // CPT channels per thread; GUARD: C < CPT*LN_G allowed; KS > 0: weighted slabs, KS key ranges per head
__global__ void __launch_bounds__(256) layernorm_kernel(const LnArgs A) {
  __shared__ float red[2][4][LN_TT];               // [pass][wave][time step]
  const int tx = threadIdx.x & (LN_TT - 1), ty = threadIdx.x >> 3;
  int b = blockIdx.y, tile = blockIdx.x;
  if (A.xcd_b && !xcd_decode(blockIdx.x, A.xcd_per, A.B, b, tile)) return;   // batch item -> XCD affinity (bv2_kernels.h)
  if (!A.xcd_b && A.pf.ptr) {                                                // B == 1: tiles padded to a multiple of 8, then PF_BLOCKS spare workgroups
    const unsigned nt8 = ((unsigned)(A.T + LN_TT - 1) / LN_TT + 7u) & ~7u;
    if (blockIdx.x >= nt8) { prefetch_tail(A.pf, blockIdx.x, blockIdx.x - nt8, threadIdx.x, 256); return; }
    if (tile * LN_TT >= A.T) return;
  }
  const int t = tile * LN_TT + tx;
  const bool tok = t < A.T;
  const int tcl = tok ? t : A.T - 1;
  const int C = A.C, T = A.T;
  const int64_t base = (int64_t)b * C * T;
  const float* ap = A.a + base;

  float v[CPT];
  if constexpr (MODE == 0 && KS > 0) {
    // key-split attention (kernels/attention.hip): slab h*ks + r carries head h's output over key range r, normalised over that range
    // only, already behind conv_o; the flash-decoding merge weights come from the ranges' (max, sum) pairs of this column
    const float* mlp = A.ml + (int64_t)b * NSLAB * 2 * T + tcl;
    float mm[NSLAB], ll[NSLAB], w[NSLAB];
#pragma unroll
    for (int sl = 0; sl < NSLAB; ++sl) { mm[sl] = mlp[(int64_t)sl * 2 * T]; ll[sl] = mlp[(int64_t)sl * 2 * T + T]; }
    float xs[CPT][NSLAB];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      int c = ty + i * LN_G;
      if (GUARD) c = c < C ? c : C - 1;
      const int off = c * T + tcl;
#pragma unroll
      for (int sl = 0; sl < NSLAB; ++sl) xs[i][sl] = ap[(int64_t)sl * A.slab_stride + off];
    }
#pragma unroll
    for (int h0 = 0; h0 < NSLAB; h0 += KS) {        // h0 = first slab of a head
      float M = mm[h0];
#pragma unroll
      for (int r = 1; r < KS; ++r) M = fmaxf(M, mm[h0 + r]);
      float den = 0.f;
#pragma unroll
      for (int r = 0; r < KS; ++r) { w[h0 + r] = ll[h0 + r] * __expf(mm[h0 + r] - M); den += w[h0 + r]; }
      const float inv = 1.0f / den;
#pragma unroll
      for (int r = 0; r < KS; ++r) w[h0 + r] *= inv;
    }
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      int c = ty + i * LN_G;
      if (GUARD) c = c < C ? c : C - 1;
      float x = A.bias ? A.bias[c] : 0.f;
#pragma unroll
      for (int sl = 0; sl < NSLAB; ++sl) x += w[sl] * xs[i][sl];
      v[i] = x;
    }
    if (A.add) {
      const float* dp = A.add + base;
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        int c = ty + i * LN_G;
        if (GUARD) c = c < C ? c : C - 1;
        v[i] += dp[c * T + tcl];
      }
    }
  } else if constexpr (MODE == 0) {
    // every slab element AND the residual of this thread are loaded before the first add: written as `x += slab[sl]` the compiler
    // issued the loads in batches of 16 with a full s_waitcnt between them (ISA of round 2: 3 serial round trips for 8 slabs, a 4th
    // for the residual) — most of this kernel's life, which is one memory latency otherwise
    float xs[CPT][NSLAB], ad[CPT];
    const float* dp = (A.add ? A.add : A.a) + base;         // no residual: any valid address, result unused
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      int c = ty + i * LN_G;
      if (GUARD) c = c < C ? c : C - 1;
      const int off = c * T + tcl;
#pragma unroll
      for (int sl = 0; sl < NSLAB; ++sl) xs[i][sl] = ap[(int64_t)sl * A.slab_stride + off];
      ad[i] = dp[off];
    }
    const bool has_add = A.add != nullptr;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      float x = xs[i][0];
#pragma unroll
      for (int sl = 1; sl < NSLAB; ++sl) x += xs[i][sl];
      v[i] = has_add ? x + ad[i] : x;
    }
  } else {
    // depthwise k=3 conv with dilation: taps at t-dil, t, t+dil; zero padding; input pre-multiplied by in_mask
    float mk3[3];
    int tt3[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int tt = t + (j - 1) * A.dil;
      const bool ok = tok && tt >= 0 && tt < T;
      tt3[j] = ok ? tt : tcl;
      mk3[j] = ok ? (A.in_mask ? A.in_mask[(int64_t)b * T + tt3[j]] : 1.f) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      int c = ty + i * LN_G;
      if (GUARD) c = c < C ? c : C - 1;
      float x = A.dwb[c];
#pragma unroll
      for (int j = 0; j < 3; ++j) x += A.dww[c * 3 + j] * (ap[c * T + tt3[j]] * mk3[j]);
      v[i] = x;
    }
  }
  // ---- everything the epilogue needs goes in flight NOW, under the reductions: loaded after the second barrier (where the
  // compiler leaves them) gamma / beta / residual / speaker vector / mask were a second exposed memory round trip per launch
  float gm[CPT], bt[CPT], rr[CPT], vc[CPT], vc2[CPT];
  const float mk = (A.mask && tok) ? A.mask[(int64_t)b * T + t] : 1.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    int c = ty + i * LN_G;
    if (GUARD) c = c < C ? c : C - 1;
    gm[i] = A.gamma[c];
    bt[i] = A.beta[c];
    rr[i] = A.res ? A.res[base + c * T + tcl] : 0.f;
    vc[i] = A.vec ? A.vec[(int64_t)b * A.vec_bstride + c] : 0.f;
    vc2[i] = A.out2 ? A.vec2[(int64_t)b * A.vec2_bstride + c] : 0.f;
  }
  // ---- wave-level reduction over the channel groups (lanes tx + 8*cg), then across the 4 waves
  const int wv = threadIdx.x >> 6;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i)
    if (!GUARD || ty + i * LN_G < C) s += v[i];
  s += __shfl_xor(s, 8);
  s += __shfl_xor(s, 16);
  s += __shfl_xor(s, 32);
  if ((threadIdx.x & 63) < LN_TT) red[0][wv][tx] = s;
  __syncthreads();
  const float mean = ((red[0][0][tx] + red[0][1][tx]) + (red[0][2][tx] + red[0][3][tx])) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i)
    if (!GUARD || ty + i * LN_G < C) {
      const float d = v[i] - mean;
      q += d * d;
    }
  q += __shfl_xor(q, 8);
  q += __shfl_xor(q, 16);
  q += __shfl_xor(q, 32);
  if ((threadIdx.x & 63) < LN_TT) red[1][wv][tx] = q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf(((red[1][0][tx] + red[1][1][tx]) + (red[1][2][tx] + red[1][3][tx])) / (float)C + A.eps);
  if (!tok) return;
  const bool gelu = A.post_gelu != 0;
  float* const outp = A.out;
  float* const out2p = A.out2;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = ty + i * LN_G;
    if (!GUARD || c < C) {
      float y = (v[i] - mean) * rstd * gm[i] + bt[i];
      if (gelu) y = 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f));
      y += rr[i];
      y += vc[i];
      outp[base + c * T + t] = y * mk;
      if (out2p) out2p[base + c * T + t] = (y + vc2[i]) * mk;
    }
  }
}

template <int CPT, bool GUARD>
static void launch_ln_cfg(hipStream_t stream, const LnArgs& a, dim3 grid) {
  if (a.mode != 0) { hipLaunchKernelGGL((layernorm_kernel<CPT, 1, 1, GUARD>), grid, dim3(256), 0, stream, a); return; }
  if (a.ml) {                                      // nslab = heads x key ranges
    if (a.nslab == 8 && a.ml_ks == 4) hipLaunchKernelGGL((layernorm_kernel<CPT, 8, 0, GUARD, 4>), grid, dim3(256), 0, stream, a);
    else if (a.nslab == 8) hipLaunchKernelGGL((layernorm_kernel<CPT, 8, 0, GUARD, 2>), grid, dim3(256), 0, stream, a);
    else if (a.ml_ks == 4) hipLaunchKernelGGL((layernorm_kernel<CPT, 4, 0, GUARD, 4>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((layernorm_kernel<CPT, 4, 0, GUARD, 2>), grid, dim3(256), 0, stream, a);
    return;
  }
  switch (a.nslab) {
    case 2: hipLaunchKernelGGL((layernorm_kernel<CPT, 2, 0, GUARD>), grid, dim3(256), 0, stream, a); break;
    case 3: hipLaunchKernelGGL((layernorm_kernel<CPT, 3, 0, GUARD>), grid, dim3(256), 0, stream, a); break;   // 3 / 5 / 6 / 7: one slab per head of the
    case 4: hipLaunchKernelGGL((layernorm_kernel<CPT, 4, 0, GUARD>), grid, dim3(256), 0, stream, a); break;   // fused attention + conv_o with n_heads off the powers of two
    case 5: hipLaunchKernelGGL((layernorm_kernel<CPT, 5, 0, GUARD>), grid, dim3(256), 0, stream, a); break;
    case 6: hipLaunchKernelGGL((layernorm_kernel<CPT, 6, 0, GUARD>), grid, dim3(256), 0, stream, a); break;
    case 7: hipLaunchKernelGGL((layernorm_kernel<CPT, 7, 0, GUARD>), grid, dim3(256), 0, stream, a); break;
    case 8: hipLaunchKernelGGL((layernorm_kernel<CPT, 8, 0, GUARD>), grid, dim3(256), 0, stream, a); break;
    default: hipLaunchKernelGGL((layernorm_kernel<CPT, 1, 0, GUARD>), grid, dim3(256), 0, stream, a); break;
  }
}

int launch_layernorm(hipStream_t stream, const LnArgs& a) {
  if (a.C > LN_G * 8 || a.C < 1 || a.T < 1 || a.B < 1) return -1;
  if ((int64_t)a.C * a.T >= (1ll << 31)) return -1;             // 32-bit in-batch offsets
  const int ns = a.nslab < 1 ? 1 : a.nslab;
  if (a.mode == 0 && ns > 8) return -1;
  if (a.ml && (a.mode != 0 || (ns != 4 && ns != 8) || (a.ml_ks != 2 && a.ml_ks != 4) || a.ml_H * a.ml_ks != ns)) return -1;
  dim3 grid((a.T + LN_TT - 1) / LN_TT, a.B);
  LnArgs ax = a;
  if (a.xcd_b) {
    ax.xcd_per = (int)grid.x;
    grid = dim3(xcd_grid(a.B, ax.xcd_per), 1, 1);
    ax.pf = Prefetch{nullptr, 0};
  } else if (a.pf.ptr && a.pf.bytes && a.B == 1) {
    grid = dim3(((grid.x + 7) & ~7u) + PF_BLOCKS, 1);
  } else {
    ax.pf = Prefetch{nullptr, 0};
  }
  if (a.C == 6 * LN_G) launch_ln_cfg<6, false>(stream, ax, grid);         // hidden_channels 192
  else if (a.C == 8 * LN_G) launch_ln_cfg<8, false>(stream, ax, grid);    // DurationPredictor filter 256
  else launch_ln_cfg<8, true>(stream, ax, grid);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace bv2

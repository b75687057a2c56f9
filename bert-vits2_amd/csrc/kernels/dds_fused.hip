// dds_fused.hip — one whole DDSConv layer per launch (reference modules.py:121-129), the unit the StochasticDurationPredictor
// is made of (models.py:197-204, 245-256: 1 trunk + 3 ConvFlows x 3 layers).  Layer by layer the reference does
//     y = convs_sep[i](x * x_mask)     depthwise k=3, dilation 3^i
//     y = gelu(norms_1[i](y));  y = convs_1x1[i](y);  y = gelu(norms_2[i](y));  x = x + y          (return x * x_mask at the end)
// which was three launches here in round 1 (depthwise+LN, split-K 1x1 conv, slab-sum+LN) for a 192 x T problem.  At batch 1 the
// whole predictor is latency: 49 dependent launches.  Now a workgroup owns 16 time steps x all C channels and runs the layer end to
// end: the neighbours' columns the dilated depthwise conv needs are read from HBM/L2 (the previous layer is a previous launch),
// the two channel LayerNorms are wavefront reductions (shuffles over the 4 lane groups of a wave + one LDS exchange between
// the C/16 waves), the 1x1 conv is 4*C/16 v_mfma_f32_16x16x4_f32 per wave on weights prefetched into registers before the first
// activation arrives, and the layer that closes a DDSConv also applies the following 1x1 projection (sdp.proj, or ConvFlow.proj
// + the inverse rational-quadratic spline of transforms.py) to its own output tile.  49 launches -> 15.
#include <hip/hip_runtime.h>
#include <cmath>
#include "../bv2_kernels.h"
#include "spline.h"

namespace bv2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int DDS_NT = 16;                          // time steps per workgroup

__device__ __forceinline__ float gelu_erf(float y) { return 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f)); }

// 1x1 GEMM of one 16-row tile against the [C][16] activation tile in LDS.  Weights: conv_w_index order with k = 1, i.e. float4
// index ((mt32*G + g)*2 + lh)*32 + (row & 31) holds channels 8g + 2q + lh (q = 0..3) of output row `row`.  Lane (m = l & 15,
// kk = l >> 4) supplies A[m][kk]; K step (unit u, q) covers channels 16u + 8*(kk >> 1) + 2q + (kk & 1), so one float4 per lane per
// unit feeds four MFMAs and the B operand of step q is row (16u + 8*(kk >> 1) + (kk & 1)) + 2q of the tile.
template <int NU>
__device__ __forceinline__ void load_tile_weights(const float* w, int row_tile16, int lane, f32x4 (&wr)[NU]) {
  const int m = lane & 15, kk = lane >> 4;
  constexpr int G = 2 * NU;
  const f32x4* wp = reinterpret_cast<const f32x4*>(w);
#pragma unroll
  for (int u = 0; u < NU; ++u)
    wr[u] = wp[((((row_tile16 >> 1) * G + 2 * u + (kk >> 1)) * 2 + (kk & 1)) * 32) + 16 * (row_tile16 & 1) + m];
}

template <int NU>
__device__ __forceinline__ f32x4 tile_gemm(const f32x4 (&wr)[NU], const float* ys, int lane) {
  const int n = lane & 15, kk = lane >> 4;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};       // two chains: the 16x16x4 f32 MFMA has 40-cycle latency
  const float* yb = ys + (8 * (kk >> 1) + (kk & 1)) * DDS_NT + n;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const float* yu = yb + u * 16 * DDS_NT;
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[u].x, yu[0 * 2 * DDS_NT], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[u].y, yu[1 * 2 * DDS_NT], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[u].z, yu[2 * 2 * DDS_NT], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[u].w, yu[3 * 2 * DDS_NT], acc1, 0, 0, 0);
  }
  return acc0 + acc1;
}

template <int NU>   // C = 16*NU channels; NU waves per workgroup
__global__ void __launch_bounds__(64 * NU) dds_layer_kernel(const DdsArgs A) {
  constexpr int C = 16 * NU, NW = NU, CG = 4 * NU;   // CG channel groups of the elementwise phase: thread (tl, cg) owns channels cg + CG*i
  __shared__ __attribute__((aligned(16))) float ys[C * DDS_NT];     // GEMM B operand: gelu(LN1(.)) tile, later the layer's output tile
  __shared__ float xres[C * DDS_NT];                                // residual stream tile (centre taps)
  __shared__ float red[4][NW][DDS_NT];
  __shared__ float prm[32 * DDS_NT];                                // ConvFlow.proj output (spline parameters)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y, t0 = blockIdx.x * DDS_NT;
  const int T = A.T;

  // ---- weights of this wave's 16-row tile: in flight before any activation is touched
  // (C = 256: 64 weight registers next to the elementwise phase's 60 do not fit the 128 a 16-wave workgroup leaves each lane — 10 spilled; there
  // the weights are requested after phase 1 instead, at the price of an exposed L2 round trip.  No scratch segment on any accepted model's path.)
  f32x4 wr[NU];
  if constexpr (NU <= 12) load_tile_weights<NU>(A.w, wv, lane, wr);

  // ---- phase 1: depthwise conv + LN1 + GELU on thread (tl = time step, cg = channel group)
  const int tl = tid & 15, cg = tid >> 4;
  const float* maskb = A.mask + (int64_t)b * T;
  // per-channel parameters of BOTH LayerNorms and the conv bias: in flight from the start (behind the barriers, where the
  // compiler would otherwise issue them, each group is an exposed L2 round trip)
  float pg1[4], pb1[4], pg2[4], pb2[4], pbias[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pg1[i] = A.g1[cg + CG * i]; pb1[i] = A.b1[cg + CG * i];
    const int row = 16 * wv + 4 * (lane >> 4) + i;               // this lane's rows of the MFMA output (phase 3)
    pg2[i] = A.g2[row]; pb2[i] = A.b2[row]; pbias[i] = A.bias[row];
  }
  float v[4], xc[4];
  {
    const int t = t0 + tl;
    const bool tok = t < T;
    const int tcl = tok ? t : T - 1;
    float mk3[3];
    int tt3[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int tt = t + (j - 1) * A.dil;
      const bool ok = tok && tt >= 0 && tt < T;
      tt3[j] = ok ? tt : tcl;
      const float mv = maskb[tt3[j]];                 // unconditional (tt3 is a valid index), selected afterwards
      mk3[j] = ok ? mv : 0.f;
    }
    float z3[3];
    {
      const float* zr = A.pre_w ? A.z + ((int64_t)b * 2 + A.z_src) * T : maskb;   // no ConvFlow.pre: valid dummy, values unused
#pragma unroll
      for (int j = 0; j < 3; ++j) z3[j] = zr[tt3[j]];
    }
    // all loads of the four channel groups first, arithmetic after: with the `pre_w ?` branch inside the unrolled channel loop
    // every iteration was a basic block of its own ending in s_waitcnt vmcnt(0) — five serial memory round trips per layer
    // launch (ISA of round 2)
    float x3[4][3], dw[4][3], db[4], pw[4], pb[4];
    const bool has_pre = A.pre_w != nullptr;
    const float* const src = (has_pre ? A.g : A.x) + (int64_t)b * C * T;
    const float* const pwp = has_pre ? A.pre_w : A.dwb;            // no ConvFlow.pre: any valid vector, values unused
    const float* const pbp = has_pre ? A.pre_b : A.dwb;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = cg + CG * i;
      const float* xp = src + (int64_t)c * T;
#pragma unroll
      for (int j = 0; j < 3; ++j) { x3[i][j] = xp[tt3[j]]; dw[i][j] = A.dww[c * 3 + j]; }
      db[i] = A.dwb[c]; pw[i] = pwp[c]; pb[i] = pbp[c];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (has_pre) {
#pragma unroll
        for (int j = 0; j < 3; ++j) x3[i][j] = pw[i] * z3[j] + pb[i] + x3[i][j];
      }
      xc[i] = x3[i][1];
      float acc = db[i];
#pragma unroll
      for (int j = 0; j < 3; ++j) acc += dw[i][j] * (x3[i][j] * mk3[j]);
      v[i] = acc;
    }
  }
  {
    float s = (v[0] + v[1]) + (v[2] + v[3]);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (lane < DDS_NT) red[0][wv][lane] = s;
    __syncthreads();
    float m = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) m += red[0][w][tl];
    const float mean = m / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float d = v[i] - mean; q += d * d; }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    if (lane < DDS_NT) red[1][wv][lane] = q;
    __syncthreads();
    float qs = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) qs += red[1][w][tl];
    const float rstd = 1.0f / sqrtf(qs / (float)C + A.eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = cg + CG * i;
      const float y = (v[i] - mean) * rstd * pg1[i] + pb1[i];
      ys[c * DDS_NT + tl] = gelu_erf(y);
      xres[c * DDS_NT + tl] = xc[i];
    }
  }
  __syncthreads();

  // ---- phase 2: 1x1 conv, wave wv -> output rows [16 wv, 16 wv + 16)
  if constexpr (NU > 12) load_tile_weights<NU>(A.w, wv, lane, wr);
  f32x4 acc = tile_gemm<NU>(wr, ys, lane);

  // weights of the optional post projection: loaded now (wr is dead), they land under the LN2 arithmetic
  const bool post = A.post_w != nullptr;
  const bool post_wave = post && 16 * wv < A.post_cout_pad;       // wave-uniform
  f32x4 pw[NU];
  if (post_wave) load_tile_weights<NU>(A.post_w, wv, lane, pw);

  // ---- phase 3: bias + LN2 + GELU + residual on the MFMA output layout: lane (n = l & 15, rg = l >> 4), reg r -> row 16 wv + 4 rg + r
  const int n = lane & 15, rg = lane >> 4;
  const int t = t0 + n;
  const bool tok = t < T;
  const float mkc = tok ? maskb[t] : 0.f;
  float o[4];
  {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = acc[r] + pbias[r];
    float s = (o[0] + o[1]) + (o[2] + o[3]);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (lane < DDS_NT) red[2][wv][lane] = s;
    __syncthreads();
    float m = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) m += red[2][w][n];
    const float mean = m / (float)C;
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float d = o[r] - mean; q += d * d; }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    if (lane < DDS_NT) red[3][wv][lane] = q;
    __syncthreads();
    float qs = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) qs += red[3][w][n];
    const float rstd = 1.0f / sqrtf(qs / (float)C + A.eps);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * wv + 4 * rg + r;
      const float y = gelu_erf((o[r] - mean) * rstd * pg2[r] + pb2[r]);
      float val = xres[row * DDS_NT + n] + y;
      if (A.last_mask) val *= mkc;
      o[r] = val;
      if (A.out && tok) A.out[((int64_t)b * C + row) * T + t] = val;
    }
  }
  if (!post) return;                                               // kernel-uniform

  // ---- phase 4: the 1x1 projection that follows the DDSConv, on this tile (every wave passed phase 2 two barriers ago)
#pragma unroll
  for (int r = 0; r < 4; ++r) ys[(16 * wv + 4 * rg + r) * DDS_NT + n] = o[r];
  __syncthreads();
  if (post_wave) {
    const f32x4 pa = tile_gemm<NU>(pw, ys, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * wv + 4 * rg + r;
      const float val = (pa[r] + A.post_b[row]) * mkc;
      if (A.post_out) {
        if (tok && row < A.post_cout) A.post_out[((int64_t)b * A.post_cout + row) * T + t] = val;
      } else if (row < 32) {
        prm[row * DDS_NT + n] = val;
      }
    }
  }
  if (A.post_out || !A.zio) return;                               // kernel-uniform
  __syncthreads();
  if (tid < DDS_NT) {
    const int ts = t0 + tid;
    if (ts < T) {
      const float cst = A.cst, wscale = A.wscale;                   // host-evaluated constants (launch_dds_layer)
      float uw[SPK], uh[SPK], ud[SPK + 1];
#pragma unroll
      for (int i = 0; i < SPK; ++i) { uw[i] = prm[i * DDS_NT + tid] / A.sqrt_fc; uh[i] = prm[(SPK + i) * DDS_NT + tid] / A.sqrt_fc; }
      ud[0] = cst; ud[SPK] = cst;
#pragma unroll
      for (int i = 1; i < SPK; ++i) ud[i] = prm[(2 * SPK + i - 1) * DDS_NT + tid];
      const float mk = maskb[ts];
      float* zs = A.zio + ((int64_t)b * 2 + A.z_src) * T + ts;
      float* zd = A.zio + ((int64_t)b * 2 + A.z_dst) * T + ts;
      const float outv = rq_spline_inverse_one(*zd, uw, uh, ud, A.tail, wscale);
      *zd = outv * mk;
      *zs = *zs * mk;
    }
  }
}

bool dds_fused_supported(int C) { return C == 128 || C == 192 || C == 256; }

int launch_dds_layer(hipStream_t stream, const DdsArgs& a) {
  if (!dds_fused_supported(a.C) || a.B < 1 || a.T < 1 || a.dil < 1 || !a.mask || !a.w || !a.bias) return -2;
  if (!a.x && !a.pre_w) return -1;
  if (a.out && a.out == a.x) return -1;                           // tiles read their neighbours' input columns
  if (a.post_w && (a.post_cout_pad % 16 || a.post_cout_pad > a.C || (!a.post_out && (!a.zio || a.post_cout_pad < 32)))) return -1;
  if ((int64_t)a.C * a.T >= (1ll << 31)) return -1;
  const dim3 grid((a.T + DDS_NT - 1) / DDS_NT, a.B);
  DdsArgs k = a;
  // constants the reference evaluates in Python doubles before they meet fp32 tensors (transforms.py:71, :128)
  k.cst = (float)std::log(std::exp(1.0 - 1e-3) - 1.0);
  k.wscale = (float)(1.0 - 1e-3 * SPK);
  switch (a.C) {
    case 128: hipLaunchKernelGGL((dds_layer_kernel<8>), grid, dim3(512), 0, stream, k); break;
    case 192: hipLaunchKernelGGL((dds_layer_kernel<12>), grid, dim3(768), 0, stream, k); break;
    default:  hipLaunchKernelGGL((dds_layer_kernel<16>), grid, dim3(1024), 0, stream, k); break;
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace bv2

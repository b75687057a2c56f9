// attention.hip — MultiHeadAttention.attention with windowed relative positions (reference attentions.py:273-322 and the
// skew helpers :324-395), flash-style on the fp32 matrix core, never materialising the [B,H,T,T] score tensor.
//
// What the reference does with zero-padded [2T-1]-wide relative matmuls + pad/reshape skew tricks is, mathematically,
//    logit[i][j] = q_i·k_j/sqrt(d) + (|j-i| <= W ? q_i·Ek[j-i+W]/sqrt(d) : 0);   masked pairs are SET to -1e4
//    out[i]      = sum_j p[i][j] v_j  +  sum_{|r|<=W, 0<=i+r<T} p[i][i+r] Ev[r+W]
// (verified bit-for-bit against the reference in oracle/).  So only 2W+1 = 9 extra dot products
// per query are needed instead of the reference's padded matmuls (which triple its attention FLOPs).
//
// Mapping: one workgroup = 32 queries of one (batch, head); its 4 waves split the KEY tiles (32 keys each) round-robin
// (flash-decoding style) and merge their (max, sum, O) partials through LDS at the end.
//   S^T tile  [32 keys x 32 queries] = K^T·Q : A[m=key][kk=c] = k[c][j] and B[kk=c][n=query] = q[c][i] are both natural
//             row reads of the [C][T] layout (time contiguous) — no transposes anywhere; Q stays in registers.
//   D layout: lane holds ONE query column (l&31) and 16 key rows -> row max / row sum are in-lane + one shfl_xor(32).
//   O^T tile  [D x 32 queries] += V·P^T : A[m=c][kk=key] from an LDS V tile (stride 33, conflict free),
//             B[kk=key][n=query] = P^T written to LDS straight from the D layout.
#include <hip/hip_runtime.h>
#include <math.h>
#include "../bv2_kernels.h"

namespace bv2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int AQ = 32;          // queries per workgroup
constexpr int AK = 32;          // keys per tile
constexpr int AMAXW = 8;        // max window

template <int DT>               // D = 32*DT head channels
__global__ void __launch_bounds__(256) attention_kernel(const AttnArgs A) {
  constexpr int D = 32 * DT;
  constexpr int VS = AK + 1;    // V tile row stride
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * AQ;
  const int T = A.T, W = A.W, NR = 2 * W + 1;
  const int HD = A.H * D;

  float* Vs = smem + wid * (D * VS + AK * AQ);      // per wave: V tile [D][VS]
  float* Ps = Vs + D * VS;                          // per wave: P^T tile [AK][AQ]
  float* Os = smem + 4 * (D * VS + AK * AQ);        // [D][AQ] merged output
  float* Qe = Os + D * AQ;                          // [NR][AQ] q_i·Ek[r]/sqrt(d)
  float* Sb = Qe + (2 * AMAXW + 1) * AQ;            // [NR][AQ] raw band logits
  float* Mw = Sb + (2 * AMAXW + 1) * AQ;            // [4][AQ]
  float* Lw = Mw + 4 * AQ;                          // [4][AQ]

  const float* qp = A.qkv + (int64_t)b * 3 * HD * T + (int64_t)(h * D) * T;
  const float* kp = qp + (int64_t)HD * T;
  const float* vp = kp + (int64_t)HD * T;
  const float* mp = A.mask + (int64_t)b * T;
  const float sq = sqrtf((float)D);

  const int iq = i0 + l31;                          // this lane's query
  const bool iok = iq < T;
  const float mi = iok ? mp[iq] : 0.f;

  // Q operand in registers (scaled: the reference divides the query by sqrt(d) first, attentions.py:280)
  float qreg[D / 2];
#pragma unroll
  for (int s = 0; s < D / 2; ++s) qreg[s] = iok ? qp[(int64_t)(2 * s + lh) * T + iq] / sq : 0.f;

  // relative-key logits Qe[r][i] = sum_c qs[c][i] * Ek[r][c]
  for (int idx = tid; idx < NR * AQ; idx += 256) {
    const int r = idx / AQ, i = idx - r * AQ;
    float acc = 0.f;
    if (i0 + i < T)
      for (int c = 0; c < D; ++c) acc += (qp[(int64_t)c * T + i0 + i] / sq) * A.erk[r * D + c];
    Qe[r * AQ + i] = acc;
  }
  __syncthreads();

  f32x16 O[DT];
#pragma unroll
  for (int m = 0; m < DT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[m][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int ntiles = (T + AK - 1) / AK;
  for (int kt = wid; kt < ntiles; kt += 4) {
    const int j0 = kt * AK;
    const int jk = j0 + l31;                        // key this lane LOADS (A operand row / V column)
    const bool jok = jk < T;
    // ---- S^T = K^T Q
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
    for (int s = 0; s < D / 2; ++s) {
      const float a = jok ? kp[(int64_t)(2 * s + lh) * T + jk] : 0.f;
      S = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qreg[s], S, 0, 0, 0);
    }
    // ---- relative logits, mask, band capture, online softmax (lane = query column, regs = key rows)
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float sv = S[r];
      const int rel = j - iq;
      const bool inband = (rel >= -W) && (rel <= W);
      if (inband) sv += Qe[(rel + W) * AQ + l31];
      if (j < T) {
        if (!(mi != 0.f && mp[j] != 0.f)) sv = -1e4f;        // masked_fill(mask == 0, -1e4), attentions.py:297
        if (inband) Sb[(rel + W) * AQ + l31] = sv;
      } else {
        sv = -INFINITY;                                       // key does not exist
      }
      S[r] = sv;
      tmax = fmaxf(tmax, sv);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = expf(m_run - m_new);                  // first tile: exp(-inf) = 0
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = expf(S[r] - m_new);
      psum += p;
      Ps[((r & 3) + 8 * (r >> 2) + 4 * lh) * AQ + l31] = p;
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int m = 0; m < DT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[m][r] *= alpha;
    // ---- V tile -> LDS (rows = channels, coalesced 128-byte segments)
#pragma unroll 4
    for (int c = lh; c < D; c += 2) Vs[c * VS + l31] = jok ? vp[(int64_t)c * T + jk] : 0.f;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): this wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    // ---- O^T += V P^T
#pragma unroll
    for (int s = 0; s < AK / 2; ++s) {
      const float pb = Ps[(2 * s + lh) * AQ + l31];
#pragma unroll
      for (int m = 0; m < DT; ++m) {
        const float a = Vs[(m * 32 + l31) * VS + 2 * s + lh];
        O[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb, O[m], 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }

  // ---- merge the 4 waves' partials
  if (lh == 0) { Mw[wid * AQ + l31] = m_run; Lw[wid * AQ + l31] = l_run; }
  __syncthreads();
  float m_tot = fmaxf(fmaxf(Mw[l31], Mw[AQ + l31]), fmaxf(Mw[2 * AQ + l31], Mw[3 * AQ + l31]));
  float l_tot = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float mw = Mw[w * AQ + l31];
    l_tot += (mw == -INFINITY) ? 0.f : Lw[w * AQ + l31] * expf(mw - m_tot);
  }
  const float fac = (m_run == -INFINITY) ? 0.f : expf(m_run - m_tot);
  for (int w = 0; w < 4; ++w) {
    if (wid == w) {
#pragma unroll
      for (int m = 0; m < DT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const float val = O[m][r] * fac;
          if (w == 0) Os[c * AQ + l31] = val; else Os[c * AQ + l31] += val;
        }
    }
    __syncthreads();
  }
  // stash the final (m, 1/l) per query for the output pass
  if (wid == 0 && lh == 0) { Mw[l31] = m_tot; Lw[l31] = 1.0f / l_tot; }
  __syncthreads();

  // ---- normalise, add relative-value term, store (coalesced over queries)
  float* op = A.out + (int64_t)b * HD * T + (int64_t)(h * D) * T;
  for (int idx = tid; idx < D * AQ; idx += 256) {
    const int c = idx / AQ, i = idx - c * AQ;
    const int ig = i0 + i;
    if (ig >= T) continue;
    const float mt = Mw[i], il = Lw[i];
    float o = Os[c * AQ + i] * il;
    for (int r = 0; r < NR; ++r) {
      const int j = ig + r - W;
      if (j >= 0 && j < T) o += (expf(Sb[r * AQ + i] - mt) * il) * A.erv[r * D + c];
    }
    op[(int64_t)c * T + ig] = o;
  }
}

static size_t attn_lds_bytes(int D) {
  return sizeof(float) * (size_t)(4 * (D * (AK + 1) + AK * AQ) + D * AQ + 2 * (2 * AMAXW + 1) * AQ + 8 * AQ);
}

int launch_attention(hipStream_t stream, const AttnArgs& a) {
  if (a.W > AMAXW || a.W < 0 || a.T < 1 || a.B < 1 || a.H < 1) return -1;
  dim3 grid((a.T + AQ - 1) / AQ, a.H, a.B);
  const size_t lds = attn_lds_bytes(a.D);
#define BV2_ATTN_CASE(DT_)                                                                                            \
  case 32 * DT_: {                                                                                                    \
    auto kern = attention_kernel<DT_>;                                                                                \
    if (lds > 64 * 1024)                                                                                              \
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);                                                        \
  } break;
  switch (a.D) {
    BV2_ATTN_CASE(1)
    BV2_ATTN_CASE(2)
    BV2_ATTN_CASE(3)
    BV2_ATTN_CASE(4)
    default: return -2;   // head dim must be a multiple of 32 up to 128
  }
#undef BV2_ATTN_CASE
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

double attention_flops(const AttnArgs& a) {
  // QK^T + PV: 4*T*T*D per head (+ 2*(2W+1) band dot products per query)
  return (double)a.B * a.H * (4.0 * a.T * (double)a.T * a.D + 4.0 * (2 * a.W + 1) * a.T * a.D);
}

}  // namespace bv2

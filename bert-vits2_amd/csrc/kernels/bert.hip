// bert.hip — the two kernels of the BERT feature extractor (bv2_bert.cpp, include/bv2_bert.h) that are not convolutions or
// attention: the embedding sum + LayerNorm (transformers BertEmbeddings.forward: word + token-type + position embeddings,
// LayerNorm(eps), dropout = identity at inference) and the residual LayerNorm behind every attention / feed-forward block
// (BertSelfOutput / BertOutput: LayerNorm(dense(h) + input)).  Reference call site: text/chinese_bert.py:34-37.
//
// Activations are fp32 [B][C][S] (channels-first, S contiguous) like everything else in libbv2 — the layout the split-K GEMM
// kernel reads as its B operand and the TextEncoder front consumes at word level, so the final hidden state needs no transpose.
#include <hip/hip_runtime.h>
#include "../bv2_kernels.h"

namespace bv2 {

// ---------------------------------------------------------------------------------------------------------------
// One workgroup (256 threads) per token: x[c] = word[id][c] + pos[s][c] + type[tt][c]; two-pass LayerNorm over C; written to column s.
// C <= 2048 (8 channels per thread).
__global__ void __launch_bounds__(256) bert_embed_ln_kernel(const BertEmbedArgs A) {
  __shared__ float red[2][4];
  const int s = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int C = A.C, S = A.S;
  if (A.pf.ptr) {                                   // B == 1: tokens padded to a multiple of 8, then PF_BLOCKS spare workgroups (Prefetch)
    const unsigned n8 = ((unsigned)S + 7u) & ~7u;
    if (blockIdx.x >= n8) { prefetch_tail(A.pf, blockIdx.x, blockIdx.x - n8, threadIdx.x, 256); return; }
    if (s >= S) return;
  }
  int64_t id = A.input_ids[(int64_t)b * S + s];
  id = id < 0 ? 0 : (id >= A.vocab ? A.vocab - 1 : id);             // clamped like every gather in libbv2 (a bad id must not fault)
  int64_t tt = A.token_type_ids ? A.token_type_ids[(int64_t)b * S + s] : 0;
  tt = tt < 0 ? 0 : (tt >= A.type_vocab ? A.type_vocab - 1 : tt);
  const int ps = s < A.max_pos ? s : A.max_pos - 1;
  const float* wrow = A.word + id * C;
  const float* prow = A.pos ? A.pos + (int64_t)ps * C : nullptr;
  const float* trow = A.type ? A.type + tt * C : nullptr;
  const float mk = (A.lengths && (int64_t)s >= A.lengths[b]) ? 0.f : 1.f;
  float v[8], gm[8], bt[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int c = tid + 256 * i;
    c = c < C ? c : C - 1;
    // BertEmbeddings: inputs_embeds + token_type_embeddings, then + position_embeddings; DebertaV2Embeddings here: inputs_embeds only
    v[i] = (wrow[c] + (trow ? trow[c] : 0.f)) + (prow ? prow[c] : 0.f);
    gm[i] = A.gamma[c];
    bt[i] = A.beta[c];
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (tid + 256 * i < C) sum += v[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if ((tid & 63) == 0) red[0][tid >> 6] = sum;
  __syncthreads();
  const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (tid + 256 * i < C) { const float d = v[i] - mean; q += d * d; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  if ((tid & 63) == 0) red[1][tid >> 6] = q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)C + A.eps);
  float* op = A.out + (int64_t)b * C * S + s;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = tid + 256 * i;
    if (c < C) op[(int64_t)c * S] = ((v[i] - mean) * rstd * gm[i] + bt[i]) * mk;
  }
}

int launch_bert_embed_ln(hipStream_t stream, const BertEmbedArgs& a) {
  if (a.C < 1 || a.C > 2048 || a.S < 1 || a.B < 1) return -1;
  if (a.pf.ptr && a.pf.bytes && a.B == 1) {
    hipLaunchKernelGGL(bert_embed_ln_kernel, dim3(((a.S + 7) & ~7) + PF_BLOCKS, 1), dim3(256), 0, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  BertEmbedArgs a2 = a;
  a2.pf = Prefetch{nullptr, 0};
  hipLaunchKernelGGL(bert_embed_ln_kernel, dim3(a.S, a.B), dim3(256), 0, stream, a2);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------------------------------------------
// out = LayerNorm_C(sum of `nslab` partial slabs of a split-K GEMM) — slab 0 already carries the GEMM's bias and the residual
// (conv_mfma.hip's split-K epilogue), so this is BertSelfOutput / BertOutput's LayerNorm(dense(h) + input).
// A workgroup owns 8 consecutive tokens x all C channels: 1024 threads = (tx = token, ty = one of 128 channel groups), CPT = C/128
// values per thread in registers, every load in flight at once.  Wave = 8 tokens x 8 channel groups, so each pass of the exact
// two-pass variance is three __shfl_xor butterflies + one exchange between the 16 waves through LDS (layernorm.hip's scheme at
// 4x the width).
// TT = 8 tokens x 128 channel groups per workgroup, or — for a single short sentence, where 8-token workgroups are only ceil(S/8) = 7
// CUs pulling 0.9 MB through 32-byte pieces — TT = 2 tokens x 512 channel groups: 27 workgroups, two channels (x nslab loads) per thread.
template <int CPT, int NSLAB, int TT>   // NSLAB > 0: compile-time slab count (every load of a thread in flight at once); 0: runtime loop
__global__ void __launch_bounds__(1024) bert_ln_kernel(const BertLnArgs A) {
  constexpr int G = 1024 / TT;              // channel groups
  __shared__ float red[2][16][TT];
  const int tx = threadIdx.x & (TT - 1), ty = threadIdx.x / TT;
  const int b = blockIdx.y;
  const int C = A.C, T = A.T;
  if (A.pf.ptr) {                                   // B == 1: token tiles padded to a multiple of 8, then PF_BLOCKS spare workgroups (Prefetch)
    const unsigned nt8 = ((unsigned)(T + TT - 1) / TT + 7u) & ~7u;
    if (blockIdx.x >= nt8) { prefetch_tail(A.pf, blockIdx.x, blockIdx.x - nt8, threadIdx.x, 1024); return; }
    if ((int)blockIdx.x * TT >= T) return;
  }
  const int t = blockIdx.x * TT + tx;
  const bool tok = t < T;
  const int tcl = tok ? t : T - 1;
  const int64_t base = (int64_t)b * C * T;
  const float* ap = A.a + base;
  const int nslab = A.nslab;
  float v[CPT], gm[CPT], bt[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = ty + i * G;
    const int off = c * T + tcl;
    float x = ap[off];
    if constexpr (NSLAB > 0) {
      // a runtime trip count made the compiler issue the slabs one dependent round trip after the other: 13 us instead of 6
      // (rocprofv3, profiles/r02_c_kernel_stats_bert.txt) for 1.7 MB of input
      float part[NSLAB > 1 ? NSLAB - 1 : 1];
#pragma unroll
      for (int sl = 1; sl < NSLAB; ++sl) part[sl - 1] = ap[(int64_t)sl * A.slab_stride + off];
#pragma unroll
      for (int sl = 1; sl < NSLAB; ++sl) x += part[sl - 1];
    } else {
      for (int sl = 1; sl < nslab; ++sl) x += ap[(int64_t)sl * A.slab_stride + off];
    }
    v[i] = x;
    gm[i] = A.gamma[c];
    bt[i] = A.beta[c];
  }
  const int wv = threadIdx.x >> 6;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) s += v[i];
#pragma unroll
  for (int o = TT; o < 64; o <<= 1) s += __shfl_xor(s, o);       // lanes tx, tx + TT, ...: the wave's 64 / TT channel groups of token tx
  if ((threadIdx.x & 63) < TT) red[0][wv][tx] = s;
  __syncthreads();
  float m = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) m += red[0][w][tx];
  const float mean = m / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) { const float d = v[i] - mean; q += d * d; }
#pragma unroll
  for (int o = TT; o < 64; o <<= 1) q += __shfl_xor(q, o);
  if ((threadIdx.x & 63) < TT) red[1][wv][tx] = q;
  __syncthreads();
  float qq = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) qq += red[1][w][tx];
  const float rstd = 1.0f / sqrtf(qq / (float)C + A.eps);
  if (!tok) return;
  float* const outp = A.out + base;
  const float mk = A.mask ? A.mask[(int64_t)b * T + t] : 1.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = ty + i * G;
    outp[c * T + t] = ((v[i] - mean) * rstd * gm[i] + bt[i]) * mk;
  }
}

int launch_bert_ln(hipStream_t stream, const BertLnArgs& a0) {
  constexpr int BLN_G = 128;                        // channel groups of the 8-token form
  if (a0.C < BLN_G || a0.C % BLN_G || a0.C > 8 * BLN_G || a0.T < 1 || a0.B < 1 || a0.nslab < 1) return -1;
  if ((int64_t)a0.C * a0.T >= (1ll << 31)) return -1;
  // few tokens (one short sentence): 2-token workgroups, 4x the CUs (measured: profiles/r03_*bert*)
  const bool narrow = (int64_t)a0.B * a0.T <= 128 && a0.C % 512 == 0 && a0.C <= 1024 && (a0.nslab == 1 || a0.nslab == 2 || a0.nslab == 4);
  BertLnArgs a = a0;
  const bool pf = a.pf.ptr && a.pf.bytes && a.B == 1;
  if (!pf) a.pf = Prefetch{nullptr, 0};
  if (narrow) {
    dim3 grid((a.T + 1) / 2, a.B);
    if (pf) grid = dim3(((grid.x + 7) & ~7u) + PF_BLOCKS, 1);
#define BLN2(CPT_, NS_) hipLaunchKernelGGL((bert_ln_kernel<CPT_, NS_, 2>), grid, dim3(1024), 0, stream, a)
    if (a.C == 512) { if (a.nslab == 1) BLN2(1, 1); else if (a.nslab == 2) BLN2(1, 2); else BLN2(1, 4); }
    else { if (a.nslab == 1) BLN2(2, 1); else if (a.nslab == 2) BLN2(2, 2); else BLN2(2, 4); }
#undef BLN2
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  dim3 grid((a.T + 7) / 8, a.B);
  if (pf) grid = dim3(((grid.x + 7) & ~7u) + PF_BLOCKS, 1);
  const int cpt = a.C / BLN_G;
#define BLN_LAUNCH(CPT_, NS_) hipLaunchKernelGGL((bert_ln_kernel<CPT_, NS_, 8>), grid, dim3(1024), 0, stream, a)
#define BLN_CPT(CPT_)                                                                                  \
  switch (a.nslab) {                                                                                   \
    case 1: BLN_LAUNCH(CPT_, 1); break;                                                                \
    case 2: BLN_LAUNCH(CPT_, 2); break;                                                                \
    case 4: BLN_LAUNCH(CPT_, 4); break;                                                                \
    case 8: BLN_LAUNCH(CPT_, 8); break;                                                                \
    default: BLN_LAUNCH(CPT_, 0); break;                                                               \
  }
  switch (cpt) {
    case 1: BLN_CPT(1); break;
    case 2: BLN_CPT(2); break;
    case 3: BLN_CPT(3); break;
    case 4: BLN_CPT(4); break;
    case 5: BLN_CPT(5); break;
    case 6: BLN_CPT(6); break;
    case 7: BLN_CPT(7); break;
    default: BLN_CPT(8); break;
  }
#undef BLN_CPT
#undef BLN_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace bv2

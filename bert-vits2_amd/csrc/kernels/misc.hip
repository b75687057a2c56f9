// misc.hip — the small, HBM/latency-bound kernels of the path: conv_post+tanh, speaker GEMVs, embeddings, masks,
// the stochastic-duration-predictor glue (ConvFlow pre, inverse rational-quadratic spline, ElementwiseAffine^-1),
// duration -> length regulation.  All fp32, all coalesced along time.
#include <hip/hip_runtime.h>
#include <math.h>
#include <cmath>
#include "../bv2_kernels.h"
#include "spline.h"

namespace bv2 {

#define BV2_CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : -1)

// ---------------------------------------------------------------------------------------------------------------
// conv_post (C -> 1, k taps, no bias) + tanh, fed by leaky_relu(mean of the three ResBlock branches)
// reference models.py:553-555 (NOTE slope 0.01 = F.leaky_relu default, not LRELU_SLOPE).
// One workgroup = 256 consecutive samples: the activated input tile lrelu(mean of the branches) [C][256 + k - 1] is staged in LDS
// ONCE (each element is read from HBM once instead of k times per branch, and the branch mean / leaky-ReLU are computed once per
// element instead of once per tap), then every thread runs its C*k FMAs out of LDS.
__global__ void __launch_bounds__(256) conv_post_kernel(const ConvPostArgs A) {
  extern __shared__ float ws[];                     // [C*k] weights, then [C][256 + k - 1] activated inputs
  const int C = A.C, k = A.k, pad = (k - 1) / 2, W = 256 + k - 1;
  float* xs = ws + C * k;
  const int b = blockIdx.y, t0 = blockIdx.x * 256;
  int Lv = A.L;
  if (A.lens) {
    const int64_t lv = A.lens[b] * A.len_mul;
    Lv = lv < Lv ? (int)lv : Lv;
  }
  for (int i = threadIdx.x; i < C * k; i += 256) ws[i] = A.w[i];
  const float* x0 = A.x[0] + (int64_t)b * A.x_bstride;
  const float* x1 = A.nsrc > 1 ? A.x[1] + (int64_t)b * A.x_bstride : nullptr;
  const float* x2 = A.nsrc > 2 ? A.x[2] + (int64_t)b * A.x_bstride : nullptr;
  const float in_scale = A.in_scale, slope = A.slope;
  for (int e = threadIdx.x; e < C * W; e += 256) {
    const int c = e / W, j = e - c * W;
    const int tt = t0 - pad + j;
    float v = 0.f;
    if (tt >= 0 && tt < Lv) {
      const int64_t o = (int64_t)c * A.x_rstride + tt;
      v = x0[o];
      if (x1) v += x1[o];
      if (x2) v += x2[o];
      v *= in_scale;
      v = v > 0.f ? v : v * slope;
    }
    xs[e] = v;
  }
  __syncthreads();
  const int t = t0 + threadIdx.x;
  if (t >= A.L) return;
  float acc = 0.f;
  for (int c = 0; c < C; ++c)
    for (int j = 0; j < k; ++j) acc += ws[c * k + j] * xs[c * W + threadIdx.x + j];
  A.out[(int64_t)b * A.out_bstride + t] = tanhf(acc);
}

// C known at compile time (the Generator's last stage: 16 channels): every load of the tile is in flight at once — one column per
// thread and channel, the k - 1 halo columns by the first k - 1 threads — instead of C*(256+k-1)/256 dependent loop iterations with
// a division each (tools/timeline.py / rocprofv3: 54 us for 37.7 MB at batch 1 = 0.7 TB/s, latency-bound).
template <int C>
__global__ void __launch_bounds__(256) conv_post_c_kernel(const ConvPostArgs A) {
  extern __shared__ float ws[];                     // [C*k] weights, then [C][256 + k - 1] activated inputs
  const int k = A.k, pad = (k - 1) / 2, W = 256 + k - 1;
  float* xs = ws + C * k;
  const int tid = threadIdx.x;
  const int b = blockIdx.y, t0 = blockIdx.x * 256;
  int Lv = A.L;
  if (A.lens) {
    const int64_t lv = A.lens[b] * A.len_mul;
    Lv = lv < Lv ? (int)lv : Lv;
  }
  const float* x0 = A.x[0] + (int64_t)b * A.x_bstride;
  const float* x1 = A.nsrc > 1 ? A.x[1] + (int64_t)b * A.x_bstride : x0;
  const float* x2 = A.nsrc > 2 ? A.x[2] + (int64_t)b * A.x_bstride : x0;
  const float f1 = A.nsrc > 1 ? 1.f : 0.f, f2 = A.nsrc > 2 ? 1.f : 0.f;
  const float in_scale = A.in_scale, slope = A.slope;
  const unsigned rs = (unsigned)A.x_rstride;
  // main column j = tid, halo column j = 256 + tid (tid < k - 1)
  const int ta = t0 - pad + tid, tb = ta + 256;
  const bool oka = ta >= 0 && ta < Lv, okb = tid < k - 1 && tb >= 0 && tb < Lv;
  // halo loads are UNCONDITIONAL: a thread without a halo column re-reads halo column k - 2 (one broadcast line per wave
  // instruction, no extra traffic).  Behind `if (tid < k - 1)` the compiler merged the load block with the store block below into
  // load-pair -> s_waitcnt vmcnt(0) -> store per channel: 16 serial round trips in the one wave every other wave of the workgroup
  // then waits for at the barrier (ISA of round 2)
  const int tbc = tid < k - 1 ? tb : t0 - pad + 256 + (k - 2);
  const unsigned ca = (unsigned)(ta < 0 ? 0 : (ta >= Lv ? Lv - 1 : ta)), cb = (unsigned)(tbc < 0 ? 0 : (tbc >= Lv ? Lv - 1 : tbc));
  float va[C][3], vb[C][3];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    va[c][0] = x0[c * rs + ca]; va[c][1] = x1[c * rs + ca]; va[c][2] = x2[c * rs + ca];
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    vb[c][0] = x0[c * rs + cb]; vb[c][1] = x1[c * rs + cb]; vb[c][2] = x2[c * rs + cb];
  }
  for (int i = tid; i < C * k; i += 256) ws[i] = A.w[i];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    float v = ((va[c][0] + f1 * va[c][1]) + f2 * va[c][2]) * in_scale;     // same order as the generic kernel: (x0 + x1) + x2
    v = v > 0.f ? v : v * slope;
    xs[c * W + tid] = oka ? v : 0.f;
  }
  if (tid < k - 1) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float v = ((vb[c][0] + f1 * vb[c][1]) + f2 * vb[c][2]) * in_scale;
      v = v > 0.f ? v : v * slope;
      xs[c * W + 256 + tid] = okb ? v : 0.f;
    }
  }
  __syncthreads();
  const int t = t0 + tid;
  if (t >= A.L) return;
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c)
    for (int j = 0; j < k; ++j) acc += ws[c * k + j] * xs[c * W + tid + j];
  A.out[(int64_t)b * A.out_bstride + t] = tanhf(acc);
}

int launch_conv_post(hipStream_t stream, const ConvPostArgs& a) {
  dim3 grid((a.L + 255) / 256, a.B);
  if (a.C == 16 && a.k <= 65 && (int64_t)a.C * a.x_rstride < (1ll << 31)) {
    hipLaunchKernelGGL(conv_post_c_kernel<16>, grid, dim3(256), sizeof(float) * (size_t)(a.C * a.k + a.C * (256 + a.k - 1)), stream, a);
    return BV2_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(conv_post_kernel, grid, dim3(256), sizeof(float) * (size_t)(a.C * a.k + a.C * (256 + a.k - 1)), stream, a);
  return BV2_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------------------------
// speaker-conditioning GEMVs: out[b][co] = bias[co] + W[co][:]·g[b][:]   (all the 1x1 convs / Linear applied to
// g [B,gin,1]: Encoder.spk_emb_linear attentions.py:108, sdp.cond models.py:201-203, dp.cond :288-289,
// dec.cond :541, WN.cond_layer modules.py:189-190).  One wave per output row, all problems in one launch.
// One output row of a speaker-conditioning GEMV for every batch item: the row's weights are loaded ONCE into registers (cin <= 512: 8 per
// lane) and four batch items run side by side, so their loads and the shuffle reductions overlap.  (Round 5: the first form walked the
// batch in a dependent loop — reload the row, reduce, store, next item: 73 us per launch at B = 32 for 0.1 MFLOP.)
template <typename GOF>
__device__ __forceinline__ void gemv_row(const float* w, int cin, int B, int lane, float bias, float* out, int64_t out_bstride, GOF gptr, int bz, int nbz) {
  if (cin <= 512) {
    float wv[8];
    int kk[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = lane + 64 * i;
      kk[i] = k < cin ? k : cin - 1;
      wv[i] = k < cin ? w[kk[i]] : 0.f;
    }
    // batch items in groups of four, dealt round-robin to the nbz workgroup layers of the grid (blockIdx.z): at B = 32 one wave walking all eight
    // groups was eight serial memory round trips — 66 us per launch
    for (int b0 = 4 * bz; b0 < B; b0 += 4 * nbz) {
      float acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int b = b0 + j < B ? b0 + j : B - 1;
        const float* g = gptr(b);
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) a += wv[i] * g[kk[i]];
        acc[j] = a;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += __shfl_xor(acc[j], off);
      if (lane == 0)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (b0 + j < B) out[(int64_t)(b0 + j) * out_bstride] = acc[j] + bias;
    }
    return;
  }
  for (int b = bz; b < B; b += nbz) {
    const float* g = gptr(b);
    float acc = 0.f;
    for (int k = lane; k < cin; k += 64) acc += w[k] * g[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) out[(int64_t)b * out_bstride] = acc + bias;
  }
}

static inline int gemv_layers(int B) { const int g = (B + 3) / 4; return g < 1 ? 1 : (g > 16 ? 16 : g); }   // grid.z: groups of four batch items side by side

__global__ void __launch_bounds__(256) gemv_kernel(const GemvLaunch L) {
  const GemvProb& P = L.p[blockIdx.y];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wid;
  if (row >= P.cout) return;
  const float* w = P.w + (int64_t)row * P.cin;
  const float* gbase = L.g;
  const int64_t gs = L.g_bstride;
  gemv_row(w, P.cin, L.B, lane, P.bias ? P.bias[row] : 0.f, P.out + row, P.out_bstride, [=](int b) { return gbase + (int64_t)b * gs; }, blockIdx.z, gridDim.z);
}

int launch_gemv(hipStream_t stream, const GemvLaunch& L) {
  if (L.nprob < 1 || L.nprob > 16) return -1;
  int maxc = 0;
  for (int i = 0; i < L.nprob; ++i) maxc = L.p[i].cout > maxc ? L.p[i].cout : maxc;
  dim3 grid((maxc + 3) / 4, L.nprob, gemv_layers(L.B));
  hipLaunchKernelGGL(gemv_kernel, grid, dim3(256), 0, stream, L);
  return BV2_CHECK_LAUNCH();
}

// phase-A front: the GEMVs above with g optionally looked up through sid (emb_g, models.py:1046), plus — in the extra block row —
// g_out, x_mask = sequence_mask(x_lengths) (commons.py:119-123) and z = noise * noise_scale_w (models.py:248-251).
__global__ void __launch_bounds__(256) front_kernel(const FrontArgs A) {
  if ((int)blockIdx.y < A.nprob) {
    const GemvProb& P = A.p[blockIdx.y];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wid;
    if (row >= P.cout) return;
    const float* w = P.w + (int64_t)row * P.cin;
    const int64_t* sid = A.sid;
    const float* table = A.table;
    const float* gbase = A.g;
    const int64_t gs = A.g_bstride, nrows = A.nrows;
    const int gin = A.gin;
    gemv_row(w, P.cin, A.B, lane, P.bias ? P.bias[row] : 0.f, P.out + row, P.out_bstride, [=](int b) {
      if (sid) {
        int64_t r = sid[b];
        r = r < 0 ? 0 : (r >= nrows ? nrows - 1 : r);
        return table + r * gin;
      }
      return gbase + (int64_t)b * gs;
    }, blockIdx.z, gridDim.z);
    return;
  }
  if (blockIdx.z) return;                          // the elementwise tail below runs in layer 0 only
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, step = (int64_t)gridDim.x * 256;
  if (A.sid && A.g_out)
    for (int64_t i = i0; i < (int64_t)A.B * A.gin; i += step) {
      const int b = (int)(i / A.gin), c = (int)(i - (int64_t)b * A.gin);
      int64_t r = A.sid[b];
      r = r < 0 ? 0 : (r >= A.nrows ? A.nrows - 1 : r);
      A.g_out[i] = A.table[r * A.gin + c];
    }
  if (A.mask)
    for (int64_t i = i0; i < (int64_t)A.B * A.T; i += step) {
      const int b = (int)(i / A.T), t = (int)(i - (int64_t)b * A.T);
      A.mask[i] = (!A.lengths || (int64_t)t < A.lengths[b]) ? 1.f : 0.f;
    }
  if (A.z)
    for (int64_t i = i0; i < A.nz; i += step) A.z[i] = A.noise[i] * A.noise_scale;
}

int launch_front(hipStream_t stream, const FrontArgs& a) {
  if (a.nprob < 0 || a.nprob > 16 || a.B < 1) return -1;
  if (a.nprob > 0 && !a.sid && !a.g) return -1;
  if (a.sid && (!a.table || a.nrows < 1 || a.gin < 1)) return -1;
  int maxc = 4;
  for (int i = 0; i < a.nprob; ++i) maxc = a.p[i].cout > maxc ? a.p[i].cout : maxc;
  dim3 grid((maxc + 3) / 4, a.nprob + 1, gemv_layers(a.B));
  hipLaunchKernelGGL(front_kernel, grid, dim3(256), 0, stream, a);
  return BV2_CHECK_LAUNCH();
}

// g = emb_g(sid)  (models.py:1046)
__global__ void gather_rows_kernel(const float* table, const int64_t* idx, float* out, int C, int nrows) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  int64_t r = idx[b];
  r = r < 0 ? 0 : (r >= nrows ? nrows - 1 : r);
  out[(int64_t)b * C + c] = table[r * C + c];
}
int launch_gather_rows(hipStream_t stream, const float* table, const int64_t* idx, float* out, int B, int C, int nrows) {
  hipLaunchKernelGGL(gather_rows_kernel, dim3((C + 255) / 256, B), dim3(256), 0, stream, table, idx, out, C, nrows);
  return BV2_CHECK_LAUNCH();
}

// x_mask = sequence_mask(x_lengths, T)  (commons.py:119-123)
__global__ void seq_mask_kernel(const int64_t* lengths, float* mask, int T) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < T) mask[(int64_t)b * T + t] = (!lengths || (int64_t)t < lengths[b]) ? 1.f : 0.f;      // lengths == null: all valid
}
// cap[b] = max over the batch of lengths[] for every b (bv2_decode_in.exact_lengths == 2: the Generator of a run padded to a T_y bucket treats
// every position past the LONGEST utterance as zero padding, i.e. sees exactly the tensor the reference's dec sees, commons.py:119-123 / models.py:1073)
__global__ void len_cap_kernel(const int64_t* lengths, int64_t* cap, int B) {
  long long mx = 0;
  for (int b = threadIdx.x; b < B; b += 64) mx = lengths[b] > mx ? lengths[b] : mx;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const long long v = __shfl_xor(mx, o);
    mx = v > mx ? v : mx;
  }
  for (int b = threadIdx.x; b < B; b += 64) cap[b] = mx;
}
int launch_len_cap(hipStream_t stream, const int64_t* lengths, int64_t* cap, int B) {
  hipLaunchKernelGGL(len_cap_kernel, dim3(1), dim3(64), 0, stream, lengths, cap, B);
  return BV2_CHECK_LAUNCH();
}

int launch_seq_mask(hipStream_t stream, const int64_t* lengths, float* mask, int B, int T) {
  hipLaunchKernelGGL(seq_mask_kernel, dim3((T + 255) / 256, B), dim3(256), 0, stream, lengths, mask, T);
  return BV2_CHECK_LAUNCH();
}

// TextEncoder front (models.py:381-396): three embedding lookups + the summed BERT projections, * sqrt(hidden), * mask
__global__ void embed_kernel(const EmbedArgs A) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= A.T) return;
  const int64_t bt = (int64_t)b * A.T + t;
  int64_t xi = A.x[bt], ti = A.tone[bt], li = A.lang[bt];
  xi = xi < 0 ? 0 : (xi >= A.n_vocab ? A.n_vocab - 1 : xi);
  ti = ti < 0 ? 0 : (ti >= A.n_tones ? A.n_tones - 1 : ti);
  li = li < 0 ? 0 : (li >= A.n_langs ? A.n_langs - 1 : li);
  const int64_t off = ((int64_t)b * A.C + c) * A.T + t;
  float v = A.emb[xi * A.C + c] + A.tone_emb[ti * A.C + c];
  v += A.lang_emb[li * A.C + c];
  float bs = 0.f;
  const int per = A.nslab / 3;                    // slabs per feature (the three projections write nslab/3 slabs each)
  for (int f = 0; f < 3; ++f) {
    int tc = t;
    if (A.idx[f]) {
      // word-level hand-over: clamp to the feature's own column count (columns >= cols of the projected slab hold the conv's
      // bias-only output: an out-of-range word index must not silently read them)
      const int nc = A.cols[f] > 0 && A.cols[f] < A.T ? A.cols[f] : A.T;
      tc = A.idx[f][bt];
      tc = tc < 0 ? 0 : (tc >= nc ? nc - 1 : tc);
    }
    const int64_t offf = ((int64_t)b * A.C + c) * A.T + tc;
    for (int sl = f * per; sl < (f + 1) * per; ++sl) bs += A.bsum[(int64_t)sl * A.slab_stride + offf];
  }
  v += bs;
  A.out[off] = v * A.scale * A.mask[bt];
}
int launch_embed(hipStream_t stream, const EmbedArgs& a) {
  hipLaunchKernelGGL(embed_kernel, dim3((a.T + 255) / 256, a.C, a.B), dim3(256), 0, stream, a);
  return BV2_CHECK_LAUNCH();
}

__global__ void add_vec_mask_kernel(const float* a, const float* vec, int vec_bstride, const float* mask, float* out,
                                    int C, int T) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int64_t off = ((int64_t)b * C + c) * T + t;
  float v = a[off];
  if (vec) v += vec[(int64_t)b * vec_bstride + c];
  if (mask) v *= mask[(int64_t)b * T + t];
  out[off] = v;
}
int launch_add_vec_mask(hipStream_t stream, const float* a, const float* vec, int vec_bstride, const float* mask,
                        float* out, int B, int C, int T) {
  hipLaunchKernelGGL(add_vec_mask_kernel, dim3((T + 255) / 256, C, B), dim3(256), 0, stream, a, vec, vec_bstride, mask,
                     out, C, T);
  return BV2_CHECK_LAUNCH();
}

// modules.Flip (reference modules.py:374-381: torch.flip(x, [1])) as data movement, in place: z[b][c][t] <-> z[b][C-1-c][t].  Only for a flow
// with an ODD number of couplings, once per reverse pass (bv2_exec.cpp flow_core): the other Flips are folded into the packed pre / post
// weights, which cancels out only for an even count
__global__ void flip_channels_kernel(float* z, int C, int T) {
  const int b = blockIdx.z, c = blockIdx.y;       // c < C / 2
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int64_t lo = ((int64_t)b * C + c) * T + t, hi = ((int64_t)b * C + (C - 1 - c)) * T + t;
  const float a = z[lo], h = z[hi];
  z[lo] = h; z[hi] = a;
}
int launch_flip_channels(hipStream_t stream, float* z, int B, int C, int T) {
  if (C < 2) return 0;
  hipLaunchKernelGGL(flip_channels_kernel, dim3((T + 255) / 256, C / 2, B), dim3(256), 0, stream, z, C, T);
  return BV2_CHECK_LAUNCH();
}

__global__ void scale_kernel(const float* in, float* out, float s, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[i] * s;
}
int launch_scale(hipStream_t stream, const float* in, float* out, float s, int64_t n) {
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, in, out, s, n);
  return BV2_CHECK_LAUNCH();
}

// ConvFlow.pre (1 -> C, k=1) fused with DDSConv's `x = x + g` (modules.py:488-489, 119-120)
__global__ void convflow_pre_kernel(const float* z, int src, const float* w, const float* bias, const float* g, float* h,
                                    int C, int T) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int64_t off = ((int64_t)b * C + c) * T + t;
  h[off] = w[c] * z[((int64_t)b * 2 + src) * T + t] + bias[c] + g[off];
}
int launch_convflow_pre(hipStream_t stream, const float* z, int src, const float* w, const float* bias, const float* g,
                        float* h, int B, int C, int T) {
  hipLaunchKernelGGL(convflow_pre_kernel, dim3((T + 255) / 256, C, B), dim3(256), 0, stream, z, src, w, bias, g, h, C, T);
  return BV2_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------------------------
// inverse piecewise rational-quadratic spline (spline.h), one thread per (b, t); always fp32.
__global__ void spline_kernel(float* z, int src, int dst, const float* params, int prow, const float* mask,
                              float sqrt_fc, float tail, float cst, float wscale, int T) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 128 + threadIdx.x;
  if (t >= T) return;
  const float mk = mask[(int64_t)b * T + t];
  const float* p = params + (int64_t)b * prow * T + t;
  float uw[SPK], uh[SPK], ud[SPK + 1];
#pragma unroll
  for (int i = 0; i < SPK; ++i) { uw[i] = p[(int64_t)i * T] / sqrt_fc; uh[i] = p[(int64_t)(SPK + i) * T] / sqrt_fc; }
  ud[0] = cst; ud[SPK] = cst;
#pragma unroll
  for (int i = 1; i < SPK; ++i) ud[i] = p[(int64_t)(2 * SPK + i - 1) * T];

  float* zs = z + ((int64_t)b * 2 + src) * T + t;
  float* zd = z + ((int64_t)b * 2 + dst) * T + t;
  const float outv = rq_spline_inverse_one(*zd, uw, uh, ud, tail, wscale);
  *zd = outv * mk;
  *zs = *zs * mk;
}
int launch_spline(hipStream_t stream, float* z, int src, int dst, const float* params, int params_rows,
                  const float* mask, float sqrt_fc, float tail_bound, int B, int T) {
  // constants the reference evaluates in Python doubles before they meet fp32 tensors (transforms.py:71, :128)
  const float cst = (float)std::log(std::exp(1.0 - 1e-3) - 1.0);
  const float wscale = (float)(1.0 - 1e-3 * SPK);
  hipLaunchKernelGGL(spline_kernel, dim3((T + 127) / 128, B), dim3(128), 0, stream, z, src, dst, params, params_rows,
                     mask, sqrt_fc, tail_bound, cst, wscale, T);
  return BV2_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------------------------
// durations: ElementwiseAffine^-1 on channel 0, mix sdp/dp, exp * mask * length_scale, ceil, sum -> y_lengths
// reference modules.py:397-399, models.py:1052-1057.  One workgroup per utterance.
__global__ void __launch_bounds__(256) durations_kernel(const DurArgs A) {
  __shared__ float part[256];
  const int b = blockIdx.x;
  const float m0 = A.ea_m[0], il0 = expf(-A.ea_logs[0]);
  float s = 0.f;
  for (int t = threadIdx.x; t < A.T; t += 256) {
    const int64_t bt = (int64_t)b * A.T + t;
    const float mk = A.mask[bt];
    const float ls = (A.z[((int64_t)b * 2) * A.T + t] - m0) * il0 * mk;
    if (A.logw_sdp) A.logw_sdp[bt] = ls;
    if (!A.logw_dp) continue;                      // bv2_stage_sdp: only the ElementwiseAffine inverse is wanted
    const float ld = A.logw_dp[bt];
    const float lw = ls * A.sdp_ratio + ld * A.one_minus_ratio;
    const float w = expf(lw) * mk * A.length_scale;
    const float wc = ceilf(w);
    A.logw[bt] = lw;
    A.w_ceil[bt] = wc;
    s += wc;
  }
  if (!A.logw_dp) return;                          // kernel-uniform
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float tot = part[0];
    A.y_lengths[b] = tot < 1.f ? 1 : (int64_t)tot;
  }
}
int launch_durations(hipStream_t stream, const DurArgs& a) {
  hipLaunchKernelGGL(durations_kernel, dim3(a.B), dim3(256), 0, stream, a);
  return BV2_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------------------------
// length regulation.  Reference: generate_path builds a one-hot [B,Ty,T] matrix from cumsum(w_ceil) and multiplies
// m_p / logs_p by it (commons.py:126-140, models.py:1061-1069); here frame j looks its symbol up directly:
// symbol i owns frames [cum[i-1], cum[i]).  Pass 1 (one workgroup per utterance): scan + scatter frame->symbol.
// Pass 2: gather + prior sampling z_p = m + noise*exp(logs)*noise_scale (models.py:1071).
__global__ void __launch_bounds__(256) frame_index_kernel(const ExpandArgs A) {
  __shared__ int part[256];
  __shared__ int carry_s;
  const int b = blockIdx.x;
  const int ylen = (int)A.y_lengths[b];
  int* fi = A.frame_idx + (int64_t)b * A.Ty;
  for (int j = threadIdx.x; j < A.Ty; j += 256) {
    fi[j] = -1;
    if (A.y_mask) A.y_mask[(int64_t)b * A.Ty + j] = j < ylen ? 1.f : 0.f;
  }
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  // symbols in chunks of 256: inclusive scan of durations, then each symbol writes its own frames
  for (int base = 0; base < A.T; base += 256) {
    const int i = base + threadIdx.x;
    int d = 0;
    if (i < A.T) d = (int)(A.w_ceil[(int64_t)b * A.T + i] * (A.x_mask[(int64_t)b * A.T + i] != 0.f ? 1.f : 0.f));
    part[threadIdx.x] = d;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
      __syncthreads();
      part[threadIdx.x] += v;
      __syncthreads();
    }
    const int end = carry_s + part[threadIdx.x];
    const int start = end - d;
    for (int j = start; j < end && j < ylen && j < A.Ty; ++j) fi[j] = i;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = end;
    __syncthreads();
  }
}

__global__ void expand_kernel(const ExpandArgs A) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= A.Ty) return;
  const int i = A.frame_idx[(int64_t)b * A.Ty + j];
  float m = 0.f, lg = 0.f;
  if (i >= 0) {
    m = A.m_p[((int64_t)b * A.C + c) * A.T + i];
    lg = A.logs_p[((int64_t)b * A.C + c) * A.T + i];
  }
  const int64_t off = ((int64_t)b * A.C + c) * A.Ty + j;
  const float nz = A.noise[(int64_t)b * A.nz_bstride + (int64_t)c * A.nz_cstride + (int64_t)j * A.nz_tstride];
  const float zp = m + nz * expf(lg) * A.noise_scale;
  A.z_p[off] = zp;
  if (A.z_p2) A.z_p2[off] = zp;
  if (A.m_e) A.m_e[off] = m;
  if (A.logs_e) A.logs_e[off] = lg;
}

__global__ void attn_path_kernel(const int* frame_idx, float* attn, int T, int Ty) {
  const int b = blockIdx.z, j = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= T) return;
  attn[((int64_t)b * Ty + j) * T + i] = (frame_idx[(int64_t)b * Ty + j] == i) ? 1.f : 0.f;
}

int launch_expand(hipStream_t stream, const ExpandArgs& a0) {
  ExpandArgs a = a0;
  if (a.nz_tstride <= 0) a.nz_tstride = 1;
  hipLaunchKernelGGL(frame_index_kernel, dim3(a.B), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(expand_kernel, dim3((a.Ty + 255) / 256, a.C, a.B), dim3(256), 0, stream, a);
  if (a.attn)
    hipLaunchKernelGGL(attn_path_kernel, dim3((a.T + 255) / 256, a.Ty, a.B), dim3(256), 0, stream, a.frame_idx, a.attn,
                       a.T, a.Ty);
  return BV2_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------------------------
// 16-bit PCM (what the reference's callers do on the host after .cpu(): gradio convert_to_16_bit_wav, webui.py:86,
// hiyoriUI.py:343): per utterance  pcm = int16( wave / max|wave| * 32767 )  (truncation, as numpy astype), over the valid
// samples [0, y_length*hop); samples past the utterance are 0.  peak[] holds the bit pattern of max|wave| (non-negative
// floats order like unsigned integers) and must be zero on entry (the launcher clears it on the stream).
__global__ void __launch_bounds__(256) pcm_peak_kernel(const float* wave, int64_t bstride, const int64_t* y_lengths, int hop,
                                                       int64_t S, unsigned* peak) {
  const int b = blockIdx.y;
  int64_t n = y_lengths[b] * hop;
  n = n < S ? n : S;
  const float* w = wave + (int64_t)b * bstride;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 3 < n && ((reinterpret_cast<uintptr_t>(w + i) & 15) == 0)) {
      const float4 v = *reinterpret_cast<const float4*>(w + i);
      m = fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
    } else {
      for (int e = 0; e < 4 && i + e < n; ++e) m = fmaxf(m, fabsf(w[i + e]));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(peak + b, __float_as_uint(m));
}
__global__ void __launch_bounds__(256) pcm_convert_kernel(const float* wave, int64_t bstride, const int64_t* y_lengths, int hop,
                                                          int64_t S, const unsigned* peak, int16_t* pcm, int64_t pstride) {
  const int b = blockIdx.y;
  int64_t n = y_lengths[b] * hop;
  n = n < S ? n : S;
  const float pk = __uint_as_float(peak[b]);
  const float* w = wave + (int64_t)b * bstride;
  int16_t* o = pcm + (int64_t)b * pstride;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < S; i += (int64_t)gridDim.x * 256) {
    // numpy: (x / peak) * 32767 in fp32, then astype(int16) truncates toward zero
    o[i] = (i < n && pk > 0.f) ? (int16_t)(int)((w[i] / pk) * 32767.f) : (int16_t)0;
  }
}
int launch_pcm16(hipStream_t stream, const float* wave, int64_t bstride, const int64_t* y_lengths, int hop, int B, int64_t S,
                 int16_t* pcm, int64_t pstride, unsigned* peak) {
  if (B < 1 || S < 1 || hop < 1) return -1;
  if (hipMemsetAsync(peak, 0, sizeof(unsigned) * (size_t)B, stream) != hipSuccess) return -1;
  int gx = (int)((S + 4095) / 4096);
  gx = gx < 1 ? 1 : (gx > 512 ? 512 : gx);
  hipLaunchKernelGGL(pcm_peak_kernel, dim3(gx, B), dim3(256), 0, stream, wave, bstride, y_lengths, hop, S, peak);
  hipLaunchKernelGGL(pcm_convert_kernel, dim3(gx, B), dim3(256), 0, stream, wave, bstride, y_lengths, hop, S, peak, pcm, pstride);
  return BV2_CHECK_LAUNCH();
}

}  // namespace bv2

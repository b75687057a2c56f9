// bv2_kernels.h — launcher interface between the host executor (bv2_exec.cpp) and the gfx950 kernels (kernels/*.hip).
// Every launcher is asynchronous on `stream`, allocates nothing, and is hipGraph-capturable.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <map>
#include <mutex>
#include <utility>

namespace bv2 {

// Raise a kernel's dynamic-LDS limit ONCE per (device, kernel) instead of on every launch: the attribute call is a driver round trip
// on the batch-1 latency path (~20 launchers x several launches per step).  Remembers the largest size granted so far.
// Batch-item -> XCD affinity: workgroup `id` of a 1-D launch runs on XCD id & 7; it takes item b = 8 * (slot / per) + (id & 7) and the
// r-th workgroup of that item, slot = id >> 3, r = slot % per.  Returns false for the padding workgroups of a batch that is not a
// multiple of 8 (uniform per workgroup: safe to exit on).
#if defined(__HIPCC__)
__device__ __forceinline__ bool xcd_decode(unsigned id, int per, int B, int& b, int& r) {
  const unsigned slot = id >> 3;
  b = (int)(slot / (unsigned)per) * 8 + (int)(id & 7u);
  r = (int)(slot % (unsigned)per);
  return b < B;
}
#endif
inline int xcd_grid(int B, int per) { return 8 * ((B + 7) / 8) * per; }

// Weight prefetch tail (round 5).  In the small-N regime (batch 1, one BERT sentence) every launch is a dependent ~10 us kernel whose
// first act is to pull its weights (0.5-17 MB) from HBM / the Infinity Cache, while the launch in front of it (a LayerNorm on 7-48
// workgroups, or a GEMM whose own weights are already on chip) leaves HBM idle.  A launch can therefore carry `pf`: the NEXT launch's
// weight stream, which PF_BLOCKS spare workgroups of THIS launch touch line by line (one dword per 128-byte line) so that it sits in L2
// when the consumer starts.  XCD-local: the split-K kernel gives XCD x the x-th eighth of the m-tiles (its virtual ids are contiguous per
// XCD and the packed stream is m-tile-major), so the prefetch workgroup with hardware id l (XCD l & 7) touches the same eighth.
struct Prefetch { const void* ptr; unsigned bytes; };
constexpr int PF_BLOCKS = 128;                       // 16 per XCD
#if defined(__HIPCC__)
// l: the workgroup's linear hardware id, j: its index among the PF_BLOCKS prefetch workgroups (both wave-uniform)
__device__ __forceinline__ void prefetch_tail(const Prefetch pf, unsigned l, unsigned j, unsigned tid, unsigned nthreads) {
  const unsigned lines = (pf.bytes + 127u) >> 7;
  const unsigned lx = (lines + 7u) >> 3;                          // lines per XCD region
  const unsigned per = PF_BLOCKS / 8, lc = (lx + per - 1u) / per; // lines per workgroup
  const unsigned xcd = l & 7u, c = j >> 3;
  const unsigned l0 = xcd * lx + c * lc;
  unsigned l1 = l0 + lc;
  if (l1 > (xcd + 1u) * lx) l1 = (xcd + 1u) * lx;
  if (l1 > lines) l1 = lines;
  const char* base = static_cast<const char*>(pf.ptr);
  for (unsigned i = l0 + tid; i < l1; i += nthreads) {
    unsigned sink;
    asm volatile("global_load_dword %0, %1, off" : "=v"(sink) : "v"(base + (size_t)i * 128u) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#endif

inline void ensure_dyn_lds(const void* kern, size_t lds) {
  if (lds <= 64 * 1024) return;
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> granted;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  size_t& g = granted[{dev, kern}];
  if (lds > g && hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) g = lds;
}

// --------------------------------------------------------------------------------------------------------------
// conv1d as implicit GEMM on fp32 MFMA (kernels/conv_mfma.hip)
//
//   out[b][co][t*out_tstride + out_toff] = epilogue( bias[co] + bias2[b][co]
//        + sum_{ci<cin, j<k} Wp[j][ci][co] * pre( in_scale * sum_s x_s[b][ci][t - pad_left + j*dil] * in_mask[b][.] ) )
//
// Wp is the PACKED weight in MFMA "fragment order", one CONTIGUOUS stream per 32-row output tile:
//   [m-tile = co/32][group = ci/8][tap j][lh = ci&1][co%32][q = (ci%8)/2]   (fp32; see conv_w_index)
// so the A operands of four consecutive K steps are one aligned float4 per lane, a (group, tap) unit of one m-tile is
// 1 KB contiguous, and a workgroup's weight stream is sequential in memory (the previous [tap][group][lh][C_out][4]
// layout made every workgroup read 512-byte pieces at a multi-KB power-of-two-ish stride, which camped on 2 of the 16
// L2 channels of an XCD: measured 125 GB/s of weight streaming).  cin_pad % 16 == 0; w_ld % 128 == 0 is the number of
// rows ALLOCATED (so any tile height may read whole m-tiles; rows >= cout are zero); cout_pad = cout rounded up to 32
// bounds the tiling.  One launch can carry several problems
// (blockIdx.z) that share B and L: the three ResBlock branches of a Generator stage, the u polyphase branches of a
// ConvTranspose1d, or the m_p / logs_p halves of enc_p.proj.
enum { PRE_NONE = 0, PRE_LRELU = 1 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GATE = 2, ACT_GELU = 3 };   // ACT_GELU: exact erf GELU (BERT intermediate, bv2_bert.cpp)
// ACT_GATE (WN, reference commons.py:98-105): the GEMM rows come packed so that rows [0,16) of every 32-row tile are the tanh half and
// rows [16,32) the sigmoid half of the same 16 channels (bv2_model.cpp wn_gate_row); the epilogue writes
// out[16*mt + j] = tanh(v[j]) * sigmoid(v[j+16]) — `out` has cout/2 rows.  No residual / masks with it.
enum { RES_NONE = 0, RES_ADD = 1, RES_RSUB = 2 };   // v += res  |  v = res - v

struct ConvProb {
  const float* x[3];        // input sources (summed); x[1], x[2] may be null
  int nsrc;
  float in_scale;
  int64_t x_bstride;        // floats between batches
  int x_rstride;            // floats between channels (row stride)
  int Lin;                  // valid input length (reads outside [0,Lin) are zero padding)
  const float* in_mask;     // [b*in_mask_bstride + t] or null
  int in_mask_bstride;
  const float* w;           // packed weights
  const uint16_t* w6;       // the same weights as three bf16 planes (x6_w_index; kernels/conv_x6.hip) or null
  const float* bias;        // [cout_pad] or null
  const float* bias2;       // [B][bias2_bstride] per-batch bias or null
  int bias2_bstride;
  float* out;
  int64_t out_bstride;
  int out_rstride, out_tstride, out_toff;
  const float* res;         // residual, indexed like out (own batch stride)
  int64_t res_bstride;
  const float* out_mask;    // [b*out_mask_bstride + t] or null
  int out_mask_bstride;
  int cin, cin_pad, cout, cout_pad, w_ld, k, dil, pad_left;
  int pre_act;  float slope;
  int act;                  // applied to acc+bias
  int mask_pre;             // multiply by out_mask before the residual op
  int res_mode;
  int mask_post;            // multiply by out_mask after the residual op
  // conv_x6.hip's two-plane fp16 form ("x3", below): the weights as two fp16 planes (x3_w_index) scaled by a power of two, that
  // scale's reciprocal, and the slot that holds max|x| of the INPUT tensor as fp32 bits (written by the launch that produced x).
  // All three set: the x3 form; any null: the three-plane bf16 form (w6).
  const uint16_t* w3;
  const float* w3inv;
  const unsigned* xmax;     // X3_SLOT_WORDS words
  unsigned* omax;           // any conv kernel: atomicMax of |out| over everything this problem stores into the slot (X3_SLOT_WORDS words), or null
};

#define BV2_MAX_PROBS 8
#define BV2_MAX_KSPLIT 8
// "exact lengths" (every launch struct below): when `lens` is set, batch item b's INPUT is valid only on
// [0, min(Lin, lens[b]*len_mul)) — positions past it read as the conv's zero padding, exactly as if the utterance had been
// run alone.  The reference's decoder is unmasked, so in a padded batch the bias-driven activations past an utterance's end
// bleed into its last samples; with lens a ragged batch reproduces the per-utterance result.
struct ConvLaunch {
  ConvProb p[BV2_MAX_PROBS];
  int nprob;
  int B;
  int L;                    // output positions per problem (index t)
  const int64_t* lens = nullptr; int len_mul = 1;
  int ksplit;               // split-K kernel only: K split across workgroups into `ksplit` partial slabs (1 = none)
  int64_t slab_stride;      // floats between the partial slabs of one output (slab z is written at out + z*slab_stride)
  unsigned long long* dbg = nullptr;   // tools/timeline.py only: 8 u64 per workgroup (s_memtime stamps + HW ids); null in the product
  Prefetch pf = {nullptr, 0};          // split-K kernel only: the NEXT launch's weight stream, touched by PF_BLOCKS spare workgroups
};
// tools/timeline.py: while a device buffer is set, every conv launch records per-workgroup timestamps into its own slice of it
void conv_set_timeline(unsigned long long* dev_buf, long long capacity_u64);
int conv_timeline_report(long long* meta, int max_launches);
// a launch's slice of the timeline buffer (null when none is set); tile / ks / cin / L are only recorded for the report
unsigned long long* timeline_slice(unsigned gx, unsigned gy, unsigned gz, int tile, int ks, int cin, int L);   // per launch: {offset_u64, gx, gy, gz, tile id, nprob k0 | k1<<8 | k2<<16, cin, L}

// tile: 0 = auto; otherwise one of the TILE_* ids (tests force each variant)
enum { TILE_AUTO = 0, TILE_128x128 = 1, TILE_64x128 = 2, TILE_64x64 = 3, TILE_32x128 = 4, TILE_32x256 = 5, TILE_SPLITK = 6, TILE_128x64 = 7,
       TILE_X6 = 8,            // the split-bf16 form of the LDS-tiled kernel (kernels/conv_x6.hip); TILE_AUTO picks it when every problem has w6
       TILE_X6_128x64 = 9, TILE_X6_128x64_LD = 10, TILE_X6_64x128 = 11, TILE_X6_32x256 = 12,     // tests / tuning: one x6 tile forced
       TILE_X3 = 14 };         // tests: the x6 choice with the two-plane fp16 form where the tile has one (the product takes it whenever w3 / w3inv / xmax are set)
// (13 was TILE_SPLITK_X6, the split-K kernel's split-bf16 form: measured at +0.4 % on config 2 for +195 MB of blob and deleted in round 6 —
//  DESIGN.md "measured and not kept", profiles/r05_ab_splitk_x6_not_kept.txt)
int launch_conv1d(hipStream_t stream, const ConvLaunch& L, int tile, const char** variant_name);
// fp32 conv on the bf16 matrix core (kernels/conv_x6.hip): every fp32 operand is the exact sum of three bf16 values
// (v = h1 + h2 + h3, 8 + 8 + 8 significand bits), and the product is accumulated from the six largest of the nine cross terms
// (w1x1, w1x2, w2x1, w1x3, w2x2, w3x1 — the three dropped ones are below 2^-23 of the product, the rounding of an fp32 multiply).
// Weight planes are split at pack time: element (tap j, ci, co) of plane p lives at x6_w_index(...) of a uint16 stream
//   [m-tile = co/32][group s = ci/16][tap j][plane p][lane = co%32 + 32*((ci%16)/8)][ci%8]
// i.e. the A operand of one v_mfma_f32_32x32x16_bf16 per (unit, plane) is one 16-byte load per lane, a unit is 3 KB contiguous.
inline int64_t x6_w_index(int j, int ci, int co, int cin, int k, int plane) {
  const int64_t U = (int64_t)(cin / 16) * k, u = (int64_t)(ci / 16) * k + j;
  const int lane = (co & 31) + 32 * ((ci % 16) / 8);
  return ((((int64_t)(co >> 5) * U + u) * 3 + plane) * 64 + lane) * 8 + (ci % 8);
}
inline int64_t x6_w_elems(int cin, int cout_pad, int k) { return (int64_t)(cout_pad / 32) * (cin / 16) * k * 3 * 512; }
// v = h[0] + h[1] + h[2] exactly (round-to-nearest-even at every step), as bf16 bit patterns.  The FIRST plane saturates: a finite
// |v| above the largest bf16 (0x7f7f = 3.3895e38; RNE would round the top 0.2 % of the fp32 range to +-inf and the remainder v - inf
// to NaN) takes +-0x7f7f and the remainder v - h[0] < 2^120 is still exact in the two planes that follow — so every FINITE fp32
// splits into three finite planes.  +-inf splits into (+-0x7f7f, +-inf, NaN), NaN into NaN: non-finite stays non-finite.
inline void x6_split(float v, uint16_t h[3]) {
  for (int p = 0; p < 3; ++p) {
    uint32_t u;
    __builtin_memcpy(&u, &v, 4);
    uint16_t b;
    if ((u & 0x7fffffffu) > 0x7f800000u) b = (uint16_t)((u >> 16) | 0x40u);
    else if (p == 0 && (u & 0x7fffffffu) > 0x7f7f0000u) b = (uint16_t)((u >> 16) & 0x8000u) | 0x7f7fu;
    else { u += 0x7fffu + ((u >> 16) & 1u); b = (uint16_t)(u >> 16); }
    h[p] = b;
    const uint32_t w = (uint32_t)b << 16;
    float f;
    __builtin_memcpy(&f, &w, 4);
    v -= f;
  }
}
// The two-plane fp16 form of the same kernel ("x3": three products instead of six).  v * S = g0 + g1 with g0 = fp16(v * S) and
// g1 = fp16(v * S - g0), S a power of two that puts the tensor's largest magnitude just below 2^15: |v * S - g0 - g1| <= 2^-24 |v * S|
// for every element within 2^15 of the largest (fp16 has 11 significand bits and subnormals down to 2^-24; smaller elements keep an
// ABSOLUTE error <= 2^-25, i.e. 2^-40 of the largest), and of the four cross terms the three largest, w0x0 + (w0x1 + w1x0), leave out
// |w1x1| <= 2^-24 |wx| — the same order as the six-product bf16 form, at half the matrix work.  fp16 has no exponent range to spare,
// hence the scales: S_w per weight tensor at pack time (x3_scale_exp of its max |w|), S_x per INPUT tensor from the max |x| slot its
// producer filled (ConvProb::omax -> ::xmax); the epilogue multiplies the accumulator by 1 / (S_w S_x), exact.
//   plane layout: [m-tile = co/32][group s = ci/16][tap j][plane p < 2][lane = co%32 + 32*((ci%16)/8)][ci%8]   (x6's with two planes)
inline int64_t x3_w_index(int j, int ci, int co, int cin, int k, int plane) {
  const int64_t U = (int64_t)(cin / 16) * k, u = (int64_t)(ci / 16) * k + j;
  const int lane = (co & 31) + 32 * ((ci % 16) / 8);
  return ((((int64_t)(co >> 5) * U + u) * 2 + plane) * 64 + lane) * 8 + (ci % 8);
}
inline int64_t x3_w_elems(int cin, int cout_pad, int k) { return (int64_t)(cout_pad / 32) * (cin / 16) * k * 2 * 512; }
// A max |x| slot is one 128-byte line per XCD (X3_SLOT_WORDS words, 128-byte aligned; word 0 of line x belongs to XCD x).  A wave
// publishes with an L2-LOCAL atomic (workgroup scope: executed in its XCD's L2, no trip to the memory side) into its own XCD's line —
// only that XCD ever writes the line, the end-of-kernel write-back makes it visible — and the reader takes the max of the eight words.
// (Device-scope atomics on one word: +12 us per launch of 1152 workgroups; a read of the word in front of the atomic: an exposed memory
// round trip at the end of every workgroup.)
constexpr int X3_LINE_WORDS = 32, X3_SLOT_WORDS = 8 * X3_LINE_WORDS;
constexpr int X3_HDR_FLOATS = 64;     // the packed region starts with 1 / S_w (one float, 256-byte slot), the planes follow
// biased fp32 exponent e of the tensor's largest magnitude, clamped so that S = 2^(141 - e) and 1 / S = 2^(e - 141) are normal floats:
// max < 2^(e - 126)  =>  max * S < 2^15
#if defined(__HIPCC__)
#define BV2_HD __host__ __device__
#else
#define BV2_HD
#endif
BV2_HD inline unsigned x3_scale_exp(unsigned max_bits) {
  unsigned e = (max_bits >> 23) & 255u;
  return e < 15u ? 15u : (e > 254u ? 254u : e);
}
BV2_HD inline float x3_scale(unsigned e) { const uint32_t u = (268u - e) << 23; float f; __builtin_memcpy(&f, &u, 4); return f; }
BV2_HD inline float x3_scale_inv(unsigned e) { const uint32_t u = (e - 14u) << 23; float f; __builtin_memcpy(&f, &u, 4); return f; }
// max |x| of a tensor into a slot (fp32 bits; the slot is zeroed by the caller): for inputs whose producer is not a conv launch
int launch_absmax(hipStream_t stream, const float* x, int64_t n, unsigned* slot);
int launch_x3_zero_slots(hipStream_t stream, unsigned* slots, int n_slots);   // n_slots x X3_SLOT_WORDS words, by a launch (not a memset node: conv_x6.hip)
bool conv_x6_supported(const ConvLaunch& L);        // every problem carries w6 and fits the staged tile
int launch_conv1d_x6(hipStream_t stream, const ConvLaunch& L, int tile, const char** variant_name);
void conv_x6_occupancy(int out[4]);                              // workgroups per CU granted to {128x64, 128x64 + loaders, 64x128, 32x256}
void conv_x6_set_tuning(int t256, int t128, int t64, int ck);   // tuning experiments only (tools/tune_x6.py); 0 = shipped choice
// true when TILE_AUTO will pick the split-K kernel for this launch (small-N regime); only then may ksplit exceed 1
bool conv_use_splitk(const ConvLaunch& L);
// K-split factor (power of two <= max_split) that fills the chip for a split-K launch whose consumer sums the slabs
int conv_pick_ksplit(const ConvLaunch& L, int max_split);
// float index of weight element (tap j, input channel ci, output channel co) in the packed layout
inline int64_t conv_w_index(int j, int ci, int co, int cin_pad, int k) {
  return ((((((int64_t)(co >> 5) * (cin_pad / 8) + ci / 8) * k + j) * 2 + (ci & 1)) * 32) + (co & 31)) * 4 + (ci % 8) / 2;
}
// tuning experiments only (tools/kbench.py through bv2_test_set_tuning): 0 = shipped heuristics
void conv_set_tuning(int splitk_waves, int force_ck, long tile_target);
double conv_flops(const ConvLaunch& L);
double conv_bytes(const ConvLaunch& L);

// --------------------------------------------------------------------------------------------------------------
// fused ResBlock1 pair for the narrow Generator stages (kernels/resblock_fused.hip):
//   out = x + conv2(lrelu(conv1(lrelu(x), k, dil) + b1), k, 1) + b2     on dense [B][C][L] tensors, C <= 32
struct FusedProb {
  const float* x; float* out;                     // out must not alias x (tiles read their neighbours' halo)
  const float* w1; const float* b1;               // packed weights (conv_w_index) / bias of convs1[d]
  const float* w2; const float* b2;               // ... of convs2[d]
  int k, dil;
  const uint16_t* w61; const uint16_t* w62;       // respair_x6.hip only: the two convs' split-bf16 weight planes (x6_w_index)
  const uint16_t* w31; const uint16_t* w32;       // ... their scaled fp16 planes (x3_w_index) and 1 / S_w: all four set = the x3 form
  const float* w3inv1; const float* w3inv2;
};
struct FusedLaunch { FusedProb p[3]; int nprob, B, C, L; float slope; const int64_t* lens = nullptr; int len_mul = 1;
                     unsigned long long* dbg = nullptr; };   // dbg: tools/timeline.py only
bool resblock_fused_supported(int C, int k, int dil);
int launch_resblock_fused(hipStream_t stream, const FusedLaunch& F);
double resblock_fused_flops(const FusedLaunch& F);
double resblock_fused_bytes(const FusedLaunch& F);
// the same pair at C = 32 with both convs on the bf16 matrix core from exact three-way bf16 splits (kernels/respair_x6.hip)
bool respair_x6_supported(int C, int k, int dil);
int launch_respair_x6(hipStream_t stream, const FusedLaunch& F);

// --------------------------------------------------------------------------------------------------------------
// bf16 Generator (kernels/gen_bf16.hip): channels-last activations [B][L][C] (C contiguous, bf16), bf16 MFMA
// (v_mfma_f32_32x32x16_bf16) with fp32 accumulation, fp32 bias / residual arithmetic, one bf16 rounding per stored tensor.
//
//   out[b][t][co] = bf16( bias[co] + bias2[b][co] + res[b][t][co]
//                         + sum_{ci<cin, j<k} W[co][ci][j] * pre( in_scale * sum_s x_s[b][t - pad_left + j*dil][ci] ) )
//   pre(v) = bf16( lrelu_slope(v) )   (identity when nsrc == 1 and pre_lrelu == 0)
//
// GEMM view: M = C_out (A operand = weights), N = time (B operand = activations from an LDS tile), K = (channel group, tap).
// Packed weight stream ("fragment order", bf16): [m-tile = co/32][unit u = (ci/16)*k + j][lane l][e < 8]
//   = W[co = mt*32 + (l&31)][ci = 16*(u/k) + 8*(l>>5) + e][tap u%k]        (1 KB per unit, one 16-byte load per lane)
// A ConvTranspose1d (stride u) is ONE such conv with C_out' = u*C_out: in channels-last memory the u output rows of input
// step t are contiguous, so out'[t][ph*C_out + co] IS out[t*u + ph][co]  (cl_w_index / bv2_model.cpp pack_up_cl).
struct ClProb {
  const uint16_t* x[3];     // sources [B][Lin][cin] bf16 (summed); x[1], x[2] may be null
  int nsrc; float in_scale;
  int64_t x_bstride;        // elements between batches
  int Lin;
  const uint16_t* w;        // packed bf16 fragments
  const float* bias;        // [cout_pad] fp32 or null
  const float* bias2;       // [B][bias2_bstride] fp32 per-batch bias or null
  int bias2_bstride;
  uint16_t* out;            // [B][L][cout]
  int64_t out_bstride;
  const uint16_t* res;      // residual [B][L][cout] or null (may alias out: each element is read then written by one lane)
  int64_t res_bstride;
  int cin, cout, cout_pad, k, dil, pad_left;
  int pre_lrelu; float slope;
  // ConvTranspose1d as ONE conv with C_out' = u*C_out (ph_cout > 0): output channels [ph*ph_cout, +ph_cout) are phase ph, whose
  // non-zero weights are the window taps [off, off + ph_ntaps), off = nibble ph of ph_offs — the GEMM of a wave runs only the taps of the
  // phases its output channels belong to (the others are zeros the packer wrote to make the u phases one problem)
  int ph_cout, ph_ntaps; unsigned long long ph_offs;
};
struct ClLaunch { ClProb p[BV2_MAX_PROBS]; int nprob, B, L; const int64_t* lens = nullptr; int len_mul = 1;
                  unsigned long long* dbg = nullptr;      // dbg: tools/timeline.py only
                  int ups = 0; };                          // a ConvTranspose1d launch (tile choice of its own: launch_conv_cl_bf16)
int launch_conv_cl_bf16(hipStream_t stream, const ClLaunch& L, const char** variant_name);
void conv_cl_set_tuning(const char* spec, int generic);      // tests / tuning only (bv2_test_set_variants)
bool conv_cl_bf16_supported(int cin, int cout, int k, int dil);
// element index (bf16 units) of weight (tap j, input channel ci, output channel co) in the packed stream
inline int64_t cl_w_index(int j, int ci, int co, int cin, int k) {
  const int64_t U = (int64_t)(cin / 16) * k, u = (int64_t)(ci / 16) * k + j;
  const int lane = (co & 31) + 32 * ((ci % 16) / 8);
  return (((int64_t)(co >> 5) * U + u) * 64 + lane) * 8 + (ci % 8);
}
inline int64_t cl_w_elems(int cin, int cout_pad, int k) { return (int64_t)(cout_pad / 32) * (cin / 16) * k * 512; }
double conv_cl_bytes(const ClLaunch& L);

// a whole ResBlock1 (nd (dilated conv, conv) pairs with residuals) of a narrow Generator stage in one launch, bf16
// channels-last, intermediates in LDS (kernels/resblock_cl_bf16.hip).  x / out: [B][L][C], C = 16 or 32, out != x.
//   w    : ONE contiguous bf16 fragment stream [d][conv e][unit u < Upad][lane][8]  (units of m-tile 0 in TAP-MAJOR order
//          u = tap*(C/16) + group, each unit laid out as in cl_w_index; Upad = resblock_cl_bf16_units(C, k) = (C/16)*k
//          rounded up to RBCL_PD, padding units zero) + RBCL_PD tail units
//   bias : fp32 [2*nd][32]  (row 2d+e = bias of conv e of pair d, zero padded to 32)
#define BV2_RBCL_MAX_D 4
constexpr int RBCL_PD = 8;        // weight ring depth = unit padding of the stream
struct RbClProb { const uint16_t* x; uint16_t* out; const uint16_t* w; const float* bias; int k; int dil[BV2_RBCL_MAX_D]; int halo; };
struct RbClLaunch { RbClProb p[3]; int nprob, B, C, L, nd; float slope; const int64_t* lens = nullptr; int len_mul = 1;
                    // stage hand-over (round 6, resblock_c16_bf16.hip only): non-null = ONE workgroup runs the nprob branches of its tile one after
                    // the other and the stage's output tensor — the branch mean with the rounding points of cl_bf16.h stage_mean — is written
                    // here instead of one tensor per branch into p[i].out
                    uint16_t* sum_out = nullptr; int halo_max = 0; float sum_scale = 1.f; };
bool resblock_cl_bf16_supported(int C, int k, const int* dil, int nd);
int resblock_cl_bf16_units(int C, int k);
int launch_resblock_cl_bf16(hipStream_t stream, const RbClLaunch& L);
double resblock_cl_bf16_flops(const RbClLaunch& L);
double resblock_cl_bf16_bytes(const RbClLaunch& L);

// the same whole-ResBlock launch for C = 16 on v_mfma_f32_16x16x32_bf16 (kernels/resblock_c16_bf16.hip, round 5): one MFMA = 16 output
// channels x 16 time steps x (2 taps x 16 input channels), unpadded 32-byte LDS rows, two workgroups per CU.  Same RbClLaunch, but
//   w    : [d][conv e][unit u < rb16_units(k)][lane 64][8 bf16], element index inside a conv = rb16_w_index (zero where tap >= k)
//   bias : fp32 [2*nd][16]
inline int rb16_units(int k) { return (k + 1) / 2; }
inline int64_t rb16_w_index(int j, int ci, int co) {            // tap j, input channel ci, output channel co (all < 16 / k)
  const int q = 2 * (j & 1) + (ci >> 3);
  return ((int64_t)(j >> 1) * 64 + co + 16 * q) * 8 + (ci & 7);
}
bool resblock_c16_bf16_supported(int C, int k, const int* dil, int nd);
int launch_resblock_c16_bf16(hipStream_t stream, const RbClLaunch& L);

// one (dilated conv, conv) pair of ResBlock1 with its residual at C = 64 / 128 / 256 in one launch, bf16 channels-last, the
// intermediate in LDS (kernels/respair_cl_bf16.hip).  x / out: [B][L][C], out != x; w1 / w2: the convs' ordinary fragment streams
// (cl_w_index), b1 / b2 fp32 [C]; conv1 has dilation dil, conv2 dilation 1, both k taps.
struct RpClProb { const uint16_t* x; uint16_t* out; const uint16_t* w1; const uint16_t* w2; const float* b1; const float* b2; int k, dil; };
struct RpClLaunch { RpClProb p[3]; int nprob, B, C, L; float slope; const int64_t* lens = nullptr; int len_mul = 1;
                    int form = 1;    // 1: 64-channel x 128-row wave tiles on the swizzled tile, 0: 32-channel waves on the padded tile
                    int mix = 1;     // 1: the problems interleaved in dispatch order (every CU holds tiles of all branches), 0: problem-major
                    // stage hand-over (round 6): non-null on a stage's LAST pair launch = one workgroup runs the nprob branches of its tile one
                    // after the other (p[0] first) and accumulates their outputs into ONE tensor here (read-modify-write of its own rows,
                    // which stay in L2) — the branch mean with the rounding points of cl_bf16.h stage_mean; p[i].out is not written
                    uint16_t* sum_out = nullptr; int kmax = 0; float sum_scale = 1.f;
                    unsigned long long* dbg = nullptr; };
bool respair_cl_bf16_supported(int C, int k, int dil);
int launch_respair_cl_bf16(hipStream_t stream, const RpClLaunch& L, const char** variant_name);
double respair_cl_bf16_flops(const RpClLaunch& L);
double respair_cl_bf16_bytes(const RpClLaunch& L);

// z[b][c][t] * mask[b][t] (fp32, channel stride z_rstride) -> bf16 channels-last out[b][t][c], t < L
int launch_cast_cl(hipStream_t stream, const float* z, int z_rstride, int64_t z_bstride, const float* mask, int mask_bstride,
                   uint16_t* out, int B, int C, int L);
// debug taps: bf16 channels-last [B][L][C] -> fp32 [B][C][L]
int launch_uncast_cl(hipStream_t stream, const uint16_t* x, float* out, int B, int C, int L);
// conv_post + tanh on channels-last bf16 branches: out[b][t] = tanh( sum_{c,j} w[c][j] * lrelu(in_scale*sum_s x_s[b][t-pad+j][c]) )
struct ConvPostClArgs {
  const uint16_t* x[3]; int nsrc; float in_scale;
  const float* w;           // [C][k] fp32
  float* out;               // [B][L] fp32
  int C, k, L, B; float slope;
  const int64_t* lens = nullptr; int len_mul = 1;
  int generic = 0;          // 1: the any-width kernel also at C = 16, k = 7 ("conv_post_rows" = 0; tests)
};
int launch_conv_post_cl(hipStream_t stream, const ConvPostClArgs& a);

// --------------------------------------------------------------------------------------------------------------
// fp16 convolutions / projections of the attention Encoder stacks (kernels/enc_f16.hip): v_mfma_f32_32x32x16_f16, fp32
// accumulate.  Sits between the fp32 [B][C][T] tensors of the fp32 kernels (LayerNorm, attention) without conversion passes:
//   in_ct  = 1: x is fp32 [B][cin][x_rstride] (T contiguous), multiplied by in_mask[b][t] and rounded to fp16 while staged
//   in_ct  = 0: x is fp16 channels-last [B][Lin][cin]
//   out_ct = 1: out is fp32 [B][cout][out_rstride]:  v = acc + bias; relu; *mask (mask_pre); res op; *mask (mask_post)
//   out_ct = 0: out is fp16 channels-last [B][L][cout]:  v = fp16( relu(acc + bias) * mask )
// w = fp16 fragment stream in cl_w_index order (cin % 16 == 0).
struct HcProb {
  const void* x; int in_ct;
  int64_t x_bstride;        // elements between batches
  int x_rstride;            // in_ct: floats between channels
  int Lin;
  const float* in_mask; int in_mask_bstride;      // in_ct only; null = no mask
  const uint16_t* w;
  const float* bias;        // [cout_pad] fp32 or null
  const float* bias2; int bias2_bstride;           // [B][bias2_bstride] per-batch bias or null (WN conditioning slice g_l)
  void* out; int out_ct;
  int64_t out_bstride;      // elements between batches
  int out_rstride;          // out_ct: floats between channels
  const float* res; int64_t res_bstride; int res_mode;   // out_ct only, indexed like out
  const float* out_mask; int out_mask_bstride;
  int mask_pre, mask_post, act;
  int cin, cout, cout_pad, k, dil, pad_left;
  // out_ct, cout = 192 (conv_f16_ln_supported): out = LayerNorm over the cout channels of the conv's result (after bias /
  // residual / masks), eps ln_eps, scale / shift ln_gamma / ln_beta [cout]; null = no LayerNorm.  out may be the residual's tensor
  // ln_vec [B][ln_vec_bstride] / ln_mask [B][out_mask_bstride] (each may be null): out = (LayerNorm(...) + ln_vec[b][c]) * ln_mask[b][t]
  const float* ln_gamma; const float* ln_beta; float ln_eps;
  const float* ln_vec; int ln_vec_bstride; const float* ln_mask;
  // out_ct, no LayerNorm (round 6, the fused q/k/v projection): output rows [kv_row0, kv_row0 + kv_rows) go to k16 as fp16 channels-last
  // [B][k16_ld][kv_rows] and rows [kv_row0 + kv_rows, kv_row0 + 2 kv_rows) to v16 as fp16 [B][kv_rows][k16_ld] — what attention.hip's KV16 form
  // reads — instead of `out`; kv_row0 and kv_rows are multiples of 32 (a wave's 32-row tile goes one way or the other).  null = everything to `out`
  uint16_t* k16; uint16_t* v16; int kv_row0, kv_rows, k16_ld;
};
bool conv_f16_ln_supported(int cout);
// act == ACT_GATE (out_ct = 0 only; WN, reference commons.py:98-105): the weight rows come in gate order (bv2_model.cpp wn_gate_row:
// rows [0,16) of every 32-row tile = tanh half, rows [16,32) = sigmoid half of the same 16 channels); the epilogue writes
// out[b][t][16*mt + j] = fp16( tanh(v[j]) * sigmoid(v[j+16]) ), `out` has cout/2 channels (out_bstride = cout/2 * L).
// A launch carries 1 or 2 problems (blockIdx.z) with the same cin / k / dil / cout_pad and input form: the two row halves of
// res_skip_layers (x update in place, skip sum), reference modules.py:203-210.
// xcd_b = 1: batch item b runs on XCD b % 8 (a 1-D launch of 8 * ceil(B / 8) * workgroups-per-item, workgroup id -> XCD id & 7 is the
// hardware's round-robin): the kernels of an Encoder layer then hand a batch item's tensors on inside ONE XCD's L2 (the eight L2s are not
// coherent: a tile produced on another XCD comes back through the fabric).  Same switch in LnArgs / AttnArgs; see xcd_decode.
struct HcLaunch { HcProb p[2]; int nprob = 1; int B, L; unsigned long long* dbg = nullptr;   // dbg: tools/timeline.py only
                  int no_ksplit = 0;                 // 1: the FFN conv_2 shape without the in-workgroup K split ("f16_ksplit" = 0)
                  int wn_pref = 0;                   // tuning ("f16_wn"): 4 / 6 / 8 = that many waves per workgroup when it wastes no more wave slots than the default choice
                  int ni_pref = 0;                   // tuning ("f16_ni"): 2 / 4 = 64 / 128 time steps per workgroup
                  int xcd_b = 0; int xcd_gx = 0, xcd_per = 0; };    // xcd_gx / xcd_per: filled by the launcher
int launch_conv_f16(hipStream_t stream, const HcLaunch& L, const char** variant_name);
void conv_f16_set_tuning(int generic);                       // tests / tuning only (bv2_test_set_variants)
bool conv_f16_supported(int cin, int cout, int k, int dil, bool out_cl);
double conv_f16_flops(const HcLaunch& L);
double conv_f16_bytes(const HcLaunch& L);

// --------------------------------------------------------------------------------------------------------------
// conv_post + tanh (kernels/misc.hip): out[b][t] = tanh( sum_{c,j} w[c][j] * lrelu_slope( in_scale*sum_s x_s[b][c][t-pad+j] ) )
struct ConvPostArgs {
  const float* x[3]; int nsrc; float in_scale; int64_t x_bstride; int x_rstride;
  const float* w;           // [C][k] fp32 (unpadded)
  float* out; int64_t out_bstride;
  int C, k, L, B; float slope;
  const int64_t* lens = nullptr; int len_mul = 1;
};
int launch_conv_post(hipStream_t stream, const ConvPostArgs& a);

// --------------------------------------------------------------------------------------------------------------
// channel LayerNorm family (kernels/layernorm.hip)
//   v[c][t]   = sum_{s<nslab} a[s*slab_stride + b][c][t] (+ add[b][c][t])   mode 0 (nslab partial slabs of a split-K conv;
//               with `ml` set the slabs are weighted: the key-split attention's merge, see below)
//             = dwb[c] + sum_j dww[c][j] * a[b][c][t+(j-1)*dil]*in_mask[b][.]   mode 1 (depthwise k=3, DDSConv)
//   y         = (v - mean_c) * rsqrt(var_c + eps) * gamma[c] + beta[c];  y = gelu(y) if post_gelu
//   out       = ((res ? res : 0) + y + (vec ? vec[b][c] : 0)) * (mask ? mask[b][t] : 1)
//   out2      = ((res ? res : 0) + y + (vec ? vec[b][c] : 0) + vec2[b][c]) * mask     (optional second output: the
//               DurationPredictor's `x + cond(g)` rides on the text encoder's last LayerNorm, models.py:288-289)
struct LnArgs {
  const float* a; const float* add;
  int nslab; int64_t slab_stride;     // nslab <= 1: a is a plain tensor
  int mode; const float* dww; const float* dwb; int dil; const float* in_mask;
  const float* gamma; const float* beta; float eps;
  int post_gelu;
  const float* res; const float* vec; int vec_bstride; const float* mask;
  float* out;
  int B, C, T;
  float* out2; const float* vec2; int vec2_bstride;
  // weighted slabs (mode 0, nslab = ml_H * ml_ks in {4, 8}): slab h*ml_ks + r is the key-split attention's partial (head h, key
  // range r) behind conv_o; ml [B][ml_H][ml_ks][2][T] holds the range's softmax (max, sum) per query.  v = sum_{h,r} w_{h,r} slab
  // + bias[c] (conv_o's bias) + add (the residual), w_{h,r} = l_r e^{m_r - M_h} / sum_r' l_r' e^{m_r' - M_h}  (AttnArgs::ksplit)
  const float* ml; int ml_H, ml_ks; const float* bias;
  int xcd_b; int xcd_per;             // batch item -> XCD affinity (HcLaunch::xcd_b); xcd_per: filled by the launcher
  Prefetch pf;                        // B == 1 only: the next launch's weight stream (Prefetch above); ptr null = none
};
int launch_layernorm(hipStream_t stream, const LnArgs& a);

// The boundary between two coupling layers of the transformer flow in one launch (kernels/flow_boundary.hip): LayerNorm-2 of the
// coupling's last Encoder layer (slab sum + LayerNorm + mask), post (1x1, C -> C/2, x1 = (x1 - post(h) - b) * mask, in place in z) and —
// unless pre_w is null (last coupling) — pre of the next coupling (1x1, C/2 -> C, (pre(x1) + b) * mask -> pre_out).  Weights in the
// packed conv layout (conv_w_index, k = 1).  h_out: optional copy of the LayerNorm output (debug taps).
struct FbArgs {
  const float* a; int nslab; int64_t slab_stride;      // LayerNorm input: nslab slabs [B][C][T]
  const float* gamma; const float* beta; float eps;
  const float* mask;                                   // [B][T]
  const float* x1; float* x1_out; int64_t z_bstride;   // [C/2 rows][T] inside z (floats between batch items)
  const float* post_w; const float* post_b;
  const float* pre_w; const float* pre_b; float* pre_out;   // pre_out [B][C][T]
  float* h_out;
  int B, C, T;
  int launched;                                        // set by the executor: the fused launch was taken (else LayerNorm + post + pre)
  int C1;                                              // rows of x1 (= post's outputs = pre's inputs); the kernel is built for C1 == C/2
};
bool flow_boundary_supported(const FbArgs& a);
int launch_flow_boundary(hipStream_t stream, const FbArgs& a);

// --------------------------------------------------------------------------------------------------------------
// BERT feature extractor (kernels/bert.hip; include/bv2_bert.h)
struct BertEmbedArgs {
  const int64_t* input_ids; const int64_t* token_type_ids;     // [B][S]; token_type_ids may be null (all 0)
  const float* word; const float* pos; const float* type;      // [vocab][C], [max_pos][C], [type_vocab][C]; pos / type may be null (DeBERTa-v2)
  const int64_t* lengths;                                      // null, or [B]: the output column of token s >= lengths[b] is multiplied by 0 (DebertaV2Embeddings)
  const float* gamma; const float* beta; float eps;
  float* out;                                                  // [B][C][S]
  int B, S, C, vocab, max_pos, type_vocab;
  Prefetch pf = {nullptr, 0};                                  // B == 1: the first q/k/v projection's weights
};
int launch_bert_embed_ln(hipStream_t stream, const BertEmbedArgs& a);
struct BertLnArgs {
  const float* a; int nslab; int64_t slab_stride;              // sum of nslab slabs [B][C][T]
  const float* gamma; const float* beta; float eps;
  const float* mask;                                           // null, or [B][T] multiplied into the output (DeBERTa-v2 ConvLayer)
  float* out;
  int B, C, T;
  Prefetch pf = {nullptr, 0};                                  // B == 1: the next GEMM's weights
};
int launch_bert_ln(hipStream_t stream, const BertLnArgs& a);
// DeBERTa-v2 disentangled attention (kernels/deberta_attn.hip)
struct DebertaAttnArgs {
  const float* qkv;         // [B][3*H*D][ld]: q rows (ALREADY divided by sqrt(3 D)), k rows, v rows; ld % 32 == 0, ld >= T
  int ld;
  const float* mask;        // [B][T]
  const float* pk;          // [H][D][2 span]   key_proj(LayerNorm(rel_embeddings)), transposed per head
  const float* pq;          // [H][D][2 span]   query_proj(LayerNorm(rel_embeddings)) / sqrt(3 D), transposed per head
  const float* tab;         // [2 P - 1] floats: t(r) = clamp(bucket(r) + span, 0, 2 span - 1) for r = -(P-1) .. P-1
  float* out;               // [B][H*D][T]
  int B, H, D, T, P, span;
};
int launch_deberta_attn(hipStream_t stream, const DebertaAttnArgs& a);

// --------------------------------------------------------------------------------------------------------------
// windowed relative-position multi-head attention (kernels/attention.hip), reference attentions.py:273-322
struct AttnArgs {
  // [B][3*H*D + H*(2W+1)][ld]: q rows [0,HD) ALREADY divided by sqrt(D), k rows [HD,2HD), v rows [2HD,3HD) (head h = rows
  // h*D..h*D+D-1), then per head the 2W+1 relative-key logit rows qe[h][r][i] = (q_i/sqrt(D))·Ek[r] — all produced by
  // ONE fused 1x1 projection (the scale and Ek are folded into its weights at pack time).  ld % 32 == 0, ld >= T.
  const float* qkv;
  int ld;
  const float* mask;        // [B][T]
  const float* erv;         // [2W+1][D]
  float* out;               // [B][H*D][T]
  int B, H, D, T, W;
  int f16;                  // 1: QK^T and PV on the fp16 matrix core (operands rounded in registers, everything else fp32)
  // f16 only (round 6): K and V handed over as fp16 by the q/k/v projection (HcProb::k16 / v16) — kh [B][ld][H*D] channels-last, vh [B][H*D][ld];
  // both or neither.  The k / v rows of `qkv` are then not read (q and the relative-key rows still are).
  const uint16_t* kh = nullptr; const uint16_t* vh = nullptr;
  // fused output projection (fp32 form only; wo == nullptr: plain attention output in `out`): each head's workgroup multiplies its
  // tile by its K-slice of the packed 1x1 weight `wo` (conv_w_index order, wo_groups = cin_pad / 8) and writes partial slab h of
  // o_out [B][Co][T] (slab h at o_out + h * o_slab_stride; head 0 adds bias `bo` and residual `res` [B][Co][T])
  const float* wo = nullptr; const float* bo = nullptr; const float* res = nullptr;
  float* o_out = nullptr; int64_t o_slab_stride = 0; int Co = 0, wo_groups = 0;
  // key split (fused conv_o form only, bo == res == nullptr): the key tiles of a (head, query tile) are dealt to `ksplit` workgroups;
  // workgroup r writes its LOCALLY normalised partial through conv_o into slab h*ksplit + r and (max, sum) of its key range to
  // ml_out [B][H][ksplit][2][T]; the consumer (LnArgs::ml) merges the slabs with the flash-decoding weights
  int ksplit = 1; float* ml_out = nullptr;
  unsigned long long* dbg = nullptr;   // tools/timeline.py only
  int xcd_b = 0; int xcd_gx = 0, xcd_per = 0;   // batch item -> XCD affinity (HcLaunch::xcd_b); xcd_gx / xcd_per: filled by the launcher
};
int attention_pick_ksplit(int B, int H, int T, int max_slabs);   // key ranges per (head, query tile) that fill the chip at small batch
int launch_attention(hipStream_t stream, const AttnArgs& a);
double attention_flops(const AttnArgs& a);

// --------------------------------------------------------------------------------------------------------------
// small kernels (kernels/misc.hip)
struct GemvProb { const float* w; const float* bias; float* out; int cout, cin; int out_bstride; };  // w [cout][cin]
struct GemvLaunch { GemvProb p[16]; int nprob; int B; const float* g; int g_bstride; };
int launch_gemv(hipStream_t stream, const GemvLaunch& L);            // out[b][co] = bias[co] + w[co][:]·g[b][:]

// phase-A front (one launch): the speaker-conditioning GEMVs with g either given or looked up (g[b] = table[sid[b]], also
// written to g_out — emb_g, models.py:1046), x_mask = sequence_mask(x_lengths) (commons.py:119-123; lengths null = all ones)
// and z = noise * noise_scale_w (models.py:248-251).  Every part is optional.
struct FrontArgs {
  GemvProb p[16]; int nprob; int B;
  const float* g; int g_bstride;
  const float* table; const int64_t* sid; int nrows; float* g_out; int gin;
  const int64_t* lengths; float* mask; int T;
  const float* noise; float* z; float noise_scale; int64_t nz;
};
int launch_front(hipStream_t stream, const FrontArgs& a);

int launch_gather_rows(hipStream_t stream, const float* table, const int64_t* idx, float* out, int B, int C, int nrows);
int launch_len_cap(hipStream_t stream, const int64_t* lengths, int64_t* cap, int B);     // cap[b] = max_b lengths[b]
int launch_seq_mask(hipStream_t stream, const int64_t* lengths, float* mask, int B, int T);

// text-encoder front: out[b][c][t] = (emb[x][c] + tone_emb[tone][c] + lang_emb[lang][c] + bsum[b][c][t]) * scale * mask
struct EmbedArgs {
  const int64_t* x; const int64_t* tone; const int64_t* lang;
  const float* emb; const float* tone_emb; const float* lang_emb;
  int n_vocab, n_tones, n_langs;
  const float* bsum; int nslab; int64_t slab_stride;   // bsum = sum of nslab partial slabs of the BERT projections
  const float* mask; float* out; float scale; int B, C, T;
  // word-level features: slab s belongs to feature s / (nslab/3); with idx[f] set, symbol t reads column idx[f][b*T + t] of that
  // feature's slabs (the word2ph repeat of text/chinese_bert.py:48-58 as a gather); null = column t
  const int32_t* idx[3];
  int cols[3];                                         // columns (words) of feature f when idx[f] is set; the gather clamps to them
};
int launch_embed(hipStream_t stream, const EmbedArgs& a);

// out[b][c][t] = (a[b][c][t] + vec[b][c]) * mask[b][t]     (vec / mask may be null)
int launch_flip_channels(hipStream_t stream, float* z, int B, int C, int T);   // z[b][c][t] <-> z[b][C-1-c][t] in place (modules.Flip)
int launch_add_vec_mask(hipStream_t stream, const float* a, const float* vec, int vec_bstride, const float* mask,
                        float* out, int B, int C, int T);

// --- stochastic duration predictor glue (reference models.py:245-256, modules.py:486-516, transforms.py) ---
int launch_scale(hipStream_t stream, const float* in, float* out, float s, int64_t n);
// h[b][c][t] = w[c]*z[b][src][t] + bias[c] + g[b][c][t]       (ConvFlow.pre + the `x + g` of DDSConv)
int launch_convflow_pre(hipStream_t stream, const float* z, int src, const float* w, const float* bias, const float* g,
                        float* h, int B, int C, int T);
// z[b][dst][t] = mask * rq_spline_inverse(z[b][dst][t]; params[b][0:29][t]);  z[b][src][t] *= mask
int launch_spline(hipStream_t stream, float* z, int src, int dst, const float* params, int params_rows,
                  const float* mask, float sqrt_fc, float tail_bound, int B, int T);
// ElementwiseAffine^-1, logw mix, exp/ceil, cumsum, y_lengths (reference modules.py:397-399, models.py:1052-1057)
struct DurArgs {
  const float* z;           // [B][2][T] sdp state; logw_sdp = (z[:,0]-m0)*exp(-logs0)*mask
  const float* ea_m; const float* ea_logs;   // device, [2] each (sdp.flows.0.m / .logs)
  const float* logw_dp;     // [B][T]
  const float* mask;
  float sdp_ratio, one_minus_ratio, length_scale;   // one_minus_ratio = (float)(1.0 - (double)sdp_ratio)
  float* logw_sdp; float* logw; float* w_ceil; int64_t* y_lengths;
  int B, T;
};
int launch_durations(hipStream_t stream, const DurArgs& a);

// --- one whole DDSConv layer per launch (kernels/dds_fused.hip; reference modules.py:121-129) -----------------------
//   xin = x                      or   pre_w[c]*z[b][z_src][t] + pre_b[c] + g[b][c][t]   (ConvFlow.pre + the `x + g`, modules.py:488-489,119-120)
//   y   = gelu(LN1(dwb + dwconv_k3,dil(xin*mask)));  y = W y + bias (1x1, MFMA);  y = gelu(LN2(y));  out = xin + y  [* mask if last_mask]
// optional post 1x1 GEMM on the (masked) layer output, for the layer that ends a DDSConv:
//   post_out != null : post_out[b][r][t] = (Wp out + bp)[r][t] * mask                        (sdp.proj, models.py:203-204)
//   zio      != null : params = (Wp out + bp) * mask (29 rows); z[b][z_dst] = mask * rq_spline_inverse(z[b][z_dst]; params);
//                      z[b][z_src] *= mask                                                   (ConvFlow.proj + spline, modules.py:491-516)
// A workgroup owns 16 time steps x all C channels (C/16 waves); both channel LayerNorms reduce with wave shuffles + one LDS
// exchange; the 1x1 convs run on v_mfma_f32_16x16x4_f32 with weights in the conv_w_index fragment order (k = 1).
// out must not alias x (tiles read their neighbours' columns).  C in {128, 192, 256}.
struct DdsArgs {
  const float* x; const float* pre_w; const float* pre_b; const float* z; int z_src; const float* g;
  const float* mask;
  const float* dww; const float* dwb; const float* g1; const float* b1; const float* g2; const float* b2;
  const float* w; const float* bias;
  float* out;                           // [B][C][T] or null (when only the post GEMM's result is needed)
  int dil, last_mask; float eps;
  const float* post_w; const float* post_b; int post_cout, post_cout_pad;
  float* post_out;
  float* zio; int z_dst; float sqrt_fc, tail;
  float cst, wscale;                    // filled by launch_dds_layer
  int B, C, T;
};
bool dds_fused_supported(int C);
int launch_dds_layer(hipStream_t stream, const DdsArgs& a);

// --- length regulation (reference models.py:1058-1071, commons.py:126-140) ---
struct ExpandArgs {
  const float* w_ceil; const float* x_mask; const int64_t* y_lengths;
  const float* m_p; const float* logs_p;             // [B][C][T]
  const float* noise; int64_t nz_bstride, nz_cstride, nz_tstride; float noise_scale;
  int* frame_idx;                                     // [B][Ty] scratch
  float* attn; float* y_mask; float* z_p; float* m_e; float* logs_e;   // outputs (attn/y_mask/m_e/logs_e may be null)
  float* z_p2;                                        // optional second copy of z_p (the flow updates z_p in place)
  int B, C, T, Ty;
};
int launch_expand(hipStream_t stream, const ExpandArgs& a);

// --- 16-bit PCM of the valid samples, peak-normalised per utterance (gradio convert_to_16_bit_wav, reference webui.py:86) ---
int launch_pcm16(hipStream_t stream, const float* wave, int64_t bstride, const int64_t* y_lengths, int hop, int B, int64_t S,
                 int16_t* pcm, int64_t pstride, unsigned* peak_scratch);

}  // namespace bv2

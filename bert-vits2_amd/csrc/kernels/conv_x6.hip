// conv_x6.hip — the LDS-tiled fp32 conv1d of conv_mfma.hip on the gfx950 BF16 matrix core: fp32 operands, fp32 results, the
// products formed from three-way bf16 splits ("bf16x6" emulation of an fp32 GEMM).
//
// Why: the wide Generator stages are bound by the fp32 MFMA rate (v_mfma_f32_32x32x2_f32: 157 TF, 32 MAC per cycle per SIMD);
// v_mfma_f32_32x32x16_bf16 multiplies 16x as many bf16 pairs per cycle (2.5 PF dense) and accumulates in fp32.  Every fp32 value is
// EXACTLY the sum of three bf16 values, v = h1 + h2 + h3 (8 + 8 + 8 significand bits, same exponent range; round-to-nearest at every
// step), every bf16 x bf16 product is exact in fp32, and of the nine cross terms of w*x the six largest
//      w1x1 + (w1x2 + w2x1) + (w1x3 + w2x2 + w3x1)
// leave out |w2x3 + w3x2 + w3x3| < 2^-23 |wx| — less than the rounding of ONE fp32 multiply-add — so the conv is computed to fp32
// accuracy with 6 bf16 MFMAs (6/16 of the fp32 MFMA time) per 16 channels instead of 8 fp32 MFMAs.
// Envelope (tests/test_x6_gpu.py::test_conv1d_x6_envelope): every FINITE fp32 operand splits exactly — plane 1 saturates at the
// largest bf16 instead of rounding the top 0.2 % of the range to inf — so finite inputs give the fp32 kernel's result up to FLT_MAX;
// +-0 and fp32 denormals behave as in fp32 except that bits below 2^-133 (the bf16 denormal step) of a tiny operand are dropped: an
// ABSOLUTE error <= 2^-134 |other operand| per term; a NaN operand gives NaN; an infinite operand gives a non-finite result that may
// be NaN where fp32 arithmetic gives +-inf (its remainder planes are inf - finite and inf - inf).  tests/test_x6_gpu.py compares
// it against the fp32-MFMA kernel and the oracle; the end-to-end parity numbers (bench.py `parity`) are the same to the last digit
// shown.  (The same technique as the BF16x9 / x6 fp32-emulation modes of vendor BLAS libraries; no reference counterpart — the
// reference runs these convs through MIOpen / cuDNN fp32.)
//
//   GEMM view:  M = C_out (rows),  N = time (columns, contiguous in HBM, fp32 [B][C][T] like every other tensor of the fp32 path),
//   K = (C_in group of 16, tap j).
//
// Workgroup = 4 MFMA waves = WM x WN (+ 2 loader waves in the NLD = 2 form, below), wave tile = MI x NI blocks of 32x32 (shipped: 1 x 2).
//   * weights: pre-split at pack time into three bf16 planes in MFMA-A fragment order (x6_w_index, bv2_kernels.h), streamed
//     global -> registers through a ring with TWO slots per 16-channel group of the chunk (taps j and j + 1; the ring is never
//     drained: every group's stream runs (g, 0) .. (g, k-1), then (g, 0) of the next chunk — a unit is requested 2*GR - 1 units ahead);
//   * X: a chunk of CK input channels x (BN + halo) columns is loaded fp32 from HBM (lane = column: coalesced), pre-activated,
//     split into its three planes ONCE per element and written CHANNELS-LAST to LDS (plane p: [column][CK + 8] bf16, row pitch an
//     odd multiple of 16 B), so that the MFMA B operand — 8 consecutive channels of one column — is one ds_read_b128 and the k taps
//     are row shifts of the same tile.  The next chunk's global loads fly under this chunk's MFMAs; two barriers per chunk (one, and
//     the staging off the MFMA waves altogether, with loader waves).
//   * per unit (16 channels x 1 tap): NI*3 ds_read_b128 + MI*3 global_load_dwordx4 feed MI*NI*6 MFMAs (12 for the 32x64 wave tile).
//
// Round 6, the two-plane fp16 form (NP = 2, "x3"; bv2_kernels.h has the arithmetic): the same kernel on two SCALED fp16 planes per operand and
// three products — w S_w = g0 + g1, x S_x = h0 + h1 (fp16 halves, round-to-nearest), acc += g1 h0 + g0 h1 + g0 h0, result = acc / (S_w S_x) —
// i.e. a third less LDS and ring registers and half the matrix instructions per unit.  S_w comes with the packed planes; S_x from the max |x|
// of the input tensor, which the launch that WROTE that tensor published from its epilogue (ConvProb::omax -> ::xmax: one 128-byte line per
// XCD, L2-local atomics, x3_publish below).  Shipped tiles of the form: 128x64 plain (two-tap ring, three workgroups per CU) for launches of
// > 512 workgroups, 128x64 with FOUR loader waves and a four-tap ring (RD = 4) for the one-workgroup-per-CU launches.
// Tiles and what was measured around them: launch_conv1d_x6 at the end of the file; DESIGN.md 3 / 5.
#include <hip/hip_runtime.h>
#include "../bv2_kernels.h"

namespace bv2 {

namespace {

typedef __bf16 xbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 xbf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 xf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 xf16x2 __attribute__((ext_vector_type(2)));
typedef float xf32x2 __attribute__((ext_vector_type(2)));
typedef float xf32x16 __attribute__((ext_vector_type(16)));
typedef unsigned xu32x4 __attribute__((ext_vector_type(4)));
// explicit global address space for the ring (see gen_bf16.hip: a FLAT load would also count on lgkmcnt)
typedef __attribute__((address_space(1))) xbf16x8 XGlobalFrag;

__device__ __forceinline__ float x6_ld(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ unsigned x6_pack(float a, float b) {     // round-to-nearest-even (v_cvt_pk_bf16_f32)
  xbf16x2 r;
  r[0] = (__bf16)a; r[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float x6_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float x6_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

constexpr float X6_BF16_MAX = 3.38953139e38f;   // 0x7f7f0000
// ConvProb::omax: the wave's max |v| into its XCD's line of the slot (bv2_kernels.h).  |v| >= 0, so fp32 bit patterns order like the
// values.  `seen`: the word as read at the start of the kernel (stale is fine: it only filters redundant atomics).
__device__ __forceinline__ unsigned* x3_slot_word(unsigned* slot) {
  return slot + X3_LINE_WORDS * (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u);      // XCC_ID
}
__device__ __forceinline__ void x3_publish(unsigned* word, unsigned seen, float vmx, int lane) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) vmx = fmaxf(vmx, __shfl_xor(vmx, d));
  const unsigned bits = __float_as_uint(vmx);
  if (lane == 0 && bits > seen) __hip_atomic_fetch_max(word, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ xf32x16 x6_mfma(xbf16x8 a, xbf16x8 b, xf32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ xf32x16 x6_mfma(xf16x8 a, xf16x8 b, xf32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
// NP = 2 (the x3 form, bv2_kernels.h): fp16 halves of the SCALED value, round-to-nearest-even (v_cvt_f16_f32 x 2 + pack)
__device__ __forceinline__ unsigned x3_pack(float a, float b) {
  xf32x2 v = {a, b};
  const xf16x2 r = __builtin_convertvector(v, xf16x2);
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ xf32x2 x3_unpack(unsigned u) { return __builtin_convertvector(__builtin_bit_cast(xf16x2, u), xf32x2); }

}  // namespace

// NLD = 2: "loader waves".  Two extra waves per workgroup do nothing but the X staging — global loads of chunk c+2, split of chunk
// c+1 into the OTHER of two LDS buffers — while the four MFMA waves run chunk c; one barrier per chunk.  In the NLD = 0 form the MFMA
// waves stage themselves between two barriers with the matrix pipe idle: 2.5-4k cycles per chunk (tools/timeline.py), which the other
// resident workgroups hide only partly and, in the one-workgroup-per-CU launches of the C = 256 stage, not at all.  Costs a second
// 31 KB buffer (two workgroups per CU instead of three; with the loaders still 12 waves per CU).
// NP = 3: three bf16 planes per operand, six products.  NP = 2: two scaled fp16 planes, three products (the "x3" form, bv2_kernels.h):
// the same kernel with a third less LDS / ring registers and half the matrix instructions per unit.
// RD: taps per group in the weight ring.  2: the ring described above.  4 (x3 form, every problem's k = 3 mod 4, i.e. the 3 / 7 / 11 ResBlock
// kernels): with three products a unit is 192 matrix cycles and two taps ahead is 576 cycles — less than an L2 round trip (PMC: the C = 256
// launches' waves waited on s_waitcnt 43 % of their life); four taps ahead restores the distance in TIME.  The slot of tap j of chunk c is
// (c k + j) mod 4, so a chunk starts at phase 0, 3, 2, 1, 0 ... and the chunk loop unrolls by four.
template <int WM, int WN, int MI, int NI, int CK, int XR, int NLD, int NP = 3, int RD = 2>
__global__ void __launch_bounds__(64 * (WM * WN + NLD), WM * WN > 4 ? 2 : ((NLD > 0 || (MI * NI <= 2 && CK == 32 && XR <= 128)) ? 3 : 2))
conv1d_x6_kernel(const ConvLaunch L, const int mtiles, const int per_xcd, const int snake_n) {
  static_assert(RD == 2 || RD == 4, "ring depth");
  constexpr int X6_UNIT = NP * 512;                // elements of one (group, tap) unit of one m-tile: NP planes x 64 lanes x 8
  constexpr int BM = WM * MI * 32;
  constexpr int BN = WN * NI * 32;
  constexpr int PITCH = CK + 8;                    // bf16 elements per LDS row: (CK/8 + 1) * 16 B, an odd multiple of 16 B
  constexpr int PLANE = XR * PITCH;                // elements per plane
  constexpr int GR = CK / 16;                      // 16-channel groups per chunk = ring slots
  constexpr int NRG = XR / 64;                     // 64-column groups of the staged tile
  constexpr int NCW = WM * WN;                     // MFMA waves: 4, or 8 (with loader waves only)
  constexpr int STW = NLD > 0 ? NLD : 4;           // waves that stage X
  constexpr int OPW = CK / 8 / STW;                // channel octets per staging wave per column group
  constexpr int BUFSZ = NP * PLANE;                // elements of one X buffer (NLD > 0: two of them)
  static_assert(NCW == 4 || (NCW == 8 && NLD > 0), "4 MFMA waves, or 8 staged by loader waves");
  static_assert((CK / 8) % STW == 0, "octets dealt evenly");
  static_assert(XR % 64 == 0 && XR >= BN && (CK == 32 || CK == 64), "staged tile");
  extern __shared__ __attribute__((aligned(16))) unsigned short xs[];   // [NP][XR][PITCH]

  // placement: identical to conv1d_mfma_kernel (grid (time tiles, m-tiles x batch, problems), or the cost-balanced snake)
  int pz = blockIdx.z, by = blockIdx.y, bx = blockIdx.x;
  if (snake_n > 0) {
    const int xcd = blockIdx.x & 7, pp = blockIdx.x >> 3;
    if (pp >= snake_n) return;
    const int r = pp >> 5, cc = pp & 31;
    const int rem = snake_n - (r << 5) < 32 ? snake_n - (r << 5) : 32;
    const int sidx = (r << 5) + ((r & 1) ? rem - 1 - cc : cc);
    const int per_prob = per_xcd * mtiles * L.B;
    pz = sidx / per_prob;
    const int rest = sidx - pz * per_prob;
    by = rest / per_xcd;
    bx = ((rest - by * per_xcd) << 3) | xcd;
  }
  const ConvProb P = L.p[pz];                     // BY VALUE: one kernarg round trip (see conv_mfma.hip)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = (wid % NCW) / WN, wn = (wid % NCW) % WN;
  const bool loader = NLD > 0 && wid >= NCW;
  const int sw = NLD > 0 ? wid - NCW : wid;                          // index among the staging waves (MFMA waves of NLD > 0: unused)
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, tsw = 0;           // timeline stamps (tools/timeline.py; L.dbg is null in the product)
  if (L.dbg) ts0 = __builtin_amdgcn_s_memtime();
  const int b = by / mtiles;
  const int m0 = (by - b * mtiles) * BM;
  const int vt = per_xcd ? (bx & 7) * per_xcd + (bx >> 3) : bx;
  const int t0 = vt * BN;
  if (t0 >= L.L) return;
  if (m0 >= P.cout_pad) return;

  const int k = P.k, dil = P.dil;
  int Lin = P.Lin;
  if (L.lens) {
    const int64_t lv = L.lens[b] * L.len_mul;
    Lin = lv < Lin ? (int)lv : Lin;
    if (t0 >= Lin) return;
  }
  float in_scale = P.in_scale;
  const float slope = P.slope;
  float acc_scale = 1.f;                           // NP = 2: 1 / (S_w S_x), applied to the accumulators in the epilogue
  if constexpr (NP == 2) {
    unsigned mb = 0;                               // S_x from the input tensor's max |x|: the largest of the slot's eight words (scalar loads)
#pragma unroll
    for (int i = 0; i < 8; ++i) mb = P.xmax[X3_LINE_WORDS * i] > mb ? P.xmax[X3_LINE_WORDS * i] : mb;
    const unsigned ex = x3_scale_exp(mb);
    in_scale *= x3_scale(ex);
    acc_scale = x3_scale_inv(ex) * *P.w3inv;
  }
  // ConvProb::omax: this wave's word of the slot, read NOW (its round trip lands under the prologue's other loads)
  unsigned* const omax_w = P.omax ? x3_slot_word(P.omax) : nullptr;
  unsigned omax_seen = 0xffffffffu;
  if (omax_w) omax_seen = *reinterpret_cast<volatile unsigned*>(omax_w);
  const bool lrelu = P.pre_act == PRE_LRELU;
  const float* const x0p = P.x[0] + (int64_t)b * P.x_bstride;
  const float* const maskp = P.in_mask ? P.in_mask + (int64_t)b * P.in_mask_bstride : nullptr;
  const unsigned x_rs4 = 4u * (unsigned)P.x_rstride;
  const int XW = BN + (k - 1) * dil;
  const int nchunks = P.cin / CK;
  const int groups = P.cin / 16;

  xf32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // ---- weight ring: unit (group s, tap j) of m-tile mt starts at element ((mt*groups + s)*k + j) * X6_UNIT; plane p 512 elements in.
  // One pointer per (group of the chunk, m-tile) walks that group's stream: (g, 0), (g, 1) ... (g, k-1), then (g, 0) of the next chunk.
  // The ring holds TWO taps per group — slot [g][(j + par) & 1] for tap j, par = parity of the taps consumed before this chunk — so a
  // unit is requested 2*GR - 1 units (>= 1150 MFMA cycles) before its first MFMA: with one slot per group (the fp32 kernel's ring, one
  // unit = 384 cycles ahead here) a workgroup alone on its CU (C = 256 at batch 1: 288 workgroups) spent 980 cycles per 384-cycle unit.
  typedef typename std::conditional<NP == 2, xf16x8, xbf16x8>::type frag_t;
  typedef __attribute__((address_space(1))) frag_t GlobalFragT;
  frag_t ar[GR][RD][MI][NP];
  const uint16_t* wq[GR][MI];
  const unsigned wlane = 16u * (unsigned)lane;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    int mt = (m0 >> 5) + wm * MI + mi;
    mt = mt * 32 < P.cout_pad ? mt : (m0 >> 5);   // rows beyond the problem: any valid tile (results are dropped)
#pragma unroll
    for (int g = 0; g < GR; ++g) wq[g][mi] = (NP == 2 ? P.w3 : P.w6) + ((int64_t)mt * groups + g) * k * X6_UNIT;
  }
  const int last_step = ((GR - 1) * k + 1) * X6_UNIT;             // elements from (g, k-1) to (g, 0) of the next chunk
  const int wrap_step = -(((nchunks - 1) * GR * k + (k - 1)) * X6_UNIT);   // from (g, k-1) of the last chunk back to (g, 0) of the first
  // pointer step after loading tap jl of chunk cl (cl < nchunks): the stream of a group never ends — past the last chunk it wraps to
  // the first one (valid memory, values unused) instead of branching around the loads
  auto step_after = [&](int jl, int cl) __attribute__((always_inline)) {
    return jl + 1 < k ? X6_UNIT : (cl + 1 < nchunks ? last_step : wrap_step);
  };
  auto load_unit = [&](int g, int SL, int step) __attribute__((always_inline)) {   // g, SL: literals at every (inlined) call site
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int p = 0; p < NP; ++p)
        ar[g][SL][mi][p] = *(const GlobalFragT*)(reinterpret_cast<const char*>(wq[g][mi]) + wlane + 1024u * (unsigned)p);
      wq[g][mi] += step;
    }
  };

  // ---- X prefetch: wave `wid` loads the channel octets {wid, wid + 4, ...} of every 64-column group (lane = column)
  float xr[NRG][OPW][8];
  float xm[NRG];
  const int tbase = t0 - P.pad_left;
  unsigned tc[NRG];
  float colsc[NRG];
#pragma unroll
  for (int rg = 0; rg < NRG; ++rg) {
    const int r = rg * 64 + lane;
    const int t = tbase + r;
    const bool tok = r < XW && t >= 0 && t < Lin;
    colsc[rg] = tok ? in_scale : 0.f;
    tc[rg] = 4u * (unsigned)(t < 0 ? 0 : (t >= Lin ? Lin - 1 : t));
  }
  auto issue_x = [&](int c) __attribute__((always_inline)) {
    if (maskp) {
#pragma unroll
      for (int rg = 0; rg < NRG; ++rg) xm[rg] = x6_ld(maskp, tc[rg]);
    } else {
#pragma unroll
      for (int rg = 0; rg < NRG; ++rg) xm[rg] = 1.f;
    }
#pragma unroll
    for (int o = 0; o < OPW; ++o) {
      const unsigned row0 = (unsigned)(c * CK + (sw + STW * o) * 8) * x_rs4;
#pragma unroll
      for (int rg = 0; rg < NRG; ++rg)
#pragma unroll
        for (int e = 0; e < 8; ++e) xr[rg][o][e] = x6_ld(x0p, row0 + (unsigned)e * x_rs4 + tc[rg]);
    }
  };
  auto store_x = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int rg = 0; rg < NRG; ++rg) {
      const float sc = colsc[rg] * xm[rg];
#pragma unroll
      for (int o = 0; o < OPW; ++o) {
        xu32x4 p1, p2, p3;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          float a = xr[rg][o][2 * w], bq = xr[rg][o][2 * w + 1];
          const float an = a * slope, bn = bq * slope;
          a = (lrelu && a < 0.f) ? an : a;
          bq = (lrelu && bq < 0.f) ? bn : bq;
          a *= sc; bq *= sc;
          if constexpr (NP == 2) {
            // |a| < 2^15 (sc carries S_x): g0 = fp16(a), g1 = fp16(a - g0); the remainder is exact in fp32
            const unsigned u1 = x3_pack(a, bq);
            const xf32x2 f1 = x3_unpack(u1);
            p1[w] = u1; p2[w] = x3_pack(a - f1[0], bq - f1[1]);
            continue;
          }
          // plane 1 saturates at the largest bf16 (x6_split, bv2_kernels.h): a finite value never rounds to +-inf, its remainder
          // a - h1 (< 2^120) is exact in planes 2 and 3; inf / NaN leave the clamp finite but their remainders are inf / NaN
          const unsigned u1 = x6_pack(__builtin_amdgcn_fmed3f(a, -X6_BF16_MAX, X6_BF16_MAX), __builtin_amdgcn_fmed3f(bq, -X6_BF16_MAX, X6_BF16_MAX));
          a -= x6_lo(u1); bq -= x6_hi(u1);
          const unsigned u2 = x6_pack(a, bq);
          a -= x6_lo(u2); bq -= x6_hi(u2);
          p1[w] = u1; p2[w] = u2; p3[w] = x6_pack(a, bq);
        }
        unsigned short* dst = xs + buf * BUFSZ + (rg * 64 + lane) * PITCH + (sw + STW * o) * 8;
        *reinterpret_cast<xu32x4*>(dst) = p1;
        *reinterpret_cast<xu32x4*>(dst + PLANE) = p2;
        if constexpr (NP == 3) *reinterpret_cast<xu32x4*>(dst + 2 * PLANE) = p3;
      }
    }
  };

  if constexpr (NLD > 0) {
    if (loader) {
      // chunk c + 1 is split into the buffer the MFMA waves left at the previous barrier while they run chunk c; chunk c + 2's loads fly
      // over the barrier and the whole of chunk c + 1
      issue_x(0);
      store_x(0);
      if (nchunks > 1) issue_x(1);
      __syncthreads();
      for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) {
          store_x((c + 1) & 1);
          if (c + 2 < nchunks) issue_x(c + 2);
        }
        __syncthreads();
      }
      return;
    }
  }
  // prologue: X chunk 0 first (the long latency), then prime the ring with the first two units of every group's stream
  if constexpr (NLD == 0) issue_x(0);
  if constexpr (RD == 2) {
    const int s0 = step_after(0, 0);
    const int j1 = k > 1 ? 1 : 0, c1 = k > 1 ? 0 : (nchunks > 1 ? 1 : 0);
    const int s1 = step_after(j1, c1);
#pragma unroll
    for (int g = 0; g < GR; ++g) { load_unit(g, 0, s0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int g = 0; g < GR; ++g) { load_unit(g, 1, s1); __builtin_amdgcn_sched_barrier(0); }
  } else {
    // the first four units of every group's stream (k >= 3: the fourth is tap 0 of chunk 1 when k = 3), slot n for unit n
    int jl = 0, cl = 0;
#pragma unroll
    for (int n = 0; n < RD; ++n) {
      const int st = step_after(jl, cl < nchunks ? cl : 0);
#pragma unroll
      for (int g = 0; g < GR; ++g) { load_unit(g, n, st); __builtin_amdgcn_sched_barrier(0); }
      if (++jl == k) { jl = 0; if (++cl >= nchunks) cl = 0; }
    }
  }
  if constexpr (NLD == 0) store_x(0);
  omax_seen = __builtin_amdgcn_readfirstlane(omax_seen);          // into an SGPR here, where the wave waits for chunk 0 anyway
  __syncthreads();
  if (L.dbg) ts1 = __builtin_amdgcn_s_memtime();

  const unsigned short* const xlane = xs + (wn * (NI * 32) + l31) * PITCH + lh * 8;
  const int tap_step = dil * PITCH;
  // k is odd (conv_x6_supported): chunk c starts at ring parity c & 1.  The control flow is loops and straight-line code only — a
  // conditional tap inside the loop made every ring register a phi (copies + spills at the merge).
  auto chunk = [&](int c, int PAR) __attribute__((always_inline)) {           // PAR: a literal at every (inlined) call site
    const bool next_chunk = (c + 1) < nchunks;
    if constexpr (NLD == 0) {
      if (next_chunk) issue_x(c + 1);             // in flight under this chunk's MFMAs
    }
    frag_t bb[2][NI][NP];
    const unsigned short* xrow = xlane + (NLD > 0 ? (c & 1) * BUFSZ : 0);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int p = 0; p < NP; ++p) bb[0][ni][p] = *reinterpret_cast<const frag_t*>(xrow + ni * 32 * PITCH + p * PLANE);
    // one tap: the GR units (g, j) from ring slots [g][SL]; each slot is refilled with the unit two taps further down its stream
    auto tap = [&](int j, int SL) __attribute__((always_inline)) {               // SL: a literal at every (inlined) call site
      const unsigned short* xnext = (j + 1 < k) ? xrow + tap_step : xrow;   // the chunk's last unit re-reads itself (unused)
      int jl = j + RD, cl = c;                    // the unit loaded during this tap: tap jl of chunk cl (branch-free wrap; RD = 2, k = 1 / RD = 4, k = 3: twice)
      if (jl >= k) { jl -= k; ++cl; }
      if (jl >= k) { jl -= k; ++cl; }
      if (cl >= nchunks) cl -= nchunks;
      if (cl >= nchunks) cl -= nchunks;
      const int step = step_after(jl, cl);
#pragma unroll
      for (int g = 0; g < GR; ++g) {
        {
          const unsigned short* xn = (g + 1 < GR) ? xrow + (g + 1) * 16 : xnext;
          // in the order the next unit's products use them (plane-major: X6_PROD(.., 0) on both column blocks first): the wait in front
          // of an MFMA then covers the OLDEST outstanding read instead of one issued two reads later
#pragma unroll
          for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              bb[(g & 1) ^ 1][ni][p] = *reinterpret_cast<const frag_t*>(xn + ni * 32 * PITCH + p * PLANE);
        }
        // the six products, smallest first; consecutive MFMAs go to different accumulators
#define X6_PROD(WP, XP)                                                                                              \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                            \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                            \
          acc[mi][ni] = x6_mfma(ar[g][SL][mi][WP], bb[g & 1][ni][XP], acc[mi][ni]);
        if constexpr (NP == 3) { X6_PROD(2, 0) X6_PROD(1, 1) X6_PROD(0, 2) }
        X6_PROD(1, 0) X6_PROD(0, 1) X6_PROD(0, 0)
#undef X6_PROD
        load_unit(g, SL, step);
        // pin the emitted order: next unit's LDS reads first (they land under this unit's MFMAs), MFMAs, ring loads
        // emitted order: one LDS read (next unit's B) or one ring load behind every MFMA, so that they issue in the shadow of the
        // 32-cycle matrix op instead of in front of / behind the block of 12 (a lone wave per SIMD ran 530 ticks per 384-cycle unit)
        constexpr int NM = MI * NI * (NP == 3 ? 6 : 3), NDS = NI * NP, NVM = MI * NP;
        static_assert(NM >= NDS + NVM, "one memory instruction per MFMA at most");
#pragma unroll
        for (int q = 0; q < NDS; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NM - NDS - NVM, 0);
#pragma unroll
        for (int q = 0; q < NVM; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);        // keep program order: the ring's vmcnt distances stay 2*GR - 1 units
      }
      xrow = xnext;
    };
    if constexpr (RD == 2) {
      for (int j = 0; j + 1 < k; j += 2) { tap(j, PAR); tap(j + 1, PAR ^ 1); }
      tap(k - 1, PAR);
    } else {                                      // k = 3 mod 4 (launcher): whole rounds of four slots, then three taps
      int j = 0;
      for (; j + 3 < k; j += 4) { tap(j, PAR); tap(j + 1, (PAR + 1) & 3); tap(j + 2, (PAR + 2) & 3); tap(j + 3, (PAR + 3) & 3); }
      tap(j, PAR); tap(j + 1, (PAR + 1) & 3); tap(j + 2, (PAR + 2) & 3);
    }
    if constexpr (NLD > 0) {
      unsigned long long ta = 0;
      if (L.dbg) ta = __builtin_amdgcn_s_memtime();
      __syncthreads();                            // the loaders have the next buffer ready; this one is free for chunk c + 2
      if (L.dbg) tsw += __builtin_amdgcn_s_memtime() - ta;
    } else if (next_chunk) {
      unsigned long long ta = 0;
      if (L.dbg) ta = __builtin_amdgcn_s_memtime();
      __syncthreads();                            // every wave is done reading this chunk's tile
      store_x(0);
      __syncthreads();
      if (L.dbg) tsw += __builtin_amdgcn_s_memtime() - ta;
    }
  };
  int c = 0;
  if constexpr (RD == 2) {
    for (; c + 1 < nchunks; c += 2) { chunk(c, 0); chunk(c + 1, 1); }
    if (c < nchunks) chunk(c, 0);
  } else {                                        // a chunk advances the phase by k = 3 (mod 4)
    for (; c + 3 < nchunks; c += 4) { chunk(c, 0); chunk(c + 1, 3); chunk(c + 2, 2); chunk(c + 3, 1); }
    if (c < nchunks) chunk(c, 0);
    if (c + 1 < nchunks) chunk(c + 1, 3);
    if (c + 2 < nchunks) chunk(c + 2, 2);
  }
  if (L.dbg) ts2 = __builtin_amdgcn_s_memtime();

  // ---- epilogue: v = relu?(acc + bias + bias2) * mask_pre, (+ res | res - v), * mask_post.  Branch-free and in ONE memory round trip:
  // every operand of the wave's MI x NI tiles (2 x 16 bias values per row block, 16 residual values per tile, the column masks) is
  // loaded unconditionally up front — a missing operand reads a valid dummy address and is masked to +0.0 bit-wise — then the
  // arithmetic runs on selects and the stores are predicated.  (The first form kept conv1d_mfma_kernel's runtime `act` / `res_mode`
  // branches around every element, the erf-GELU of the BERT path included: ~10 scalar branches per row, the loads of the second tile
  // behind the stores of the first — tools/timeline.py: 10-20k cycles of a 30-125k-cycle workgroup.)
  {
    const int cout = P.cout, Lout = L.L;
    const unsigned o_rs = (unsigned)P.out_rstride, o_ts = (unsigned)P.out_tstride, o_to = (unsigned)P.out_toff;
    float* const outb = P.out + (int64_t)b * P.out_bstride;
    const float* const dummy = reinterpret_cast<const float*>(NP == 2 ? P.w3 : P.w6);
    const bool has_res = P.res_mode != RES_NONE;
    const float* const resb = has_res ? P.res + (int64_t)b * P.res_bstride : dummy;
    const float* const b1p = P.bias ? P.bias : dummy;
    const float* const b2p = P.bias2 ? P.bias2 + (int64_t)b * P.bias2_bstride : dummy;
    const float* const omaskp = P.out_mask ? P.out_mask + (int64_t)b * P.out_mask_bstride : dummy;
    const unsigned m_b1 = P.bias ? 0xffffffffu : 0u, m_b2 = P.bias2 ? 0xffffffffu : 0u, m_res = has_res ? 0xffffffffu : 0u;
    const unsigned m_om = P.out_mask ? 0xffffffffu : 0u;
    const bool relu = P.act == ACT_RELU;          // ACT_GATE / ACT_GELU never get here (conv_x6_supported admits NONE and RELU only)
    const bool rsub = P.res_mode == RES_RSUB, mpre = P.mask_pre != 0, mpost = P.mask_post != 0;
    float bs[MI][16], bs2[MI][16], rv[MI][NI][16], omr[NI];
    unsigned off0[MI][NI];
    bool colok[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = t0 + wn * (NI * 32) + ni * 32 + l31;
      colok[ni] = col < Lout;
      const int colc = colok[ni] ? col : Lout - 1;
      omr[ni] = x6_ld(omaskp, m_om ? 4u * (unsigned)colc : 0u);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row0 = m0 + wm * (MI * 32) + mi * 32 + 4 * lh;
        off0[mi][ni] = (unsigned)row0 * o_rs + (unsigned)colc * o_ts + o_to;
      }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int row0 = m0 + wm * (MI * 32) + mi * 32 + 4 * lh;       // this lane's rows: row0 + (r & 3) + 8 * (r >> 2)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        const int row = row0 + dr < cout ? row0 + dr : cout - 1;       // clamped: the load is unconditional, the store is not
        bs[mi][r] = x6_ld(b1p, m_b1 ? 4u * (unsigned)row : 0u);
        bs2[mi][r] = x6_ld(b2p, m_b2 ? 4u * (unsigned)row : 0u);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const unsigned off = row0 + dr < cout ? off0[mi][ni] + (unsigned)dr * o_rs : off0[mi][ni];
          rv[mi][ni][r] = x6_ld(resb, m_res ? 4u * off : 0u);
        }
      }
    }
    float vmx = 0.f;                                // max |v| over what this lane stores (ConvProb::omax)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int row0 = m0 + wm * (MI * 32) + mi * 32 + 4 * lh;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const float om = m_om ? omr[ni] : 1.f;
        const float fpre = mpre ? om : 1.f, fpost = mpost ? om : 1.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          const float bsum = __uint_as_float(__float_as_uint(bs[mi][r]) & m_b1) + __uint_as_float(__float_as_uint(bs2[mi][r]) & m_b2);
          const float rr = __uint_as_float(__float_as_uint(rv[mi][ni][r]) & m_res);
          const float pre = NP == 2 ? __builtin_fmaf(acc[mi][ni][r], acc_scale, bsum) : acc[mi][ni][r] + bsum;
          float v = ((relu && pre < 0.f) ? 0.f : pre) * fpre;        // a select, not v_max: a NaN accumulator stays NaN (as in conv_mfma.hip / torch.relu)
          v = rsub ? rr - v : v + rr;
          v *= fpost;
          if (colok[ni] && row0 + dr < cout) {
            outb[off0[mi][ni] + (unsigned)dr * o_rs] = v;
            vmx = fmaxf(vmx, fabsf(v));
          }
        }
      }
    }
    if (omax_w) x3_publish(omax_w, omax_seen, vmx, lane);
  }
  if (L.dbg && tid == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long* d = L.dbg + 8ull * (snake_n > 0 ? (unsigned long long)blockIdx.x
                                                        : ((unsigned long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
    d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = __builtin_amdgcn_s_memtime();
    d[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);               // HW_ID
    d[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);              // XCC_ID
    d[6] = (unsigned long long)k | (tsw << 16);                     // taps | ticks spent in the chunk switches (barrier, split, barrier)
    d[7] = 1;
  }
}

// max |x| of a tensor into a zeroed slot: for x3 inputs whose producer is not a conv launch (tests; the product's producers publish
// from their epilogues)
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, const int64_t n, unsigned* slot) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
  x3_publish(x3_slot_word(slot), 0u, m, threadIdx.x & 63);
}
// The slots of one decode are zeroed by a LAUNCH, i.e. by ordinary stores under the ordinary kernel-to-kernel visibility rules: inside a
// replayed hipGraph a memset node's zeros were not seen by an XCD whose L2 still held the slot's line from the previous replay (its
// producers read the old maximum, found nothing larger to publish, and the consumer scaled by a maximum of 0: NaN on the second replay).
__global__ void __launch_bounds__(256) x3_zero_slots_kernel(unsigned* slots, const int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) slots[i] = 0u;
}
int launch_x3_zero_slots(hipStream_t stream, unsigned* slots, int n_slots) {
  if (!slots || n_slots < 1) return -1;
  const int n = n_slots * X3_SLOT_WORDS;
  hipLaunchKernelGGL(x3_zero_slots_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, slots, n);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int launch_absmax(hipStream_t stream, const float* x, int64_t n, unsigned* slot) {
  if (!x || !slot || n < 1) return -1;
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, stream, x, n, slot);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------------------------------------------
// the two-plane fp16 form: every problem carries its planes, their scale and the slot with its input's max |x|
static bool conv_x3_ready(const ConvLaunch& L) {
  for (int i = 0; i < L.nprob; ++i)
    if (!L.p[i].w3 || !L.p[i].w3inv || !L.p[i].xmax) return false;
  return true;
}

bool conv_x6_supported(const ConvLaunch& L) {
  if (L.nprob < 1 || L.ksplit > 1) return false;
  for (int i = 0; i < L.nprob; ++i) {
    const ConvProb& p = L.p[i];
    if (!p.w6 || p.nsrc != 1 || p.cin % 32 || p.cin != p.cin_pad || p.cout_pad % 32 || p.k < 1 || p.dil < 1) return false;
    if ((p.k - 1) * p.dil > 64 || (p.act != ACT_NONE && p.act != ACT_RELU) || p.k % 2 == 0) return false;   // odd k: the ring's tap parity alternates per chunk
  }
  return true;
}

// tuning experiments (tools/tune_x6.py through bv2_test_set_x6_tuning): forced tile per C_out class and chunk size; 0 = shipped choice
static int g_x6_tile[3] = {0, 0, 0};
static int g_x3_ring = 0;                         // conv_x6_set_tuning's fourth argument (A/B): 2 = the x3 form on the two-tap ring, 3 = four-tap ring with TWO loader waves
void conv_x6_set_tuning(int t256, int t128, int t64, int ring) { g_x6_tile[0] = t256; g_x6_tile[1] = t128; g_x6_tile[2] = t64; g_x3_ring = ring; }

template <int WM, int WN, int MI, int NI, int CK, int XR, int NLD = 0, int NP = 3, int RD = 2>
static int launch_x6_variant(hipStream_t stream, const ConvLaunch& L0, int max_cout_pad) {
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
  const int mtiles = (max_cout_pad + BM - 1) / BM;
  const int ntx = (L0.L + BN - 1) / BN;
  const int per_xcd = (ntx % 8 == 0 || ntx >= 64) ? (ntx + 7) / 8 : 0;      // contiguous per-XCD ranges only if they balance
  dim3 grid(per_xcd ? per_xcd * 8 : ntx, mtiles * L0.B, L0.nprob);
  ConvLaunch Ls = L0;
  int snake_n = 0;
  if (per_xcd && L0.nprob > 1) {                  // cost-balanced placement, see conv_mfma.hip launch_variant
    const long total = (long)per_xcd * 8 * mtiles * L0.B * L0.nprob;
    bool differ = false;
    for (int i = 1; i < L0.nprob; ++i)
      if (L0.p[i].k * L0.p[i].cin_pad != L0.p[0].k * L0.p[0].cin_pad) differ = true;
    if (differ && total > 256 && total <= 512) {
      for (int i = 0; i < Ls.nprob; ++i)
        for (int j = i; j > 0 && Ls.p[j].k * Ls.p[j].cin_pad > Ls.p[j - 1].k * Ls.p[j - 1].cin_pad; --j) {
          const ConvProb t = Ls.p[j]; Ls.p[j] = Ls.p[j - 1]; Ls.p[j - 1] = t;
        }
      snake_n = per_xcd * mtiles * L0.B * L0.nprob;
      grid = dim3(snake_n * 8, 1, 1);
    }
  }
  if (Ls.dbg == nullptr) {                        // tools/timeline.py: a slice of the stamp buffer while one is set
    int ks = 0;
    for (int i = 0; i < Ls.nprob && i < 3; ++i) ks |= (Ls.p[i].k & 255) << (8 * i);
    Ls.dbg = timeline_slice(grid.x, grid.y, grid.z, 6000000 + BM * 1000 + BN, ks, Ls.p[0].cin, Ls.L);
  }
  const size_t lds = (size_t)(NLD > 0 ? 2 : 1) * NP * XR * (CK + 8) * 2;
  auto kern = conv1d_x6_kernel<WM, WN, MI, NI, CK, XR, NLD, NP, RD>;
  ensure_dyn_lds((const void*)kern, lds);
  hipLaunchKernelGGL(kern, grid, dim3(64 * (WM * WN + NLD)), lds, stream, Ls, mtiles, per_xcd, snake_n);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// workgroups per CU the runtime grants each shipped variant (tools/x6_occupancy.py): {128x64, 128x64 with loaders, 64x128, 32x256}.
// Measured 3 / 2 / 2 / 2.  (tools/timeline.py: the 6-wave loader workgroups actually share a CU only 1.14-fold at 145 registers — a
// second one fits only if its waves land on complementary SIMDs; a 117-register build that always fits ran two at once and each twice
// as long: ONE loader workgroup already keeps the CU's matrix pipes ~70 % busy inside its main loop.)
void conv_x6_occupancy(int out[4]) {
  auto q = [](auto kern, int threads, size_t lds) {
    int n = -1;
    ensure_dyn_lds((const void*)kern, lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, threads, lds) != hipSuccess) n = -1;
    return n;
  };
  out[0] = q(conv1d_x6_kernel<4, 1, 1, 2, 32, 128, 0>, 256, (size_t)3 * 128 * 40 * 2);
  out[1] = q(conv1d_x6_kernel<4, 1, 1, 2, 32, 128, 2>, 384, (size_t)2 * 3 * 128 * 40 * 2);
  out[2] = q(conv1d_x6_kernel<2, 2, 1, 2, 32, 192, 0>, 256, (size_t)3 * 192 * 40 * 2);
  out[3] = q(conv1d_x6_kernel<1, 4, 1, 2, 32, 320, 0>, 256, (size_t)3 * 320 * 40 * 2);
}

int launch_conv1d_x6(hipStream_t stream, const ConvLaunch& L, int tile, const char** variant_name) {
  if (!conv_x6_supported(L)) return -2;
  int max_cout_pad = 0;
  for (int i = 0; i < L.nprob; ++i)
    if (L.p[i].cout_pad > max_cout_pad) max_cout_pad = L.p[i].cout_pad;
  if (tile == TILE_X3) tile = TILE_X6;
  if (tile == TILE_X6) {
    const int cls = max_cout_pad % 256 == 0 ? 0 : (max_cout_pad % 128 == 0 ? 1 : 2);
    if (g_x6_tile[cls]) tile = g_x6_tile[cls];
    else if (max_cout_pad % 128 == 0) {
      // few workgroups per CU (batch 1: 288 / 1152 on 256 CUs): the loader-wave form — Generator pass 2.045 -> 1.937 ms; with dozens per
      // CU (B = 8 x 512 frames) the third resident workgroup of the plain form is worth more: 18.80 against 18.89 ms
      // (tools/tune_x6.py, profiles/r03_tune_x6_loader_*.txt)
      // The x3 form (119 / 131 registers, 41 / 20 KB): at 1152 workgroups (C = 128 at batch 1) three plain workgroups per CU beat two
      // loader ones — Generator pass 1.194 against 1.235 ms — and only the 288-workgroup launches (one per CU) keep the loaders.
      const long wgs = (long)((L.L + 63) / 64) * (max_cout_pad / 128) * L.B * L.nprob;
      const bool x3f = conv_x3_ready(L);
      tile = wgs <= (x3f ? 512 : 2048) ? TILE_X6_128x64_LD : TILE_X6_128x64;
    } else tile = max_cout_pad % 64 == 0 ? TILE_X6_64x128 : TILE_X6_32x256;
  }
  // Both tiles: wave tile 32x64 (MI = 1, NI = 2), 32-channel chunks — 164 / 178 registers, 31 / 46 KB of LDS.  Measured and removed in
  // round 3 (tools/tune_x6.py, profiles/r03_tune_x6_*.txt): 64-channel chunks (229 registers, two workgroups per CU: Generator pass
  // 2.11 against 1.97 ms), wave tiles 64x64 as 128x128 / 256x64 / 64x256 workgroups and 32x128 (all within +-2 % at B = 8, slower at
  // B = 1), and an eight-wave form with K split over two wave sets for the 288-workgroup launches of the C = 256 stage (no change).
  const bool x3 = conv_x3_ready(L);  // the two-plane fp16 form where the tile has one (the 128-row tiles)
  bool rd4 = x3 && g_x3_ring != 2;   // its four-tap weight ring: every problem's k = 3 mod 4
  for (int i = 0; i < L.nprob; ++i) rd4 = rd4 && (L.p[i].k & 3) == 3;
  switch (tile) {
    case TILE_X6_128x64:                          // all four waves on the same 64 columns, one 32-row block each
      if (variant_name) *variant_name = x3 ? "conv1d_x3<128x64>" : "conv1d_x6<128x64>";
      // (four workgroups per CU instead of three — 128 registers without the input-mask path, no spills — measured slower too: 0.375 ->
      //  0.386 ms for the six launches, same-box A/B of builds)
      // (the four-tap ring on THIS tile — 168 registers, three workgroups per CU already hiding the distance — measured slower: Generator
      //  pass 1.192 -> 1.200 ms, rocprofv3 59.6 -> 64.2 us per launch; it stays on the two-tap ring)
      if (x3) return launch_x6_variant<4, 1, 1, 2, 32, 128, 0, 2>(stream, L, max_cout_pad);
      return launch_x6_variant<4, 1, 1, 2, 32, 128>(stream, L, max_cout_pad);
    case TILE_X6_128x64_LD:                       // the same tile with two loader waves and two X buffers
      if (x3 && rd4 && g_x3_ring != 3) {
        // x3 with the four-tap ring: FOUR loader waves, one per SIMD (a chunk's split is 1 300 VALU cycles for one of two loaders against
        // 1 150 matrix cycles of a k = 3 chunk: those workgroups were loader-bound, and the two SIMDs that carried a loader ran their MFMA
        // wave behind it) — Generator pass 1.183 / 1.181 -> 1.173 ms for the C = 256 launches, slower for 1 152-workgroup ones (which take
        // the plain tile anyway)
        if (variant_name) *variant_name = "conv1d_x3<128x64,ld4>";
        return launch_x6_variant<4, 1, 1, 2, 32, 128, 4, 2, 4>(stream, L, max_cout_pad);
      }
      if (variant_name) *variant_name = x3 ? "conv1d_x3<128x64,ld>" : "conv1d_x6<128x64,ld>";
      if (x3 && rd4) return launch_x6_variant<4, 1, 1, 2, 32, 128, 2, 2, 4>(stream, L, max_cout_pad);
      if (x3) return launch_x6_variant<4, 1, 1, 2, 32, 128, 2, 2>(stream, L, max_cout_pad);
      return launch_x6_variant<4, 1, 1, 2, 32, 128, 2>(stream, L, max_cout_pad);
    case TILE_X6_64x128:                          // 2 x 2 waves
      if (variant_name) *variant_name = "conv1d_x6<64x128>";
      return launch_x6_variant<2, 2, 1, 2, 32, 192>(stream, L, max_cout_pad);
    case TILE_X6_32x256:                          // C = 32: the four waves side by side in time, all on the same 32-row weight stream
      if (variant_name) *variant_name = "conv1d_x6<32x256>";
      return launch_x6_variant<1, 4, 1, 2, 32, 320>(stream, L, max_cout_pad);
  }
  return -1;
}

}  // namespace bv2

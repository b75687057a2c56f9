// respair_x6.hip — one (dilated conv, conv) pair of ResBlock1 with its residual (reference modules.py:296-309) of an fp32 Generator
// stage (C = 64 / 32 / 16; C = 128 behind an option) in ONE launch, both convs on the bf16 matrix core from exact three-way bf16
// splits (conv_x6.hip's arithmetic: fp32 operands, fp32 results, six of the nine cross products), the intermediate in LDS.
//
// Why.  At C <= 64 the layer-wise split-bf16 convs (conv1d_x6<64x128> / <32x256>, two launches per pair) move five fp32 tensor
// passes per pair at 1.6-2.2 TB/s and are as much HBM- / latency- as MFMA-bound (PMC: MFMA busy 0.36 / 0.23); the fused fp32-MFMA pair
// kernel (resblock_fused.hip, C <= 32) has three passes but runs on the 16x slower fp32 matrix pipe.  Here: two passes AND the bf16
// pipe.  The x tile ([C channels][HT + (k-1)(d+1) columns]) is loaded once (lane = column: coalesced), pre-activated, split into its
// three planes and written channels-last to LDS; conv1 runs on it; t = acc + b1, h = lrelu(t) (zero outside [0, L): conv2's padding)
// is split into its planes in registers and written OVER the dead x planes; conv2 runs on h; the epilogue adds b2 and the fp32
// residual (re-read from L2, coalesced) and stores fp32 [B][C][T].  A tile computes HT columns of h and HT - (k-1) outputs (the k-1
// halo columns are recomputed by the neighbour: 1-8 %).  Same unit order (conv_x6's 32-channel chunks) and the same values at every
// step as the two layer-wise launches: bit-identical to them for C >= 32 (tests/test_x6_gpu.py); C = 16 has no layer-wise x6 form and
// is held to the fp32-MFMA pair kernel at fp32 round-off.  Measured (same-box A/Bs, config 2): C = 32 3.769 -> 3.681 ms per step,
// C = 64 3.72 -> 3.64, C = 16 3.640 -> 3.541; C = 128 no gain against the loader-wave kernel (profiles/r04_ab_x6_pair*.txt).
// out must not alias x (a tile's halo columns are another tile's outputs): the host ping-pongs between two buffers per branch.
// Round 6: the two-plane fp16 form (NP = 2, "x3", default; bv2_kernels.h has the arithmetic): both convs on three products of scaled fp16
// halves, the activation scales taken from the workgroup's own tiles (max |x| staged, max |h| computed) — C = 64 / 32 / 16: 94.4 / 53.7 / 48.9
// -> 61.8 / 36.2 / 32.7 us per launch at batch 1.  NP = 3 (the text above) stays as "conv_x3" = 0 and is the form that is bit-identical to
// the layer-wise launches.
#include <hip/hip_runtime.h>
#include "../bv2_kernels.h"

namespace bv2 {

namespace {

typedef __bf16 pxbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pxbf16x2 __attribute__((ext_vector_type(2)));
typedef float pxf32x16 __attribute__((ext_vector_type(16)));
typedef unsigned pxu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned pxu32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 pxf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pxf16x2 __attribute__((ext_vector_type(2)));
typedef float pxf32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pxf32x16 px_mfma(pxbf16x8 a, pxbf16x8 b, pxf32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ pxf32x16 px_mfma(pxf16x8 a, pxf16x8 b, pxf32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
// the x3 form (bv2_kernels.h): the two fp16 halves of a pair of SCALED values (|a|, |b| < 2^15), round-to-nearest-even
__device__ __forceinline__ void px_split2h(float a, float b, unsigned& u1, unsigned& u2) {
  const pxf16x2 g = __builtin_convertvector((pxf32x2){a, b}, pxf16x2);
  const pxf32x2 f = __builtin_convertvector(g, pxf32x2);
  u1 = __builtin_bit_cast(unsigned, g);
  u2 = __builtin_bit_cast(unsigned, __builtin_convertvector((pxf32x2){a - f[0], b - f[1]}, pxf16x2));
}
// workgroup-wide max of a per-thread magnitude: wave butterflies, one float per wave through `red` (the caller's barrier in between)
__device__ __forceinline__ void px_wave_max_to(float* red, int wid, int lane, float m) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  if (lane == 0) red[wid] = m;
}

__device__ __forceinline__ float px_ld(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ unsigned px_pack(float a, float b) {     // round-to-nearest-even (v_cvt_pk_bf16_f32)
  pxbf16x2 r;
  r[0] = (__bf16)a; r[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float px_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float px_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
// the three planes of a pair of values (conv_x6.hip store_x: plane 1 saturates at the largest bf16)
__device__ __forceinline__ void px_split2(float a, float b, unsigned& u1, unsigned& u2, unsigned& u3) {
  constexpr float M = 3.38953139e38f;
  u1 = px_pack(__builtin_amdgcn_fmed3f(a, -M, M), __builtin_amdgcn_fmed3f(b, -M, M));
  a -= px_lo(u1); b -= px_hi(u1);
  u2 = px_pack(a, b);
  a -= px_lo(u2); b -= px_hi(u2);
  u3 = px_pack(a, b);
}


}  // namespace

// Workgroup = (C / 32 row blocks) x (WNT 64-column blocks) waves; HT = 64 WNT columns of h per tile, XR = HT + 64 staged columns.
//   C = 32 : 1 x 4 waves, HT 256,  77 KB, two workgroups per CU          C = 128: 4 x 2 waves, HT 128, 157 KB, one workgroup per CU
//   C = 64 : 2 x 4 waves, HT 256, 138 KB, one workgroup per CU (two waves per SIMD)
// The 16-channel groups run in passes of two (conv_x6's 32-channel chunks: the layer-wise kernel's fp32 summation order) through a
// ring of 2 groups x 2 taps whose streams jump from one pass's groups to the next's — the registers do not grow with C.
//   C = 16 : 1 x 4 waves, HT 256,  46 KB, three per CU: ONE 16-channel group (passes of one), the upper half of the 32-row block is padding
// NP = 2: the "x3" form (bv2_kernels.h) — two scaled fp16 planes per operand, three products.  The activation scales are the
// WORKGROUP's own: S_x from the max |x| of the tile it staged, S_h from the max |h| of the tile it computed (one float per wave
// through LDS; x costs one more barrier, h rides on the existing one), so a quiet stretch of audio keeps its own 2^15 of range.
template <int PX_C, int WNT, int NP = 3>
__global__ void __launch_bounds__((PX_C >= 32 ? PX_C / 32 : 1) * WNT * 64, PX_C <= 32 ? 2 : 1)
respair_x6_kernel(const FusedLaunch L, const int per_xcd) {
  constexpr int PX_UNIT = NP * 512;               // elements of one (group, tap) unit: NP planes x 64 lanes x 8
  typedef typename std::conditional<NP == 2, pxf16x8, pxbf16x8>::type frag_t;
  typedef __attribute__((address_space(1))) frag_t GlobalFragT;
  constexpr int NI = 2, PX_HT = 64 * WNT, PX_XR = PX_HT + 64, NRG = PX_XR / 64;
  constexpr int MB = PX_C >= 32 ? PX_C / 32 : 1;  // 32-row blocks
  constexpr int NW = MB * WNT;                    // waves
  constexpr int OCT = PX_C / 8;                   // channel octets
  constexpr int OPW = OCT >= NW ? OCT / NW : 1;   // octets a wave stages per column group ...
  constexpr int CGS = OCT >= NW ? 1 : NW / OCT;   // ... of every CGS-th column group (C = 16: two waves share an octet's column groups)
  constexpr int NRGW = (NRG + CGS - 1) / CGS;     // column groups per wave
  static_assert(OCT >= NW ? OCT % NW == 0 : NW % OCT == 0, "octets dealt evenly");
  constexpr int PX_PITCH = PX_C + 8;              // bf16 elements per LDS row (48 / 80 / 144 / 272 B: odd multiples of 16 B)
  constexpr int PX_PLANE = PX_XR * PX_PITCH;
  constexpr int GPP = PX_C >= 32 ? 2 : 1;         // 16-channel groups per pass
  constexpr int NPASS = PX_C / 16 / GPP;          // passes
  constexpr int NJ = PX_C >= 32 ? 4 : PX_C / 8;   // 8-channel sub-blocks of a row block that exist
  extern __shared__ __attribute__((aligned(16))) unsigned short xs[];   // [NP][XR][PITCH]: x planes, then h planes; NP = 2: + 2 x NW floats
  float* const red = reinterpret_cast<float*>(xs + NP * PX_PLANE);       // NP = 2 only: the waves' max |x| (first NW), max |h| (next NW)
  const FusedProb P = L.p[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WNT, wn = wid % WNT;       // 32-row block, 64-column block of this wave's tile
  const int l31 = lane & 31, lh = lane >> 5;
  const int k = P.k, dil = P.dil;
  const int BT = PX_HT - (k - 1);
  const int bx = blockIdx.x;
  const int vt = per_xcd ? (bx & 7) * per_xcd + (bx >> 3) : bx;
  const int t0 = vt * BT;
  if (t0 >= L.L) return;
  const int b = blockIdx.y;
  int Lin = L.L;
  if (L.lens) {
    const int64_t lv = L.lens[b] * L.len_mul;
    Lin = lv < Lin ? (int)lv : Lin;
    if (t0 >= Lin) return;
  }
  const int p2 = (k - 1) / 2, p1 = p2 * dil;
  const float slope = L.slope;
  const int64_t bstride = (int64_t)PX_C * L.L;
  const float* const x0p = P.x + (int64_t)b * bstride;
  const unsigned x_rs4 = 4u * (unsigned)L.L;
  const unsigned wlane = 16u * (unsigned)lane;

  // ---- weight ring (conv_x6.hip): ring group gl carries group 2 c + gl of pass c through its k taps (two slots: taps j, j + 1), then
  // jumps to the next pass's group; k is odd, so pass c starts at slot parity c & 1
  frag_t ar[GPP][2][NP];
  const uint16_t* wq[GPP];
  auto load_unit = [&](int gl, int SL, int step) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
      ar[gl][SL][p] = *(const GlobalFragT*)(reinterpret_cast<const char*>(wq[gl]) + wlane + 1024u * (unsigned)p);
    wq[gl] += step;
  };
  const int last_step = ((GPP - 1) * k + 1) * PX_UNIT;                  // from (g, k-1) to (g + GPP, 0)
  const int wrap_step = -(((NPASS - 1) * GPP * k + (k - 1)) * PX_UNIT); // from the last pass back to the first (valid memory, values unused)
  auto step_after = [&](int jl, int cl) __attribute__((always_inline)) {
    return jl + 1 < k ? PX_UNIT : (cl + 1 < NPASS ? last_step : wrap_step);
  };
  auto prime = [&](const uint16_t* w6) __attribute__((always_inline)) {
#pragma unroll
    for (int gl = 0; gl < GPP; ++gl) wq[gl] = w6 + (int64_t)(wm * (PX_C / 16) + gl) * k * PX_UNIT;
    const int s0 = step_after(0, 0), s1 = step_after(1, 0);          // k >= 3
#pragma unroll
    for (int gl = 0; gl < GPP; ++gl) { load_unit(gl, 0, s0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int gl = 0; gl < GPP; ++gl) { load_unit(gl, 1, s1); __builtin_amdgcn_sched_barrier(0); }
  };
  prime(NP == 2 ? P.w31 : P.w61);
  float s1 = 1.f;                                 // NP = 2: 1 / (S_w1 S_x), the scale of conv1's accumulators

  // ---- stage x: wave `wid` loads channel octet `wid` of every 64-column group (lane = column), lrelu, split, channels-last planes
  {
    const int tbase = t0 - p2 - p1;
    const int XW = PX_HT + (k - 1) * dil;
    float xr[NRGW][OPW][8];
    float colsc[NRGW];
    const int rg0 = OCT >= NW ? 0 : wid / OCT;    // this wave's first column group
    const int oct0 = OCT >= NW ? wid : wid % OCT; // ... and first octet
#pragma unroll
    for (int i = 0; i < NRGW; ++i) {
      const int rg = rg0 + CGS * i;
      const int r = rg * 64 + lane;
      const int t = tbase + r;
      const bool tok = r < XW && t >= 0 && t < Lin;
      colsc[i] = tok ? 1.f : 0.f;
      const unsigned tc = 4u * (unsigned)(t < 0 ? 0 : (t >= Lin ? Lin - 1 : t));
#pragma unroll
      for (int o = 0; o < OPW; ++o) {
        const unsigned row0 = (unsigned)((oct0 + NW * o) * 8) * x_rs4;
#pragma unroll
        for (int e = 0; e < 8; ++e) xr[i][o][e] = px_ld(x0p, row0 + (unsigned)e * x_rs4 + tc);
      }
    }
    if constexpr (NP == 2) {
      // S_x: |lrelu(x)| <= |x|, so the tile's max |x| over the valid columns bounds every staged value
      float mx = 0.f;
#pragma unroll
      for (int i = 0; i < NRGW; ++i)
#pragma unroll
        for (int o = 0; o < OPW; ++o)
#pragma unroll
          for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(xr[i][o][e]) * colsc[i]);
      px_wave_max_to(red, wid, lane, mx);
      __syncthreads();
      float M = red[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) M = fmaxf(M, red[w]);
      const unsigned ex = x3_scale_exp(__float_as_uint(M));
      const float Sx = x3_scale(ex);
      s1 = x3_scale_inv(ex) * *P.w3inv1;
#pragma unroll
      for (int i = 0; i < NRGW; ++i) colsc[i] *= Sx;
    }
#pragma unroll
    for (int i = 0; i < NRGW; ++i) {
      const int rg = rg0 + CGS * i;
#pragma unroll
      for (int o = 0; o < OPW; ++o) {
        pxu32x4 q1, q2, q3;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          float a = xr[i][o][2 * w], bq = xr[i][o][2 * w + 1];
          const float an = a * slope, bn = bq * slope;
          a = a < 0.f ? an : a;
          bq = bq < 0.f ? bn : bq;
          a *= colsc[i]; bq *= colsc[i];
          unsigned u1, u2, u3 = 0;
          if constexpr (NP == 2) px_split2h(a, bq, u1, u2);
          else px_split2(a, bq, u1, u2, u3);
          q1[w] = u1; q2[w] = u2; q3[w] = u3;
        }
        if (CGS == 1 || rg < NRG) {
          unsigned short* dst = xs + (rg * 64 + lane) * PX_PITCH + (oct0 + NW * o) * 8;
          *reinterpret_cast<pxu32x4*>(dst) = q1;
          *reinterpret_cast<pxu32x4*>(dst + PX_PLANE) = q2;
          if constexpr (NP == 3) *reinterpret_cast<pxu32x4*>(dst + 2 * PX_PLANE) = q3;
        }
      }
    }
  }
  __syncthreads();

  pxf32x16 acc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
  const unsigned short* const xlane = xs + (wn * 64 + l31) * PX_PITCH + lh * 8;

  // one GEMM over the tile in LDS: acc[ni] += sum over (pass c, tap j, group gl of the pass) of the six cross products (conv_x6.hip's unit)
  auto gemm = [&](int tap_step) __attribute__((always_inline)) {
    frag_t bb[2][NI][NP];
    auto pass = [&](int c, int PAR) __attribute__((always_inline)) {              // PAR: a literal at every (inlined) call site
      const unsigned short* xrow = xlane + c * (16 * GPP);
      // B double buffer: alternates with the group inside a tap (two groups per pass) or with the tap's ring slot (one group per pass)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int p = 0; p < NP; ++p)
          bb[GPP == 2 ? 0 : PAR][ni][p] = *reinterpret_cast<const frag_t*>(xrow + ni * 32 * PX_PITCH + p * PX_PLANE);
      auto tap = [&](int j, int SL) __attribute__((always_inline)) {
        const unsigned short* xnext = (j + 1 < k) ? xrow + tap_step : xrow;
        int jl = j + 2, cl = c;                     // the unit loaded during this tap: tap jl of pass cl (branch-free wrap)
        if (jl >= k) { jl -= k; ++cl; }
        if (cl >= NPASS) cl -= NPASS;
        const int step = step_after(jl, cl);
#pragma unroll
        for (int gl = 0; gl < GPP; ++gl) {
          const int cur = GPP == 2 ? gl : SL;
          {
            const unsigned short* xn = (gl + 1 < GPP) ? xrow + 16 : xnext;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
              for (int p = 0; p < NP; ++p)
                bb[cur ^ 1][ni][p] = *reinterpret_cast<const frag_t*>(xn + ni * 32 * PX_PITCH + p * PX_PLANE);
          }
#define PX_PROD(WP, XP)                                                                                              \
          _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                          \
            acc[ni] = px_mfma(ar[gl][SL][WP], bb[cur][ni][XP], acc[ni]);
          if constexpr (NP == 3) { PX_PROD(2, 0) PX_PROD(1, 1) PX_PROD(0, 2) }
          PX_PROD(1, 0) PX_PROD(0, 1) PX_PROD(0, 0)
#undef PX_PROD
          load_unit(gl, SL, step);
          constexpr int NM = NI * (NP == 3 ? 6 : 3), NDS = NI * NP, NVM = NP;
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
          __builtin_amdgcn_sched_barrier(0);
        }
        xrow = xnext;
      };
      for (int j = 0; j + 1 < k; j += 2) { tap(j, PAR); tap(j + 1, PAR ^ 1); }
      tap(k - 1, PAR);
    };
    int c = 0;
    for (; c + 1 < NPASS; c += 2) { pass(c, 0); pass(c + 1, 1); }
    if (c < NPASS) pass(c, 0);
  };
  gemm(dil * PX_PITCH);

  // ---- h = lrelu(conv1 + b1), zero outside [0, L) (conv2's padding), split into its planes, written over the x planes
  prime(NP == 2 ? P.w32 : P.w62);
  float b1v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) b1v[r] = P.b1[wm * 32 + 4 * lh + (r & 3) + 8 * (r >> 2)];
  float s2 = 1.f, Sh = 1.f;                       // NP = 2: 1 / (S_w2 S_h) for conv2's accumulators, S_h for the h planes
  if constexpr (NP == 2) {
    // h in place of the accumulators (fp32), its max through LDS on the barrier that frees the x planes
    float mh = 0.f;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int th = t0 - p2 + wn * 64 + ni * 32 + l31;
      const float ok = (th >= 0 && th < Lin) ? 1.f : 0.f;
#pragma unroll
      for (int r = 0; r < 4 * NJ; ++r) {
        float t = __builtin_fmaf(acc[ni][r], s1, b1v[r]);
        const float tn = t * slope;
        t = t < 0.f ? tn : t;
        acc[ni][r] = t * ok;
        mh = fmaxf(mh, fabsf(acc[ni][r]));
      }
    }
    px_wave_max_to(red + NW, wid, lane, mh);
  }
  __syncthreads();                                // every wave is done reading the x planes
  if constexpr (NP == 2) {
    float M = red[NW];
#pragma unroll
    for (int w = 1; w < NW; ++w) M = fmaxf(M, red[NW + w]);
    const unsigned eh = x3_scale_exp(__float_as_uint(M));
    Sh = x3_scale(eh);
    s2 = x3_scale_inv(eh) * *P.w3inv2;
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int hc = wn * 64 + ni * 32 + l31;       // column of h: time t0 - p2 + hc
    const int th = t0 - p2 + hc;
    const float ok = (th >= 0 && th < Lin) ? 1.f : 0.f;
    unsigned short* dst = xs + hc * PX_PITCH + wm * 32 + 4 * lh;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {                // channels 8 j + 4 lh + {0, 1, 2, 3} = registers 4 j .. 4 j + 3
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (NP == 2) { v[i] = acc[ni][4 * j + i] * Sh; continue; }
        float t = acc[ni][4 * j + i] + b1v[4 * j + i];
        const float tn = t * slope;
        t = t < 0.f ? tn : t;
        v[i] = t * ok;
      }
      unsigned a1, a2, a3 = 0, c1, c2, c3 = 0;
      if constexpr (NP == 2) { px_split2h(v[0], v[1], a1, a2); px_split2h(v[2], v[3], c1, c2); }
      else { px_split2(v[0], v[1], a1, a2, a3); px_split2(v[2], v[3], c1, c2, c3); }
      *reinterpret_cast<pxu32x2*>(dst + 8 * j) = pxu32x2{a1, c1};
      *reinterpret_cast<pxu32x2*>(dst + 8 * j + PX_PLANE) = pxu32x2{a2, c2};
      if constexpr (NP == 3) *reinterpret_cast<pxu32x2*>(dst + 8 * j + 2 * PX_PLANE) = pxu32x2{a3, c3};
    }
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
  // the epilogue's operands: in flight under conv2
  float b2v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) b2v[r] = P.b2[wm * 32 + 4 * lh + (r & 3) + 8 * (r >> 2)];
  __syncthreads();
  gemm(PX_PITCH);

  // ---- out = conv2 + b2 + x: fp32 [B][C][T], lane = column (coalesced residual reads and stores)
  {
    float* const outb = P.out + (int64_t)b * bstride;
    const int rows = L.L - t0 < BT ? L.L - t0 : BT;
    float rv[NI][16];
    bool colok[NI];
    unsigned off0[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int oc = wn * 64 + ni * 32 + l31;
      colok[ni] = oc < rows;
      const int t = t0 + (colok[ni] ? oc : 0);
      off0[ni] = (unsigned)(wm * 32 + 4 * lh) * (unsigned)L.L + (unsigned)t;
#pragma unroll
      for (int r = 0; r < 4 * NJ; ++r) rv[ni][r] = px_ld(x0p, 4u * (off0[ni] + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)L.L));
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 4 * NJ; ++r) {           // rows (r & 3) + 8 (r >> 2) + 4 lh < C
        const float v = (NP == 2 ? __builtin_fmaf(acc[ni][r], s2, b2v[r]) : acc[ni][r] + b2v[r]) + rv[ni][r];
        if (colok[ni]) outb[off0[ni] + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)L.L] = v;
      }
  }
}

bool respair_x6_supported(int C, int k, int dil) {
  if ((C != 16 && C != 32 && C != 64 && C != 128) || k < 3 || k % 2 == 0 || dil < 1) return false;
  const int HT = C == 128 ? 128 : 256;
  return (k - 1) * dil <= 64 && HT - (k - 1) >= HT / 2;
}

int launch_respair_x6(hipStream_t stream, const FusedLaunch& F) {
  if (F.nprob < 1 || F.nprob > 3 || F.B < 1 || F.L < 1 || (F.C != 16 && F.C != 32 && F.C != 64 && F.C != 128)) return -1;
  const int HT = F.C == 128 ? 128 : 256;
  if ((int64_t)F.C * F.L >= (1ll << 29)) return -1;               // 32-bit byte offsets inside a batch item
  int ntx = 0;
  for (int i = 0; i < F.nprob; ++i) {
    const FusedProb& p = F.p[i];
    if (!respair_x6_supported(F.C, p.k, p.dil) || !p.x || !p.out || p.x == p.out || !p.w61 || !p.w62 || !p.b1 || !p.b2) return -1;
    const int BT = HT - (p.k - 1);
    const int n = (F.L + BT - 1) / BT;
    ntx = n > ntx ? n : ntx;
  }
  const int per_xcd = ntx >= 16 ? (ntx + 7) / 8 : 0;
  dim3 grid(per_xcd ? per_xcd * 8 : ntx, F.B, F.nprob);
  bool x3 = true;                                 // the two-plane fp16 form: every problem carries its planes and their scales
  for (int i = 0; i < F.nprob; ++i) x3 = x3 && F.p[i].w31 && F.p[i].w32 && F.p[i].w3inv1 && F.p[i].w3inv2;
  if (x3) {
    const size_t lds3 = (size_t)2 * (HT + 64) * (F.C + 8) * 2 + 256;
    if (F.C == 16) {
      ensure_dyn_lds((const void*)respair_x6_kernel<16, 4, 2>, lds3);
      hipLaunchKernelGGL((respair_x6_kernel<16, 4, 2>), grid, dim3(256), lds3, stream, F, per_xcd);
    } else if (F.C == 32) {
      ensure_dyn_lds((const void*)respair_x6_kernel<32, 4, 2>, lds3);
      hipLaunchKernelGGL((respair_x6_kernel<32, 4, 2>), grid, dim3(256), lds3, stream, F, per_xcd);
    } else if (F.C == 64) {
      ensure_dyn_lds((const void*)respair_x6_kernel<64, 4, 2>, lds3);
      hipLaunchKernelGGL((respair_x6_kernel<64, 4, 2>), grid, dim3(512), lds3, stream, F, per_xcd);
    } else {
      ensure_dyn_lds((const void*)respair_x6_kernel<128, 2, 2>, lds3);
      hipLaunchKernelGGL((respair_x6_kernel<128, 2, 2>), grid, dim3(512), lds3, stream, F, per_xcd);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  const size_t lds = (size_t)3 * (HT + 64) * (F.C + 8) * 2;
  if (F.C == 16) {
    ensure_dyn_lds((const void*)respair_x6_kernel<16, 4>, lds);
    hipLaunchKernelGGL((respair_x6_kernel<16, 4>), grid, dim3(256), lds, stream, F, per_xcd);
  } else if (F.C == 32) {
    ensure_dyn_lds((const void*)respair_x6_kernel<32, 4>, lds);
    hipLaunchKernelGGL((respair_x6_kernel<32, 4>), grid, dim3(256), lds, stream, F, per_xcd);
  } else if (F.C == 64) {
    ensure_dyn_lds((const void*)respair_x6_kernel<64, 4>, lds);
    hipLaunchKernelGGL((respair_x6_kernel<64, 4>), grid, dim3(512), lds, stream, F, per_xcd);
  } else {
    ensure_dyn_lds((const void*)respair_x6_kernel<128, 2>, lds);
    hipLaunchKernelGGL((respair_x6_kernel<128, 2>), grid, dim3(512), lds, stream, F, per_xcd);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace bv2

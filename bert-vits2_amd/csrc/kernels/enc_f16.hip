// enc_f16.hip — the convolutions / projections of the attention Encoder stacks (reference attentions.py:103-120:
// conv_q/k/v, conv_o, FFN conv_1 / conv_2) in fp16 on the gfx950 matrix core (v_mfma_f32_32x32x16_f16, fp32 accumulate).
// BASELINE config 5 "fp16 flow + fp32 spline" / config 3: at batch 32 the flow's fp32 FFN convs are 40 % of the step once
// the Generator runs in bf16; here they move to the half-precision MFMA rate while LayerNorm, softmax, the residual stream
// and the attention core stay fp32.
//
//   GEMM view: M = C_out (A operand = fp16 weight fragments, streamed global -> registers through an 8-deep ring),
//              N = time (B operand = 8 consecutive input channels of one time step: ONE ds_read_b128 from the LDS tile),
//              K = (16-channel group, tap).
//
// The LDS tile is channels-last [rows = 32*NI + (k-1)*dil][pitch = ck + 8] fp16 (pitch = odd multiple of 16 B: the
// ds_read_b128 of 32 consecutive rows are conflict-free); C_in is walked in chunks of <= 256 channels so that the tile fits
// twice per CU for any C_in (FFN conv_2 has C_in = 768).  Two input forms and two output forms, so that the kernel sits
// directly between the fp32 [B][C][T] tensors of the surrounding fp32 kernels with no conversion pass:
//   IN_CT : fp32 [B][C][T] (+ per-column mask) -> rounded to fp16 and TRANSPOSED while it is staged (lane = 16 time steps x
//           4 channel pairs: 64-byte global segments, conflict-free 4-byte LDS writes);
//   IN_CL : fp16 channels-last [B][T][C] (the FFN hidden activation), 16-byte pieces;
//   OUT_CT: fp32 [B][C][ld] with bias, masks and the residual add (the D fragment's lane index is the time step: each
//           register stores 32 consecutive floats of one channel row);
//   OUT_CL: fp16 [B][T][C] with bias, ReLU and mask (FFN hidden).
// Weight stream: cl_w_index (bv2_kernels.h) in fp16, [m-tile][unit = (ci/16)*k + tap][lane][8] — 1 KB per unit.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "../bv2_kernels.h"

namespace bv2 {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int HC_PD = 8;          // weight prefetch ring depth (units of NI MFMAs)
constexpr int HC_CK = 256;        // input channels per LDS chunk

__device__ __forceinline__ unsigned h_pack(float a, float b) {       // round-to-nearest-even (v_cvt_f16_f32)
  f16x2 r;
  r[0] = (_Float16)a; r[1] = (_Float16)b;
  return __builtin_bit_cast(unsigned, r);
}

// acc[ni] += sum over units u = (s, j) of Wfrag(u) x B(u, ni);  B(u, ni) = channels [16s + 8lh, +8) of LDS row
// (ni*32 + l31 + j*tstep) relative to xb.  wp = this wave's contiguous weight stream (+ lane*8 elements).
template <int NI>
__device__ __forceinline__ void hc_gemm(f32x16 (&acc)[NI], const uint16_t* wp, int U, int k, const unsigned short* xb,
                                        int pitch, int tstep) {
  f16x8 ar[HC_PD];
  int lu = 0;
  auto load_unit = [&](int slot) __attribute__((always_inline)) {
    const int uc = lu < U ? lu : U - 1;                             // past the end: re-read the last unit, result unused
    ar[slot] = *reinterpret_cast<const f16x8*>(wp + (int64_t)uc * 512);
    ++lu;
  };
#pragma unroll
  for (int i = 0; i < HC_PD; ++i) { load_unit(i); __builtin_amdgcn_sched_barrier(0); }
  f16x8 bb[2][NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) bb[0][ni] = *reinterpret_cast<const f16x8*>(xb + ni * 32 * pitch);
  int s = 0, j = 0;
  for (int u0 = 0; u0 < U; u0 += HC_PD) {
#pragma unroll
    for (int i = 0; i < HC_PD; ++i) {
      if (u0 + i < U) {
        int jn = j + 1, sn = s;
        if (jn == k) { jn = 0; ++sn; }
        const bool more = u0 + i + 1 < U;
        const unsigned short* xn = xb + (more ? jn : j) * tstep * pitch + (more ? sn : s) * 16;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bb[(i & 1) ^ 1][ni] = *reinterpret_cast<const f16x8*>(xn + ni * 32 * pitch);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[i], bb[i & 1][ni], acc[ni], 0, 0, 0);
        j = jn; s = sn;
      }
      load_unit(i);
      __builtin_amdgcn_sched_group_barrier(0x100, NI, 0);   // next unit's LDS reads first, then the MFMAs, then the ring load
      __builtin_amdgcn_sched_group_barrier(0x008, NI, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_barrier(0);          // keep program order: the ring's vmcnt distances stay HC_PD - 1 units
    }
  }
}

// Tap-major form for a chunk of 16*G channels known at compile time (G = 12: C_in = 192; G = 16: 256-channel chunks): ring
// slot = channel group (unit (s, tap j) is followed by (s, j+1)), groups of a tap fully unrolled, every LDS offset an
// immediate, explicit global loads — 3.5 instead of ~9 instructions per MFMA (same restructuring as gen_bf16.hip cl_gemm_tm).
typedef __attribute__((address_space(1))) f16x8 GlobalFragH;
template <int NI, int G, int PITCH = 16 * G + 8>   // PITCH: elements per LDS row (a tile wider than the 16*G channels this call runs: K split)
__device__ __forceinline__ void hc_gemm_tm(f32x16 (&acc)[NI], const uint16_t* wbase, unsigned wlane_bytes, int k,
                                           const unsigned short* xb, int tstep) {
  static_assert(G % 2 == 0, "the B double buffer alternates with the group index");
  f16x8 ar[G];
  const uint16_t* wq[G];
  const int first_step = k > 1 ? 512 : 0;
#pragma unroll
  for (int s = 0; s < G; ++s) {
    wq[s] = wbase + (int64_t)s * k * 512;
    ar[s] = *(const GlobalFragH*)(reinterpret_cast<const char*>(wq[s]) + wlane_bytes);
    wq[s] += first_step;
    __builtin_amdgcn_sched_barrier(0);
  }
  f16x8 bb[2][NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) bb[0][ni] = *reinterpret_cast<const f16x8*>(xb + ni * 32 * PITCH);
  const unsigned short* xrow = xb;
  for (int j = 0; j < k; ++j) {
    const unsigned short* xnext = (j + 1 < k) ? xrow + tstep * PITCH : xrow;
    const int step = (j + 2 < k) ? 512 : 0;
#pragma unroll
    for (int s = 0; s < G; ++s) {
      const unsigned short* xn = (s + 1 < G) ? xrow + (s + 1) * 16 : xnext;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bb[(s & 1) ^ 1][ni] = *reinterpret_cast<const f16x8*>(xn + ni * 32 * PITCH);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[s], bb[s & 1][ni], acc[ni], 0, 0, 0);
      ar[s] = *(const GlobalFragH*)(reinterpret_cast<const char*>(wq[s]) + wlane_bytes);
      wq[s] += step;
      __builtin_amdgcn_sched_group_barrier(0x100, NI, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NI, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    xrow = xnext;
  }
}

// fp32 [C][T] rows -> fp16 channels-last LDS tile (rows tb .. tb+rows-1, channels c0 .. c0+ck-1), x mask, zero padding.
// Round 5: a lane loads FOUR consecutive time steps of its two channels as one 16-byte load each (a wave instruction = 4 time quads x 16
// channel pairs = 16 time steps x 32 channels; a channel row is read in 64-byte segments as before, with a quarter of the load
// instructions), rounds, and writes four (channel pair) words to rows 4 tq .. 4 tq + 3.  Rows of x are only 4-byte aligned in general
// (T_y is arbitrary): the vector type says so, and the hardware's unaligned-access mode serves a dwordx4 at any dword address.  The
// scalar form (one 4-byte load per element, 54 load instructions per lane for a 132-row x 192-channel tile) was 16-26k cycles of a
// 39-57k-cycle workgroup at B = 32 (tools/timeline.py, profiles/r05_timeline_c3_f16_convs.txt).
typedef float hc_f32x4u __attribute__((ext_vector_type(4), aligned(4)));
template <int NT>
__device__ __forceinline__ void hc_stage_ct(unsigned short* xs, int pitch, const float* x, int x_rstride, const float* mask,
                                            int tb, int rows, int c0, int ck, int Lin, int tid) {
  const int lane = tid & 63, wid = tid >> 6;
  const int tq = lane & 3, cp = lane >> 2;        // 4 time quads x 16 channel pairs per wave instruction
  const int tblocks = (rows + 15) >> 4, cblocks = (ck + 31) >> 5;
  const int units = tblocks * cblocks;
  constexpr int NWV = NT / 64;
  unsigned* xs32 = reinterpret_cast<unsigned*>(xs);
  const int p32 = pitch >> 1;
  constexpr int HQ = 7;                           // units per thread in flight per batch: a 132-row x 192-channel tile in ONE round trip with 8 waves
  for (int u0 = wid; u0 < units; u0 += HQ * NWV) {
    float a[HQ][4], b[HQ][4], m[HQ][4];
    int dst[HQ], r0[HQ];
    bool wr[HQ];
#pragma unroll
    for (int q = 0; q < HQ; ++q) {
      int u = u0 + q * NWV;
      const bool inb = u < units;
      u = inb ? u : units - 1;
      const int cbk = u / tblocks, tbk = u - cbk * tblocks;
      const int r = tbk * 16 + tq * 4;
      const int t = tb + r;
      int cl = cbk * 32 + cp * 2;                  // channel inside the chunk (ck is a multiple of 16: a pair never straddles its end)
      const bool cok = cl < ck;
      cl = cok ? cl : 0;
      const float* src = x + (int64_t)(c0 + cl) * x_rstride;
      if (t >= 0 && t + 4 <= Lin) {
        const hc_f32x4u va = *reinterpret_cast<const hc_f32x4u*>(src + t);
        const hc_f32x4u vb = *reinterpret_cast<const hc_f32x4u*>(src + x_rstride + t);
        hc_f32x4u vm = hc_f32x4u{1.f, 1.f, 1.f, 1.f};
        if (mask) vm = *reinterpret_cast<const hc_f32x4u*>(mask + t);
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[q][i] = va[i]; b[q][i] = vb[i]; m[q][i] = vm[i]; }
      } else {                                     // a quad that straddles [0, Lin): element by element, clamped
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ti = t + i;
          const bool ok = ti >= 0 && ti < Lin;
          const int tc = ti < 0 ? 0 : (ti >= Lin ? Lin - 1 : ti);
          a[q][i] = src[tc];
          b[q][i] = src[x_rstride + tc];
          m[q][i] = ok ? (mask ? mask[tc] : 1.f) : 0.f;
        }
      }
      wr[q] = inb && cok;
      r0[q] = r;
      dst[q] = r * p32 + (cl >> 1);
    }
#pragma unroll
    for (int q = 0; q < HQ; ++q)
      if (wr[q]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (r0[q] + i < rows) xs32[dst[q] + i * p32] = h_pack(a[q][i] * m[q][i], b[q][i] * m[q][i]);
      }
  }
}

// fp16 channels-last [L][cin] rows -> LDS tile (16-byte pieces)
template <int NT>
__device__ __forceinline__ void hc_stage_cl(unsigned short* xs, int pitch, const uint16_t* x, int cin, int tb, int rows,
                                            int c0, int ck, int Lin, int tid) {
  const int ppr = ck >> 3;
  const int total = rows * ppr;
  constexpr int CQ = 9;                           // 16-byte pieces per thread in flight: a 256-channel x 132-row chunk in ONE round trip
  for (int base = 0; base < total; base += CQ * NT) {
    u32x4 v[CQ];
    int dst[CQ];
    bool ok[CQ], inb[CQ];
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
      int p = base + q * NT + tid;
      inb[q] = p < total;
      p = inb[q] ? p : total - 1;
      const int r = p / ppr, cb = p - r * ppr;
      const int t = tb + r;
      ok[q] = t >= 0 && t < Lin;
      const int tc = t < 0 ? 0 : (t >= Lin ? Lin - 1 : t);
      dst[q] = r * pitch + cb * 8;
      v[q] = *reinterpret_cast<const u32x4*>(x + (int64_t)tc * cin + c0 + cb * 8);
    }
#pragma unroll
    for (int q = 0; q < CQ; ++q)
      if (inb[q]) *reinterpret_cast<u32x4*>(xs + dst[q]) = ok[q] ? v[q] : u32x4{0u, 0u, 0u, 0u};
  }
}

// KS > 1 (round 5, the FFN's conv_2: C_in = 768 -> 192 rows, k = 5): the K dimension split INSIDE the workgroup.  One 32-row output tile per
// wave makes that conv 6 waves per 64 columns = 192 workgroups at B = 32, each streaming the WHOLE 1.47 MB weight set through a loop of
// 240 units per wave that tools/timeline.py had at 58k of the workgroup's 88k ticks (MFMA-only: 15k) with three re-stagings of the tile in
// it.  Here the whole C_in = KS*16*G tile is staged once (68 rows x 1552 B = 105 KB), wave (wm, kh) runs channel groups [kh*G, +G) of
// output tile wm, the KS partial accumulators meet in LDS (over the dead tile) and the waves kh = 0 run the epilogue: half the loop per
// wave, twice the waves per CU to hide the weight stream's latency, one staging phase instead of three.
// LN: the channel LayerNorm of the result in the epilogue (its own instantiations: the 50 extra live registers stay out of the others)
template <int WN, int NI, bool IN_CT, bool OUT_CT, int G, int KS = 1, bool LN = false>   // G > 0: every chunk has 16*G channels (tap-major GEMM)
__global__ void __launch_bounds__(64 * WN * KS) conv_f16_kernel(const HcLaunch L, const int ngrp) {
  constexpr int NT = 64 * WN * KS, BT = 32 * NI;
  static_assert(KS == 1 || (G > 0 && !IN_CT && OUT_CT), "the in-workgroup K split: tap-major, channels-last input, [C][T] output");
  extern __shared__ __attribute__((aligned(16))) unsigned short xs[];
  int b, cg, tile, pz = blockIdx.z;
  if (L.xcd_b) {                                                    // batch item -> XCD affinity (bv2_kernels.h xcd_decode)
    int r;
    if (!xcd_decode(blockIdx.x, L.xcd_per, L.B, b, r)) return;
    tile = r % L.xcd_gx; r /= L.xcd_gx;
    cg = r % ngrp; pz = r / ngrp;
  } else {
    b = blockIdx.y / ngrp;
    cg = blockIdx.y - b * ngrp;
    tile = blockIdx.x;
  }
  const HcProb& P = L.p[pz];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;                    // timeline stamps (tools/timeline.py; L.dbg is null in the product)
  if (L.dbg) ts0 = __builtin_amdgcn_s_memtime();
  const int wm = KS > 1 ? wid % WN : wid, kh = KS > 1 ? wid / WN : 0;
  const int mt = cg * WN + wm;
  const int t0 = tile * BT;
  const int cin = P.cin, k = P.k, dil = P.dil;
  const int rows = BT + (k - 1) * dil;
  const bool active = mt * 32 < P.cout_pad;
  const int Utot = (cin >> 4) * k;

  f32x16 acc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;

  if constexpr (KS > 1) {
    constexpr int PITCH = KS * 16 * G + 8;         // cin == KS*16*G (launcher)
    hc_stage_cl<NT>(xs, PITCH, static_cast<const uint16_t*>(P.x) + (int64_t)b * P.x_bstride, cin, t0 - P.pad_left, rows, 0, cin, P.Lin, tid);
    __syncthreads();
    if (L.dbg) ts1 = __builtin_amdgcn_s_memtime();
    if (active)
      hc_gemm_tm<NI, G, PITCH>(acc, P.w + ((int64_t)mt * Utot + (int64_t)kh * G * k) * 512, 16u * (unsigned)lane, k,
                               xs + l31 * PITCH + lh * 8 + kh * G * 16, dil);
    __syncthreads();                               // every wave is done reading the tile: it becomes the merge buffer [wm][kh-1][ni][r][lane]
    float* red = reinterpret_cast<float*>(xs);
    if (kh > 0) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((((kh - 1) * WN + wm) * NI + ni) * 16 + r) * 64 + lane] = acc[ni][r];
    }
    __syncthreads();
    if (kh > 0) {
      // the kh == 0 waves go on to the LayerNorm epilogue's three barriers: the K-half waves keep the workgroup's barrier count whole
      // (a wave that has ended is dropped from s_barrier on CDNA, but HIP leaves a barrier not reached by every thread undefined)
      if constexpr (LN && OUT_CT) { __syncthreads(); __syncthreads(); __syncthreads(); }
      return;
    }
#pragma unroll
    for (int q = 0; q < KS - 1; ++q)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ni][r] += red[(((q * WN + wm) * NI + ni) * 16 + r) * 64 + lane];
  } else {
  for (int c0 = 0; c0 < cin; c0 += HC_CK) {
    const int ck = cin - c0 < HC_CK ? cin - c0 : HC_CK;
    const int pitch = ck + 8;
    if (c0) __syncthreads();                       // every wave is done reading the previous chunk's tile
    if (IN_CT)
      hc_stage_ct<NT>(xs, pitch, static_cast<const float*>(P.x) + (int64_t)b * P.x_bstride, P.x_rstride,
                      P.in_mask ? P.in_mask + (int64_t)b * P.in_mask_bstride : nullptr, t0 - P.pad_left, rows, c0, ck, P.Lin, tid);
    else
      hc_stage_cl<NT>(xs, pitch, static_cast<const uint16_t*>(P.x) + (int64_t)b * P.x_bstride, cin, t0 - P.pad_left, rows,
                      c0, ck, P.Lin, tid);
    __syncthreads();
    if (L.dbg && c0 == 0) ts1 = __builtin_amdgcn_s_memtime();
    if (active) {
      if constexpr (G > 0)
        hc_gemm_tm<NI, G>(acc, P.w + ((int64_t)mt * Utot + (int64_t)(c0 >> 4) * k) * 512, 16u * (unsigned)lane, k,
                          xs + l31 * pitch + lh * 8, dil);
      else
        hc_gemm<NI>(acc, P.w + ((int64_t)mt * Utot + (int64_t)(c0 >> 4) * k) * 512 + lane * 8, (ck >> 4) * k, k,
                    xs + l31 * pitch + lh * 8, pitch, dil);
    }
  }
  }
  if (L.dbg) ts2 = __builtin_amdgcn_s_memtime();
  if (!active) return;

  const int cout = P.cout;
  if (OUT_CT) {
    // fp32 [C][T] epilogue.  The residual may alias the output (in-place updates), so a loop that loads a residual element,
    // then stores an output element, is SERIALISED by the compiler — 16 x NI dependent L2 round trips (tools/timeline.py:
    // 38-43k cycles, more than the whole rest of the workgroup).  All residual / bias loads of the tile are issued first (each
    // thread only ever reads elements it writes itself, and reads them before), then the arithmetic, then the stores;
    // 32-bit offsets from wave-uniform bases.
    float* const outp = static_cast<float*>(P.out) + (int64_t)b * P.out_bstride;
    const int res_mode = P.res_mode, act = P.act, mask_pre = P.mask_pre, mask_post = P.mask_post;
    const float* const resp = res_mode != RES_NONE ? P.res + (int64_t)b * P.res_bstride : nullptr;
    const float* const biasp = P.bias;
    const float* const omp = P.out_mask ? P.out_mask + (int64_t)b * P.out_mask_bstride : nullptr;
    const unsigned o_rs = (unsigned)P.out_rstride;
    const int Lout = L.L;
    float bs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      co = co < cout ? co : cout - 1;
      bs[r] = biasp ? biasp[co] : 0.f;
    }
    float rv[NI][16], om[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int t = t0 + ni * 32 + l31;
      const int tc = t < Lout ? t : Lout - 1;
      om[ni] = omp ? omp[tc] : 1.f;
      if (resp) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          co = co < cout ? co : cout - 1;
          rv[ni][r] = resp[(unsigned)co * o_rs + (unsigned)tc];
        }
      }
    }
    if constexpr (LN) {
      const float* const lng = P.ln_gamma;
      // ---- channel LayerNorm of the conv's result in this epilogue (reference attentions.py:103-120: x = norm_layers_1(x + y), x =
      // norm_layers_2(x + y); modules.LayerNorm = F.layer_norm over channels, eps 1e-5, biased variance).  The workgroup owns ALL cout
      // channels of its columns (launcher: one 32-row tile per wave, cout = 32 * waves), so the two passes (mean, centred sum of squares)
      // are a 16-register sum per lane, one cross-half shuffle and a WN-entry LDS exchange each — instead of writing s, a launch of its
      // own and s read back (7.7 us per LayerNorm at B = 32).  out may be the residual's tensor: every lane has read its residual values.
      const float* const lnb = P.ln_beta;
      const float eps = P.ln_eps, invc = 1.f / (float)cout;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[ni][r] + bs[r];
          if (act == ACT_RELU) v = fmaxf(v, 0.f);
          if (mask_pre) v *= om[ni];
          if (res_mode == RES_ADD) v += rv[ni][r];
          else if (res_mode == RES_RSUB) v = rv[ni][r] - v;
          if (mask_post) v *= om[ni];
          acc[ni][r] = v;
        }
      __syncthreads();                             // every wave is done with the tile (K split: with the merge buffer): LDS is free
      float* lnr = reinterpret_cast<float*>(xs);   // [2][WN][NI][32]
      float mean[NI], rstd[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        float sacc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += acc[ni][r];
        sacc += __shfl_xor(sacc, 32);
        if (lh == 0) lnr[(wm * NI + ni) * 32 + l31] = sacc;
      }
      __syncthreads();
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) t += lnr[(w * NI + ni) * 32 + l31];
        mean[ni] = t * invc;
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = acc[ni][r] - mean[ni]; q += d * d; }
        q += __shfl_xor(q, 32);
        if (lh == 0) lnr[WN * NI * 32 + (wm * NI + ni) * 32 + l31] = q;
      }
      __syncthreads();
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) t += lnr[WN * NI * 32 + (w * NI + ni) * 32 + l31];
        rstd[ni] = rsqrtf(t * invc + eps);
      }
      // scale / shift of this lane's 16 rows: loaded here (L2-hot, one exposed round trip), not in front of the reductions — with the
      // residual values and the accumulators live they did not fit the K-split variant's 168 registers
      float g16[16], b16[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        g16[r] = lng[co]; b16[r] = lnb[co];
      }
      // what layernorm.hip folds behind a LayerNorm-2: the speaker vector of the NEXT (conditioning) layer (attentions.py:103-109:
      // x = x + spk_emb_linear(g), then x * x_mask) — (y + vec[b][c]) * mask[b][t]
      if (P.ln_vec) {
        const float* const vp = P.ln_vec + (int64_t)b * P.ln_vec_bstride;
#pragma unroll
        for (int r = 0; r < 16; ++r) b16[r] += vp[mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
      }
      const float* const lmp = P.ln_mask ? P.ln_mask + (int64_t)b * P.out_mask_bstride : nullptr;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int t = t0 + ni * 32 + l31;
        if (t >= Lout) continue;
        const float lm = lmp ? lmp[t] : 1.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          outp[(unsigned)co * o_rs + (unsigned)t] = ((acc[ni][r] - mean[ni]) * rstd[ni] * g16[r] + b16[r]) * lm;
        }
      }
    } else if (P.k16 && mt * 32 >= P.kv_row0 && mt * 32 < P.kv_row0 + 2 * P.kv_rows) {
      // K / V rows of the fused q/k/v projection: fp16 for attention.hip's KV16 form (wave-uniform branch: a 32-row tile is K or V as a whole).
      // No activation / residual / mask on these rows (attentions.py:263-266: k = conv_k(c), v = conv_v(c)).
      const int kr = P.kv_rows, ldk = P.k16_ld;
      const bool is_k = mt * 32 < P.kv_row0 + kr;
      const int c0 = mt * 32 - P.kv_row0 - (is_k ? 0 : kr);          // first channel of this tile inside K (or V)
      if (is_k) {
        uint16_t* const kp = P.k16 + (int64_t)b * ldk * kr;           // [ld][kv_rows]: this lane's 4 consecutive channels are one 8-byte store
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int t = t0 + ni * 32 + l31;
          if (t >= Lout) continue;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            u32x2 o;
            o.x = h_pack(acc[ni][4 * g] + bs[4 * g], acc[ni][4 * g + 1] + bs[4 * g + 1]);
            o.y = h_pack(acc[ni][4 * g + 2] + bs[4 * g + 2], acc[ni][4 * g + 3] + bs[4 * g + 3]);
            *reinterpret_cast<u32x2*>(kp + (int64_t)t * kr + c0 + 8 * g + 4 * lh) = o;
          }
        }
      } else {
        _Float16* const vp = reinterpret_cast<_Float16*>(P.v16) + (int64_t)b * kr * ldk;   // [kv_rows][ld]: 32 lanes = 64 contiguous bytes of a row
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int t = t0 + ni * 32 + l31;
          if (t >= Lout) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            vp[(unsigned)c * (unsigned)ldk + (unsigned)t] = (_Float16)(acc[ni][r] + bs[r]);
          }
        }
      }
    } else {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int t = t0 + ni * 32 + l31;
      if (t >= Lout) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co >= cout) continue;
        float v = acc[ni][r] + bs[r];
        if (act == ACT_RELU) v = fmaxf(v, 0.f);
        if (mask_pre) v *= om[ni];
        if (res_mode == RES_ADD) v += rv[ni][r];
        else if (res_mode == RES_RSUB) v = rv[ni][r] - v;
        if (mask_post) v *= om[ni];
        outp[(unsigned)co * o_rs + (unsigned)t] = v;
      }
    }
    }
  } else if (P.act == ACT_GATE) {
    // fused_add_tanh_sigmoid_multiply (commons.py:98-105) on gate-ordered rows: registers r and r + 8 of a lane are the (tanh,
    // sigmoid) pair of output channel 16*mt + 8*(r>>2) + 4*lh + (r&3), r < 8; bias and the per-batch conditioning slice in fp32
    uint16_t* outp = static_cast<uint16_t*>(P.out) + (int64_t)b * P.out_bstride;
    const int couth = cout >> 1;
    const float* const b2 = P.bias2 ? P.bias2 + (int64_t)b * P.bias2_bstride : nullptr;
    float bt[2][4], bsg[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int rt = mt * 32 + 8 * g + 4 * lh + i;
        rt = rt + 16 < cout ? rt : 0;
        bt[g][i] = (P.bias ? P.bias[rt] : 0.f) + (b2 ? b2[rt] : 0.f);
        bsg[g][i] = (P.bias ? P.bias[rt + 16] : 0.f) + (b2 ? b2[rt + 16] : 0.f);
      }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int t = t0 + ni * 32 + l31;
      if (t >= L.L) continue;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int co = mt * 16 + 8 * g + 4 * lh;
        if (co >= couth) continue;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float a = acc[ni][4 * g + i] + bt[g][i], sg = acc[ni][4 * g + 8 + i] + bsg[g][i];
          v[i] = tanhf(a) * (1.f / (1.f + expf(-sg)));
        }
        u32x2 o;
        o.x = h_pack(v[0], v[1]); o.y = h_pack(v[2], v[3]);
        *reinterpret_cast<u32x2*>(outp + (int64_t)t * couth + co) = o;
      }
    }
  } else {
    // every problem field, the four bias float4 and the NI mask values into registers BEFORE the first store (after a global store
    // the compiler re-loads P.* from the kernarg segment and, in the loop form, waited for each bias load: (ni, g) serial round trips)
    uint16_t* outp = static_cast<uint16_t*>(P.out) + (int64_t)b * P.out_bstride;
    const float* const biasp = P.bias;
    const float* const omp = P.out_mask ? P.out_mask + (int64_t)b * P.out_mask_bstride : nullptr;
    const bool relu = P.act == ACT_RELU, masked = P.mask_pre || P.mask_post;
    const int Lout = L.L;
    f32x4 bvv[4];
    float omv[NI];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int co = mt * 32 + 8 * g + 4 * lh;
      co = co + 4 <= cout ? co : 0;
      bvv[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (biasp) bvv[g] = *reinterpret_cast<const f32x4*>(biasp + co);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int t = t0 + ni * 32 + l31;
      omv[ni] = omp ? omp[t < Lout ? t : Lout - 1] : 1.f;
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int t = t0 + ni * 32 + l31;
      if (t >= Lout) continue;
      const float om = omv[ni];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = mt * 32 + 8 * g + 4 * lh;
        if (co >= cout) continue;
        float v0 = acc[ni][4 * g], v1 = acc[ni][4 * g + 1], v2 = acc[ni][4 * g + 2], v3 = acc[ni][4 * g + 3];
        v0 += bvv[g].x; v1 += bvv[g].y; v2 += bvv[g].z; v3 += bvv[g].w;
        if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        if (masked) { v0 *= om; v1 *= om; v2 *= om; v3 *= om; }
        u32x2 o;
        o.x = h_pack(v0, v1); o.y = h_pack(v2, v3);
        *reinterpret_cast<u32x2*>(outp + (int64_t)t * cout + co) = o;
      }
    }
  }
  if (L.dbg && tid == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long* d = L.dbg + 8ull * ((unsigned long long)blockIdx.y * gridDim.x + blockIdx.x);
    d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = __builtin_amdgcn_s_memtime();
    d[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    d[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    d[6] = (unsigned long long)k; d[7] = 1;
  }
}

// the LayerNorm epilogue needs the workgroup to own every output channel: one 32-row tile per wave and no idle wave (the epilogue's
// barriers are workgroup-wide); instantiated for the Encoder stacks' hidden width
bool conv_f16_ln_supported(int cout) { return cout == 192; }   // six waves x 64 columns: the instantiations below

bool conv_f16_supported(int cin, int cout, int k, int dil, bool out_cl) {
  if (cin < 16 || cin % 16 || cout < 1 || k < 1 || dil < 1) return false;
  if (out_cl && cout % 4) return false;
  const int ck = cin < HC_CK ? cin : HC_CK;
  return (int64_t)(64 + (k - 1) * dil) * (ck + 8) * 2 <= 160 * 1024;
}

template <int WN, int NI, bool IN_CT, bool OUT_CT, int G, bool LN = false>
static int launch_hc_g(hipStream_t stream, const HcLaunch& L, int nt) {
  constexpr int BT = 32 * NI;
  const HcProb& p = L.p[0];
  const int ck = p.cin < HC_CK ? p.cin : HC_CK;
  const size_t lds = (size_t)(BT + (p.k - 1) * p.dil) * (size_t)(ck + 8) * 2;
  if (lds > 160 * 1024) return -2;
  const int ngrp = (nt + WN - 1) / WN;
  dim3 grid((L.L + BT - 1) / BT, L.B * ngrp, L.nprob);
  auto kern = conv_f16_kernel<WN, NI, IN_CT, OUT_CT, G, 1, LN>;
  ensure_dyn_lds((const void*)kern, lds);
  HcLaunch Lt = L;
  if (L.xcd_b) {
    Lt.xcd_gx = (int)grid.x; Lt.xcd_per = (int)grid.x * ngrp * L.nprob;
    grid = dim3(xcd_grid(L.B, Lt.xcd_per), 1, 1);
  }
  Lt.dbg = L.nprob > 1 ? nullptr : timeline_slice(grid.x, grid.y, 1, 99000 + WN * 100 + NI * 10 + (IN_CT ? 2 : 0) + (OUT_CT ? 1 : 0), p.k, p.cin, L.L);   // 99xxx: fp16 conv
  hipLaunchKernelGGL(kern, grid, dim3(64 * WN), lds, stream, Lt, ngrp);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// the in-workgroup K split (KS waves per output tile; C_in = KS*16*G staged whole)
template <int WN, int NI, int G, int KS, bool LN = false>
static int launch_hc_ks(hipStream_t stream, const HcLaunch& L, int nt) {
  constexpr int BT = 32 * NI;
  const HcProb& p = L.p[0];
  if (p.cin != KS * 16 * G || L.nprob != 1) return -2;
  const size_t lds_tile = (size_t)(BT + (p.k - 1) * p.dil) * (size_t)(p.cin + 8) * 2;
  const size_t lds_red = (size_t)(KS - 1) * WN * NI * 16 * 64 * 4;
  const size_t lds = lds_tile > lds_red ? lds_tile : lds_red;
  if (lds > 160 * 1024) return -2;
  const int ngrp = (nt + WN - 1) / WN;
  dim3 grid((L.L + BT - 1) / BT, L.B * ngrp, 1);
  auto kern = conv_f16_kernel<WN, NI, false, true, G, KS, LN>;
  ensure_dyn_lds((const void*)kern, lds);
  HcLaunch Lt = L;
  if (L.xcd_b) {
    Lt.xcd_gx = (int)grid.x; Lt.xcd_per = (int)grid.x * ngrp;
    grid = dim3(xcd_grid(L.B, Lt.xcd_per), 1, 1);
  }
  Lt.dbg = timeline_slice(grid.x, grid.y, 1, 99000 + WN * 100 + NI * 10 + 1 + 4, p.k, p.cin, L.L);   // 99xx5: K split
  hipLaunchKernelGGL(kern, grid, dim3(64 * WN * KS), lds, stream, Lt, ngrp);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

static bool g_hc_generic = false;     // tests / tuning only (bv2_test_set_variants): the generic GEMM loop instead of the C_in-specialised one
void conv_f16_set_tuning(int generic) { g_hc_generic = generic != 0; }

template <int WN, int NI, bool IN_CT, bool OUT_CT, bool LN = false>
static int launch_hc(hipStream_t stream, const HcLaunch& L, int nt) {
  const bool generic = g_hc_generic;
  if (!generic && L.p[0].cin == 192) return launch_hc_g<WN, NI, IN_CT, OUT_CT, 12, LN>(stream, L, nt);
  if (!generic && L.p[0].cin % 256 == 0) return launch_hc_g<WN, NI, IN_CT, OUT_CT, 16, LN>(stream, L, nt);
  return launch_hc_g<WN, NI, IN_CT, OUT_CT, 0, LN>(stream, L, nt);
}

template <int WN, int NI>
static int launch_hc_io(hipStream_t stream, const HcLaunch& L, int nt) {
  const bool ict = L.p[0].in_ct != 0, oct = L.p[0].out_ct != 0;
  if (ict && oct) return launch_hc<WN, NI, true, true>(stream, L, nt);
  if (ict) return launch_hc<WN, NI, true, false>(stream, L, nt);
  if (oct) return launch_hc<WN, NI, false, true>(stream, L, nt);
  return launch_hc<WN, NI, false, false>(stream, L, nt);
}

int launch_conv_f16(hipStream_t stream, const HcLaunch& L, const char** variant_name) {
  const HcProb& p = L.p[0];
  if (L.B < 1 || L.L < 1 || !conv_f16_supported(p.cin, p.cout, p.k, p.dil, !p.out_ct) || p.cout_pad % 32 || p.cout_pad < p.cout)
    return -1;
  if (L.nprob < 1 || L.nprob > 2) return -1;
  for (int i = 0; i < L.nprob; ++i) {
    const HcProb& q = L.p[i];
    if (q.cin != p.cin || q.k != p.k || q.dil != p.dil || q.cout_pad != p.cout_pad || q.in_ct != p.in_ct || q.out_ct != p.out_ct ||
        q.pad_left != p.pad_left || q.cout > q.cout_pad)
      return -1;
    if (!q.out_ct && (q.res_mode != RES_NONE)) return -1;          // the residual add lives in the fp32 [C][T] epilogue
    if (q.act == ACT_GATE && (q.out_ct || q.cout % 32)) return -1; // the gate writes fp16 channels-last, whole (tanh, sigmoid) tiles
    if (q.bias2 && q.act != ACT_GATE) return -1;                   // the per-batch bias only exists in the gate epilogue
    if (q.k16 && (!q.v16 || !q.out_ct || q.ln_gamma || q.kv_row0 % 32 || q.kv_rows % 32 || q.kv_rows < 32 || q.k16_ld < L.L ||
                  q.kv_row0 + 2 * q.kv_rows > q.cout || q.res_mode != RES_NONE || q.act != ACT_NONE || q.out_mask)) return -1;
    if (q.ln_gamma && (!conv_f16_ln_supported(q.cout) || !q.out_ct || !q.ln_beta || q.cout_pad != q.cout || L.nprob != 1)) return -1;
  }
  const int nt = p.cout_pad / 32;
  // waves per workgroup (each owns one 32-channel output tile): the count in {8, 6, 4} that wastes the fewest wave slots
  int wn = 8, best = 1 << 30;
  const int cand[3] = {8, 6, 4};
  for (int c : cand) {
    const int waste = (nt + c - 1) / c * c - nt;
    if (waste < best || (waste == best && c == L.wn_pref)) { best = waste; wn = c; }
  }
  // time steps per workgroup: 128 when that still gives every CU a workgroup, else 64
  const long wg128 = (long)((L.L + 127) / 128) * L.B * ((nt + wn - 1) / wn);
  int ni = wg128 >= 256 ? 4 : 2;
  if (L.ni_pref == 2 || L.ni_pref == 4) ni = L.ni_pref;
  {
    const int ck = p.cin < HC_CK ? p.cin : HC_CK;
    if ((int64_t)(128 + (p.k - 1) * p.dil) * (ck + 8) * 2 > 160 * 1024) ni = 2;
  }
  static const char* names[3][2] = {{"conv_f16<8w,64>", "conv_f16<8w,128>"}, {"conv_f16<6w,64>", "conv_f16<6w,128>"},
                                    {"conv_f16<4w,64>", "conv_f16<4w,128>"}};
  if (variant_name) *variant_name = names[wn == 8 ? 0 : (wn == 6 ? 1 : 2)][ni == 4 ? 1 : 0];
  const bool ks_shape = wn == 6 && nt == 6 && p.cin == 768 && !p.in_ct && p.out_ct && L.nprob == 1 && !L.no_ksplit && !g_hc_generic;
  if (p.ln_gamma) {                               // LayerNorm epilogue: 6 waves x 64 columns (the 128-column tile has no registers left for it)
    if (wn != 6 || nt != 6) return -1;
    int r = -2;
    if (ks_shape) r = launch_hc_ks<6, 2, 24, 2, true>(stream, L, nt);
    if (r != -2) {
      if (variant_name) *variant_name = "conv_f16<6w x 2k,64,ln>";
      return r;
    }
    if (variant_name) *variant_name = "conv_f16<6w,64,ln>";
    return p.in_ct ? launch_hc<6, 2, true, true, true>(stream, L, nt) : launch_hc<6, 2, false, true, true>(stream, L, nt);
  }
  // FFN conv_2 (768 -> 192 rows) on 64-column tiles: K halves inside the workgroup (12 waves)
  if (ks_shape && ni == 2) {
    const int r = launch_hc_ks<6, 2, 24, 2>(stream, L, nt);
    if (r != -2) {
      if (variant_name) *variant_name = "conv_f16<6w x 2k,64>";
      return r;
    }
  }
  if (wn == 8) return ni == 4 ? launch_hc_io<8, 4>(stream, L, nt) : launch_hc_io<8, 2>(stream, L, nt);
  if (wn == 6) return ni == 4 ? launch_hc_io<6, 4>(stream, L, nt) : launch_hc_io<6, 2>(stream, L, nt);
  return ni == 4 ? launch_hc_io<4, 4>(stream, L, nt) : launch_hc_io<4, 2>(stream, L, nt);
}

double conv_f16_flops(const HcLaunch& L) {
  double f = 0;
  for (int i = 0; i < L.nprob; ++i) f += 2.0 * L.p[i].cout * L.p[i].cin * L.p[i].k * (double)L.L * L.B;
  return f;
}

double conv_f16_bytes(const HcLaunch& L) {   // input read once, output written once (+ residual read), weights once
  const double n = (double)L.L * L.B;
  double by = (L.p[0].in_ct ? 4.0 : 2.0) * L.p[0].cin * n;           // the problems of a launch share their input
  for (int i = 0; i < L.nprob; ++i) {
    const HcProb& p = L.p[i];
    const double co = p.act == ACT_GATE ? 0.5 * p.cout : (double)p.cout;
    by += (p.out_ct ? 4.0 : 2.0) * co * n + (p.res_mode ? 4.0 * p.cout * n : 0.0) + 2.0 * p.cout * p.cin * p.k;
  }
  return by;
}

}  // namespace bv2

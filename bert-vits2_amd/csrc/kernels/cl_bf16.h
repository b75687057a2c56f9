// cl_bf16.h — device helpers shared by the bf16 channels-last Generator kernels (gen_bf16.hip: one conv per launch;
// respair_cl_bf16.hip: a (dilated conv, conv) pair of ResBlock1 per launch): tile staging HBM -> LDS and the two GEMM loops
// LDS (B operand) x global weight-fragment stream (A operand) on v_mfma_f32_32x32x16_bf16.
#pragma once
#include <hip/hip_runtime.h>
#include "../bv2_kernels.h"

namespace bv2 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned bf_pack(float a, float b) {     // round-to-nearest-even (v_cvt_pk_bf16_f32)
  bf16x2 r;
  r[0] = (__bf16)a; r[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, r);
}

__device__ __forceinline__ float bf_round(float v) { return (float)(__bf16)v; }     // round-to-nearest-even to a bf16 value

// The stage hand-over's rounding points (oracle generator_bf16; reference models.py:545-552 `xs / self.num_kernels`): a stage's n <= 3 branch
// outputs r_j (each already bf16) are summed widest kernel first — branch n-1, n-2, ... 0 — with the running sum rounded to bf16 wherever the
// summed-output kernels store it, and the mean is bf16((running sum + r_0) * fp32(1/n)).  Both forms of the hand-over compute exactly this:
// the producer that writes one tensor (RpClLaunch / RbClLaunch sum_out) and the consumer that reads the n branch tensors (x0 = branch 0).
__device__ __forceinline__ float stage_mean(float x0, float x1, float x2, int nsrc, float scale) {
  if (nsrc <= 1) return x0;
  const float s = nsrc > 2 ? bf_round(x2 + x1) + x0 : x1 + x0;
  return bf_round(s * scale);
}
// the producer side of the same sum, one 16-byte piece (8 bf16) at a time: prev = the running sum, r = this branch's output
__device__ __forceinline__ u32x4 stage_accum(u32x4 r, u32x4 prev, bool last, float scale) {
  u32x4 o;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    float lo = bf_lo(prev[w]) + bf_lo(r[w]), hi = bf_hi(prev[w]) + bf_hi(r[w]);
    if (last) { lo *= scale; hi *= scale; }
    o[w] = bf_pack(lo, hi);
  }
  return o;
}

constexpr int CL_PD = 8;          // weight prefetch ring depth (units of 4 MFMAs)

// Stage rows [tb, tb + rows) x cin channels of up to 3 sources into LDS (pitch in elements), applying
// pre(v) = bf16(lrelu(stage_mean(sources))).  Rows outside [0, Lin) are zero (the conv's padding).
// Every load of a batch is issued before the first one is used (QB pieces of 16 B per thread in flight): tools/timeline.py showed
// the staging at three SERIAL global round trips of ~4.6k cycles each with batches of 4 — as long as the k = 11 GEMM it feeds.
template <int NT, int QB, bool MULTI>
__device__ __forceinline__ void cl_stage_impl(unsigned short* xs, int pitch, const uint16_t* s0, const uint16_t* s1,
                                              const uint16_t* s2, int nsrc, float in_scale, bool lrelu, float slope, int tb,
                                              int rows, int cin, int Lin, int tid) {
  const int ppr = cin >> 3;                       // 16-byte pieces per row
  const int total = rows * ppr;
  const bool raw = nsrc == 1 && !lrelu;
  for (int base = 0; base < total; base += QB * NT) {
    u32x4 v[QB][MULTI ? 3 : 1];
    int dst[QB];
    bool ok[QB], inb[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      int p = base + q * NT + tid;
      inb[q] = p < total;
      p = inb[q] ? p : total - 1;
      const int r = p / ppr, cb = p - r * ppr;
      const int t = tb + r;
      ok[q] = inb[q] && t >= 0 && t < Lin;
      const int tc = t < 0 ? 0 : (t >= Lin ? Lin - 1 : t);           // clamped: the loads are unconditional
      const int64_t off = (int64_t)tc * cin + cb * 8;
      dst[q] = r * pitch + cb * 8;
      v[q][0] = *reinterpret_cast<const u32x4*>(s0 + off);
      if (MULTI) {
        if (nsrc > 1) v[q][1] = *reinterpret_cast<const u32x4*>(s1 + off);
        if (nsrc > 2) v[q][2] = *reinterpret_cast<const u32x4*>(s2 + off);
      }
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      if (!inb[q]) continue;
      u32x4 o;
      if (raw) {
        o = v[q][0];
      } else {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          float a = bf_lo(v[q][0][w]), b = bf_hi(v[q][0][w]);
          if (MULTI) {                            // the branch mean, rounded where the summed-output producers round it (stage_mean)
            a = stage_mean(a, nsrc > 1 ? bf_lo(v[q][1][w]) : 0.f, nsrc > 2 ? bf_lo(v[q][2][w]) : 0.f, nsrc, in_scale);
            b = stage_mean(b, nsrc > 1 ? bf_hi(v[q][1][w]) : 0.f, nsrc > 2 ? bf_hi(v[q][2][w]) : 0.f, nsrc, in_scale);
          }
          if (lrelu) { a = a < 0.f ? a * slope : a; b = b < 0.f ? b * slope : b; }
          o[w] = bf_pack(a, b);
        }
      }
      if (!ok[q]) o = u32x4{0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(xs + dst[q]) = o;
    }
  }
}

// Stage rows [tb, tb + rows) x cin channels of up to 3 sources into LDS (pitch in elements), applying
// pre(v) = bf16(lrelu(in_scale * sum)).  Rows outside [0, Lin) are zero (the conv's padding).
template <int NT>
__device__ __forceinline__ void cl_stage(unsigned short* xs, int pitch, const uint16_t* s0, const uint16_t* s1,
                                         const uint16_t* s2, int nsrc, float in_scale, bool lrelu, float slope, int tb,
                                         int rows, int cin, int Lin, int tid) {
  if (nsrc == 1) cl_stage_impl<NT, 12, false>(xs, pitch, s0, s1, s2, nsrc, in_scale, lrelu, slope, tb, rows, cin, Lin, tid);
  else cl_stage_impl<NT, 6, true>(xs, pitch, s0, s1, s2, nsrc, in_scale, lrelu, slope, tb, rows, cin, Lin, tid);
}

// acc[mi][ni] += sum over units u = (s, j) of Wfrag(mi, u) x B(u, ni);  B(u, ni) = 8 channels [16s + 8lh, +8) of LDS row
// (ni*32 + l31 + j*tstep) relative to xb.  wp points at the wave's FIRST m-tile's contiguous weight stream (+ lane*8
// elements); m-tile mi's stream starts mstride elements later.  Register blocking MI x NI: every B fragment read from LDS
// feeds MI MFMAs and every A fragment NI MFMAs (at MI = 1 the B reads alone need the full 128 B/clk LDS bandwidth at the
// MFMA issue rate).
template <int MI, int NI, int PD>
__device__ __forceinline__ void cl_gemm(f32x16 (&acc)[MI][NI], const uint16_t* wp, int64_t mstride, int U, int k,
                                        const unsigned short* xb, int pitch, int tstep, int kfull) {
  // k taps of a stream laid out for kfull >= k taps per group (wp points at the first tap that is run): U = groups * k units
  bf16x8 ar[PD][MI];
  int lu = 0, lj = 0;
  int64_t loff = 0;
  const int64_t gskip = (int64_t)(kfull - k) * 512;
  auto load_unit = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) ar[slot][mi] = *reinterpret_cast<const bf16x8*>(wp + mi * mstride + loff);
    if (++lu < U) {                                                 // past the end: re-read the last unit, result unused
      loff += 512;
      if (++lj == k) { lj = 0; loff += gskip; }
    }
  };
#pragma unroll
  for (int i = 0; i < PD; ++i) { load_unit(i); __builtin_amdgcn_sched_barrier(0); }
  bf16x8 bb[2][NI];                               // B fragments of the current / next unit (parity of the ring slot)
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) bb[0][ni] = *reinterpret_cast<const bf16x8*>(xb + ni * 32 * pitch);
  int s = 0, j = 0;
  for (int u0 = 0; u0 < U; u0 += PD) {
#pragma unroll
    for (int i = 0; i < PD; ++i) {
      if (u0 + i < U) {
        int jn = j + 1, sn = s;
        if (jn == k) { jn = 0; ++sn; }
        const bool more = u0 + i + 1 < U;
        const unsigned short* xn = xb + (more ? jn : j) * tstep * pitch + (more ? sn : s) * 16;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bb[(i & 1) ^ 1][ni] = *reinterpret_cast<const bf16x8*>(xn + ni * 32 * pitch);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[i][mi], bb[i & 1][ni], acc[mi][ni], 0, 0, 0);
        j = jn; s = sn;
      }
      load_unit(i);
      // pin the emitted order: next unit's LDS reads first (they land under this unit's MFMAs), MFMAs, ring loads
      __builtin_amdgcn_sched_group_barrier(0x100, NI, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MI * NI, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, MI, 0);
      __builtin_amdgcn_sched_barrier(0);          // keep program order: the ring's vmcnt distances stay PD - 1 units
    }
  }
}

// Tap-major form of the same product for C_in = 16*G known at compile time (G = 4 / 8 / 16: the ResBlock convs of the wide
// stages).  The generic loop above spends ~9 scalar / vector instructions per MFMA on unit bookkeeping (tap wrap, LDS
// address arithmetic with runtime pitch, end-of-stream clamps, a branch per unit) — more than a 32-cycle bf16 MFMA hides
// from one wave.  Here the ring has one slot per 16-channel group (slot s: unit (s, tap j) is followed by (s, j+1)), the
// groups of a tap are a fully unrolled inner loop, every LDS offset is an immediate (pitch = 16*G + 8 is a constant) and
// a tap costs one pointer add: NI ds_read_b128 + NI MFMAs + 1 global load + 2 scalar adds per unit.
// explicit global address space: pointers kept in an array and advanced in a loop defeat the address-space inference, and a
// FLAT load counts on lgkmcnt as well — every LDS wait would then also wait for the weight ring
typedef __attribute__((address_space(1))) bf16x8 GlobalFrag;
template <int MI, int NI, int G>
__device__ __forceinline__ void cl_gemm_tm(f32x16 (&acc)[MI][NI], const uint16_t* wbase, int64_t mstride, unsigned wlane_bytes,
                                           int k, const unsigned short* xb, int tstep, int kfull) {   // wbase: wave-uniform stream start
  // k taps of a stream laid out for kfull >= k taps per group (wbase = the first tap that is run)
  constexpr int PITCH = 16 * G + 8;
  static_assert(G % 2 == 0, "the B double buffer alternates with the group index");
  bf16x8 ar[G][MI];
  const uint16_t* wq[G][MI];
  const int first_step = k > 1 ? 512 : 0;
#pragma unroll
  for (int s = 0; s < G; ++s) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      wq[s][mi] = wbase + mi * mstride + (int64_t)s * kfull * 512;
      ar[s][mi] = *(const GlobalFrag*)(reinterpret_cast<const char*>(wq[s][mi]) + wlane_bytes);
      wq[s][mi] += first_step;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // (Round 3: reading the B operands TWO units ahead — four rotating buffers — was measured in a same-box A/B of builds: <4x1> 4.03 ->
  // 4.05 ms, <8x1> 2.15 -> 2.17 ms per step.  The LDS latency is not what keeps the GEMM phase at 1.8x its MFMA-only time.)
  bf16x8 bb[2][NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) bb[0][ni] = *reinterpret_cast<const bf16x8*>(xb + ni * 32 * PITCH);
  const unsigned short* xrow = xb;
  for (int j = 0; j < k; ++j) {
    const unsigned short* xnext = (j + 1 < k) ? xrow + tstep * PITCH : xrow;   // after the last tap: re-read (unused)
    const int step = (j + 2 < k) ? 512 : 0;       // slot s is refilled with (s, j+1); the last tap's unit is not followed
#pragma unroll
    for (int s = 0; s < G; ++s) {
      const unsigned short* xn = (s + 1 < G) ? xrow + (s + 1) * 16 : xnext;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bb[(s & 1) ^ 1][ni] = *reinterpret_cast<const bf16x8*>(xn + ni * 32 * PITCH);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[s][mi], bb[s & 1][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        ar[s][mi] = *(const GlobalFrag*)(reinterpret_cast<const char*>(wq[s][mi]) + wlane_bytes);
        wq[s][mi] += step;
      }
      __builtin_amdgcn_sched_group_barrier(0x100, NI, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MI * NI, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, MI, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    xrow = xnext;
  }
}

// The same tap-major product with the ring priming split from the loop (one m-tile per wave), for kernels that run several GEMMs per
// workgroup (respair_cl_bf16.hip): the G loads of the NEXT GEMM's first units are issued before the phase that precedes it (tile
// staging, the epilogue of the previous GEMM, a barrier) and land under it, instead of behind it with the matrix pipe idle.
template <int G>
__device__ __forceinline__ void cl_tm_prime(bf16x8 (&ar)[G], const uint16_t* (&wq)[G], const uint16_t* wbase, unsigned wlane_bytes, int k) {
  const int first_step = k > 1 ? 512 : 0;
#pragma unroll
  for (int s = 0; s < G; ++s) {
    wq[s] = wbase + (int64_t)s * k * 512;
    ar[s] = *(const GlobalFrag*)(reinterpret_cast<const char*>(wq[s]) + wlane_bytes);
    wq[s] += first_step;
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int NI, int G>
__device__ __forceinline__ void cl_tm_run(f32x16 (&acc)[NI], bf16x8 (&ar)[G], const uint16_t* (&wq)[G], unsigned wlane_bytes, int k,
                                          const unsigned short* xb, int tstep) {
  constexpr int PITCH = 16 * G + 8;
  static_assert(G % 2 == 0, "the B double buffer alternates with the group index");
  bf16x8 bb[2][NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) bb[0][ni] = *reinterpret_cast<const bf16x8*>(xb + ni * 32 * PITCH);
  const unsigned short* xrow = xb;
  for (int j = 0; j < k; ++j) {
    const unsigned short* xnext = (j + 1 < k) ? xrow + tstep * PITCH : xrow;   // after the last tap: re-read (unused)
    const int step = (j + 2 < k) ? 512 : 0;       // slot s is refilled with (s, j+1); the last tap's unit is not followed
#pragma unroll
    for (int s = 0; s < G; ++s) {
      const unsigned short* xn = (s + 1 < G) ? xrow + (s + 1) * 16 : xnext;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bb[(s & 1) ^ 1][ni] = *reinterpret_cast<const bf16x8*>(xn + ni * 32 * PITCH);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[s], bb[s & 1][ni], acc[ni], 0, 0, 0);
      ar[s] = *(const GlobalFrag*)(reinterpret_cast<const char*>(wq[s]) + wlane_bytes);
      wq[s] += step;
      __builtin_amdgcn_sched_group_barrier(0x100, NI, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NI, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    xrow = xnext;
  }
}

}  // namespace bv2

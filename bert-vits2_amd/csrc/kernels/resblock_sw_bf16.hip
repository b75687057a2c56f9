// resblock_sw_bf16.hip — a WHOLE HiFi-GAN ResBlock1 (reference modules.py:296-309: for each dilation d,
// x = x + conv2(lrelu(conv1(lrelu(x), k, d)), k, 1)) of the C = 64 / C = 32 bf16 Generator stages in ONE kernel, channels-last,
// on v_mfma_f32_32x32x16_bf16 (round 5).
//
// Why.  Pair by pair (respair_cl_bf16.hip) these stages move six tensor passes per branch: the C = 64 launches run at MFMA-busy 0.40 with
// "stage + epilogue half of a workgroup's life" (0.30 of the bf16 peak, 2.17 ms per step at B = 32), the C = 32 launches are HBM-bound at
// 3.2 TB/s (1.36 ms).  resblock_c16_bf16.hip showed what the whole block in one kernel buys once the LDS tile is unpadded: two passes.
// Here the same structure at 64 / 128 bytes per row:
//   * the tile is [rows][C] bf16 with NO row padding; 16-byte piece p of LDS row r sits at p ^ f(r), f(r) = (r >> 2) & 3 (64-byte rows) or
//     (r >> 1) & 7 (128-byte rows): the four hardware lane groups of a ds_read_b128 ({0-3, 12-15, 20-27}, ... — MI355X_MICROARCH.md §LDS)
//     then touch 16 distinct 16-byte slots of the 256-byte bank row for any row offset, and f is invariant under r -> r + 32, so the NB
//     row blocks of a wave share one address register per (tap, group) and differ by immediates;
//   * a wave owns ONE 32-channel output tile x NB blocks of 32 rows (C = 64: 2 x 4 waves, NB = 4; C = 32: 1 x 8 waves, NB = 2): every
//     weight fragment feeds NB MFMAs; the residual stays in registers (packed, exact bf16 values) across the three pairs.  (First
//     form, C = 64: a wave owned both channel tiles of 64 rows — every tap pulled 8 KB of fragments per wave, 64 KB per workgroup,
//     through the CU's L1 at 64 B/clk = as long as the tap's MFMAs: 2.88 ms per step against 2.17 pair by pair.)
//   * weights: ONE contiguous stream per branch, [conv][tap][group][m-tile][lane][8] — a tap's G fragments of the wave's tile are one
//     ring slot, DT taps deep, primed before the phase (epilogue + barrier) that precedes each GEMM;
//   * tiles of 512 rows: C = 64: 2 x (512 + 64) rows x 128 B = 144 KB, one 8-wave workgroup per CU (a tile's six GEMMs are 67k MFMA
//     cycles per CU at k = 11 against ~10k of staging and epilogues); C = 32: 2 x (512 + 64) x 64 B = 72 KB, TWO workgroups per CU, so
//     one's epilogues / barriers run under the other's GEMMs (<= 128 registers).
// MEASURED (profiles/r05_ab_resblock_sw_not_kept.txt, config 3, same box): C = 64 2.58 ms per step against 2.17 pair by pair, C = 32 1.44
// against 1.36 — config 3 15.34 -> 15.64 ms with both.  Two passes instead of six do not pay here: with one 32-channel tile per wave every
// MFMA needs its own ds_read_b128 (the operand mix tools/probe/mfma_bf16_probe.hip puts at 0.45-0.52 of the peak on real data, where
// the pair kernel's 64-channel x 128-row wave tiles reach 0.57-0.59), the 64 x 128 wave tile needs 128 accumulator registers AND a
// 144 KB tile pair — one 4-wave workgroup per CU — and the halo of three pairs (120 of 512 rows at k = 11) is recomputed.  OFF by
// default ("resblock_sw"); bit-identical to the pair kernels, so it stays as a second implementation the tests hold them to.
// Same unit order (tap-major, groups ascending) and the same rounding points as the pair kernels and the layer-wise bf16 path: the
// outputs are BIT-IDENTICAL to theirs (tests/test_resblock_sw_gpu.py).
#include <hip/hip_runtime.h>
#include "../bv2_kernels.h"

namespace bv2 {

typedef __bf16 swbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 swbf16x2 __attribute__((ext_vector_type(2)));
typedef float swf32x16 __attribute__((ext_vector_type(16)));
typedef float swf32x4 __attribute__((ext_vector_type(4)));
typedef unsigned swu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned swu32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) swbf16x8 SwGlobalFrag;   // explicit global address space: a FLAT load would also count on lgkmcnt

namespace {

__device__ __forceinline__ float sw_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float sw_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned sw_pack(float a, float b) {     // round-to-nearest-even (v_cvt_pk_bf16_f32)
  swbf16x2 r;
  r[0] = (__bf16)a; r[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float sw_lrelu(float v, float slope) { return v < 0.f ? v * slope : v; }
__device__ __forceinline__ unsigned sw_act(unsigned u, float slope) {   // bf16 pair -> bf16(lrelu(.)) pair
  return sw_pack(sw_lrelu(sw_lo(u), slope), sw_lrelu(sw_hi(u), slope));
}

constexpr int SW_G = 32;             // guard rows on each side of the LDS tiles (dilated taps reach <= 25 rows outside)
constexpr int SW_NW = 8;             // waves per workgroup

extern __shared__ __attribute__((aligned(16))) unsigned char sw_lds[];

template <int C> struct SwGeo {
  static constexpr int MTC = C / 32;                // 32-channel output tiles of the conv = waves side by side in channels
  static constexpr int MT = 1;                      // ... of which a wave owns one
  static constexpr int G = C / 16;                  // 16-channel groups = K steps per tap
  static constexpr int NB = C == 64 ? 4 : 2;        // 32-row blocks per wave
  static constexpr int DT = C == 64 ? 2 : 4;        // ring depth in taps
  static constexpr int R = 32 * NB * (SW_NW / MTC); // rows per tile incl. halo: 512 (C = 64: 2 x 4 waves; C = 32: 1 x 8 waves)
  static constexpr int ROWS = R + 2 * SW_G;
  static constexpr unsigned RB = 2u * C;            // bytes per LDS row
  static constexpr unsigned TILE = (unsigned)ROWS * RB;
  static constexpr int PPR = C / 8;                 // 16-byte pieces per row
  static constexpr int TAPF = G * MTC * 512;        // elements of one tap's fragments in the weight stream
};
// byte offset of 16-byte piece p of LDS row r inside a tile
template <int C> __device__ __forceinline__ unsigned sw_swz(unsigned r) { return C == 64 ? ((r >> 1) & 7u) : ((r >> 2) & 3u); }
template <int C> __device__ __forceinline__ unsigned sw_addr(unsigned r, unsigned p) { return r * SwGeo<C>::RB + ((p ^ sw_swz<C>(r)) << 4); }

template <int C>
struct SwRing {
  swbf16x8 a[SwGeo<C>::DT][SwGeo<C>::G][SwGeo<C>::MT];
};

// the fragments of tap j of conv `wconv` (wave-uniform stream pointer) into ring slot S
// (wl: lane * 16 + this wave's m-tile * 1024 — the byte offset of its fragment inside a (tap, group) block of the stream)
template <int C, int S>
__device__ __forceinline__ void sw_load_tap(SwRing<C>& ring, const uint16_t* wconv, int j, unsigned wl) {
  using Gm = SwGeo<C>;
  const char* base = reinterpret_cast<const char*>(wconv + (int64_t)j * Gm::TAPF) + wl;
#pragma unroll
  for (int g = 0; g < Gm::G; ++g) ring.a[S][g][0] = *(const SwGlobalFrag*)(base + g * Gm::MTC * 1024);
}
template <int C>
__device__ __forceinline__ void sw_prime(SwRing<C>& ring, const uint16_t* wconv, int k, unsigned wl) {
  using Gm = SwGeo<C>;
  sw_load_tap<C, 0>(ring, wconv, 0, wl);
  __builtin_amdgcn_sched_barrier(0);
  sw_load_tap<C, 1>(ring, wconv, k > 1 ? 1 : 0, wl);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (Gm::DT == 4) {
    sw_load_tap<C, 2>(ring, wconv, k > 2 ? 2 : 0, wl);
    __builtin_amdgcn_sched_barrier(0);
    sw_load_tap<C, 3>(ring, wconv, k > 3 ? 3 : 0, wl);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// one tap: acc[mt][nb] += sum_g W(j, g, mt) x B(row + j * dil, g, nb); refills slot S with tap j + DT (clamped: past the end the last tap
// is re-read, unused).  rowj: this lane's LDS row of the tap (block 0), src: byte offset of the tile read.
template <int C, int S>
__device__ __forceinline__ void sw_tap(swf32x16 (&acc)[SwGeo<C>::MT][SwGeo<C>::NB], SwRing<C>& ring, const uint16_t* wconv, int j, int k,
                                       unsigned wl, unsigned src, unsigned rowj, unsigned h) {
  using Gm = SwGeo<C>;
  const unsigned rbase = src + rowj * Gm::RB;
  const unsigned xv = sw_swz<C>(rowj) << 4;
  swbf16x8 bb[2][Gm::NB];
#pragma unroll
  for (int nb = 0; nb < Gm::NB; ++nb)
    bb[0][nb] = *reinterpret_cast<const swbf16x8*>(sw_lds + rbase + ((h << 4) ^ xv) + nb * 32 * Gm::RB);
#pragma unroll
  for (int g = 0; g < Gm::G; ++g) {
    if (g + 1 < Gm::G) {
      const unsigned pa = ((unsigned)((2 * (g + 1)) << 4) | (h << 4)) ^ xv;
#pragma unroll
      for (int nb = 0; nb < Gm::NB; ++nb)
        bb[(g & 1) ^ 1][nb] = *reinterpret_cast<const swbf16x8*>(sw_lds + rbase + pa + nb * 32 * Gm::RB);
    }
#pragma unroll
    for (int mt = 0; mt < Gm::MT; ++mt)
#pragma unroll
      for (int nb = 0; nb < Gm::NB; ++nb)
        acc[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring.a[S][g][mt], bb[g & 1][nb], acc[mt][nb], 0, 0, 0);
    if (g + 1 < Gm::G) __builtin_amdgcn_sched_group_barrier(0x100, Gm::NB, 0);   // the next group's LDS reads first: they land under these MFMAs
    __builtin_amdgcn_sched_group_barrier(0x008, Gm::MT * Gm::NB, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  const int jn = j + Gm::DT < k ? j + Gm::DT : k - 1;
  sw_load_tap<C, S>(ring, wconv, jn, wl);
  __builtin_amdgcn_sched_barrier(0);
}

// the whole GEMM of one conv over the wave's NB blocks (the ring is primed with taps 0 .. DT-1 of this conv)
template <int C>
__device__ __forceinline__ void sw_gemm(swf32x16 (&acc)[SwGeo<C>::MT][SwGeo<C>::NB], SwRing<C>& ring, const uint16_t* wconv, int k, int dil,
                                        unsigned wl, unsigned src, unsigned row_first, unsigned h) {
  using Gm = SwGeo<C>;
#pragma unroll
  for (int mt = 0; mt < Gm::MT; ++mt)
#pragma unroll
    for (int nb = 0; nb < Gm::NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nb][r] = 0.f;
  unsigned rowj = row_first;
  int j = 0;
  if constexpr (Gm::DT == 2) {                    // k odd: pairs of taps, then the last one on slot 0
    for (; j + 1 < k; j += 2) {
      sw_tap<C, 0>(acc, ring, wconv, j, k, wl, src, rowj, h);
      sw_tap<C, 1>(acc, ring, wconv, j + 1, k, wl, src, rowj + dil, h);
      rowj += 2 * dil;
    }
    sw_tap<C, 0>(acc, ring, wconv, j, k, wl, src, rowj, h);
  } else {                                        // k = 3 mod 4: quads of taps, then three on slots 0, 1, 2
    for (; j + 3 < k; j += 4) {
      sw_tap<C, 0>(acc, ring, wconv, j, k, wl, src, rowj, h);
      sw_tap<C, 1>(acc, ring, wconv, j + 1, k, wl, src, rowj + dil, h);
      sw_tap<C, 2>(acc, ring, wconv, j + 2, k, wl, src, rowj + 2 * dil, h);
      sw_tap<C, 3>(acc, ring, wconv, j + 3, k, wl, src, rowj + 3 * dil, h);
      rowj += 4 * dil;
    }
    sw_tap<C, 0>(acc, ring, wconv, j, k, wl, src, rowj, h);
    sw_tap<C, 1>(acc, ring, wconv, j + 1, k, wl, src, rowj + dil, h);
    sw_tap<C, 2>(acc, ring, wconv, j + 2, k, wl, src, rowj + 2 * dil, h);
  }
}

}  // namespace

template <int C>
__global__ void __launch_bounds__(64 * SW_NW, C == 64 ? 2 : 4) resblock_sw_bf16_kernel(const RbClLaunch L) {
  using Gm = SwGeo<C>;
  constexpr int NT = 64 * SW_NW, MT = Gm::MT, NB = Gm::NB, R = Gm::R;
  constexpr unsigned RB = Gm::RB, XA = 0u, TA = Gm::TILE;
  const RbClProb& P = L.p[blockIdx.z];
  const int TT = R - 2 * P.halo;                  // output rows per tile
  const int t0 = blockIdx.x * TT;
  if (t0 >= L.L) return;                          // branches with a smaller halo need fewer tiles
  const int b = blockIdx.y;
  int Lseq = L.L;
  if (L.lens) {
    const int64_t lv = L.lens[b] * L.len_mul;
    Lseq = lv < L.L ? (int)lv : L.L;
    if (t0 >= Lseq) return;                       // a tile wholly past the utterance: nobody reads its outputs
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned l31 = lane & 31, h = lane >> 5;
  const int wm = wid % Gm::MTC, wr = wid / Gm::MTC;    // this wave's 32-channel tile / group of NB row blocks
  const int tb = t0 - P.halo;                     // time step of tile row 0
  const int k = P.k, nd = L.nd, halfk = (k - 1) / 2;
  const float slope = L.slope;
  const unsigned wl = (unsigned)lane * 16u + (unsigned)wm * 1024u;
  const uint16_t* xg = P.x + (int64_t)b * L.L * C;
  uint16_t* outg = P.out + (int64_t)b * L.L * C;
  const int convw = k * Gm::TAPF;                 // elements of one conv in the stream

  SwRing<C> ring;
  sw_prime<C>(ring, P.w, k, wl);                  // conv 0's first taps go in flight before anything else

  // ---- stage the tile: raw rows -> TA (for the residual registers), bf16(lrelu) -> XA; rows outside [0, Lseq) and the guards are zero
  for (int p = tid; p < 2 * SW_G * Gm::PPR; p += NT) {
    const int gr = p / Gm::PPR, pc = p - gr * Gm::PPR;
    const unsigned row = gr < SW_G ? gr : R + gr;
    const unsigned a = sw_addr<C>(row, pc);
    *reinterpret_cast<swu32x4*>(sw_lds + XA + a) = swu32x4{0u, 0u, 0u, 0u};
    *reinterpret_cast<swu32x4*>(sw_lds + TA + a) = swu32x4{0u, 0u, 0u, 0u};
  }
  {
    constexpr int PIECES = R * Gm::PPR, QB = PIECES / NT;     // 16-byte pieces per thread: all in flight at once
    static_assert(PIECES % NT == 0, "whole pieces per thread");
    swu32x4 v[QB];
    bool ok[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int p = q * NT + tid;
      const int r = p / Gm::PPR, pc = p - r * Gm::PPR;
      const int t = tb + r;
      ok[q] = t >= 0 && t < Lseq;
      const int tc = t < 0 ? 0 : (t >= Lseq ? Lseq - 1 : t);
      v[q] = *reinterpret_cast<const swu32x4*>(xg + (unsigned)(tc * C + pc * 8));
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int p = q * NT + tid;
      const int r = p / Gm::PPR, pc = p - r * Gm::PPR;
      const swu32x4 raw = ok[q] ? v[q] : swu32x4{0u, 0u, 0u, 0u};
      swu32x4 act;
#pragma unroll
      for (int w = 0; w < 4; ++w) act[w] = sw_act(raw[w], slope);
      const unsigned a = sw_addr<C>((unsigned)(SW_G + r), (unsigned)pc);
      *reinterpret_cast<swu32x4*>(sw_lds + TA + a) = raw;
      *reinterpret_cast<swu32x4*>(sw_lds + XA + a) = act;
    }
  }
  __syncthreads();

  // ---- this wave's rows of x (the residual): lane = row l31 of block nb, channels 32 wm + 8 rg + 4 h + {0..3} = 8 bytes at piece 4 wm + rg
  const unsigned row0 = (unsigned)(SW_G + wr * 32 * NB) + l31;       // LDS row of block 0
  unsigned xr[MT][NB][4][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const unsigned row = row0 + 32 * nb;
        const swu32x2 u = *reinterpret_cast<const swu32x2*>(sw_lds + TA + sw_addr<C>(row, 4 * (wm + mt) + rg) + 8 * h);
        xr[mt][nb][rg][0] = u.x; xr[mt][nb][rg][1] = u.y;
      }
  __syncthreads();                                // every wave has its X before conv1 overwrites TA

  swf32x16 acc[MT][NB];
  const int tg0 = tb + wr * 32 * NB + (int)l31;   // time step of this lane's row of block 0
  const unsigned wrow_x = sw_swz<C>(row0) << 4;   // the swizzle of this lane's own rows (invariant under + 32)

  for (int d = 0; d < nd; ++d) {
    const int dil = P.dil[d];
    const uint16_t* w1 = P.w + (int64_t)(2 * d) * convw;
    const uint16_t* w2 = w1 + convw;
    // ---- conv1 (dilated): XA -> TA
    sw_gemm<C>(acc, ring, w1, k, dil, wl, XA, row0 - (unsigned)(halfk * dil), h);
    sw_prime<C>(ring, w2, k, wl);                 // conv2's first taps land under the epilogue and the barrier
    {
      const float* bias = P.bias + (2 * d) * C;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        swf32x4 bv[4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) bv[rg] = *reinterpret_cast<const swf32x4*>(bias + 32 * (wm + mt) + 8 * rg + 4 * h);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int tg = tg0 + 32 * nb;
          const bool inside = tg >= 0 && tg < Lseq;
          const unsigned rowb = (row0 + 32 * nb) * RB + 8 * h;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            // t = bf16(conv1 + b1); conv2's operand = bf16(lrelu(t)), zero outside [0, Lseq) (conv2's padding)
            swu32x2 o;
            o.x = sw_act(sw_pack(acc[mt][nb][4 * rg] + bv[rg].x, acc[mt][nb][4 * rg + 1] + bv[rg].y), slope);
            o.y = sw_act(sw_pack(acc[mt][nb][4 * rg + 2] + bv[rg].z, acc[mt][nb][4 * rg + 3] + bv[rg].w), slope);
            if (!inside) o = swu32x2{0u, 0u};
            *reinterpret_cast<swu32x2*>(sw_lds + TA + rowb + ((unsigned)((4 * (wm + mt) + rg) << 4) ^ wrow_x)) = o;
          }
        }
      }
    }
    __syncthreads();
    // ---- conv2 (dilation 1) + residual: TA -> the residual registers and XA
    sw_gemm<C>(acc, ring, w2, k, 1, wl, TA, row0 - (unsigned)halfk, h);
    const bool last = d + 1 == nd;
    if (!last) sw_prime<C>(ring, w2 + convw, k, wl);
    {
      const float* bias = P.bias + (2 * d + 1) * C;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        swf32x4 bv[4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) bv[rg] = *reinterpret_cast<const swf32x4*>(bias + 32 * (wm + mt) + 8 * rg + 4 * h);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int tg = tg0 + 32 * nb;
          const bool inside = tg >= 0 && tg < Lseq;
          const unsigned rowb = (row0 + 32 * nb) * RB + 8 * h;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            swu32x2 x;
            x.x = sw_pack(acc[mt][nb][4 * rg] + bv[rg].x + sw_lo(xr[mt][nb][rg][0]), acc[mt][nb][4 * rg + 1] + bv[rg].y + sw_hi(xr[mt][nb][rg][0]));
            x.y = sw_pack(acc[mt][nb][4 * rg + 2] + bv[rg].z + sw_lo(xr[mt][nb][rg][1]), acc[mt][nb][4 * rg + 3] + bv[rg].w + sw_hi(xr[mt][nb][rg][1]));
            if (!inside) x = swu32x2{0u, 0u};
            xr[mt][nb][rg][0] = x.x; xr[mt][nb][rg][1] = x.y;
            swu32x2 o = x;                        // last pair: the raw rows go to XA for the coalesced copy-out below
            if (!last) { o.x = sw_act(x.x, slope); o.y = sw_act(x.y, slope); }
            *reinterpret_cast<swu32x2*>(sw_lds + XA + rowb + ((unsigned)((4 * (wm + mt) + rg) << 4) ^ wrow_x)) = o;
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- copy-out: rows [halo, halo + TT) of XA -> HBM in 16-byte pieces (full 64 / 128-byte lines per row)
  {
    const int rows = (Lseq - t0 < TT ? Lseq - t0 : TT);
    const int total = rows * Gm::PPR;
    for (int p = tid; p < total; p += NT) {
      const int r = p / Gm::PPR, pc = p - r * Gm::PPR;
      const swu32x4 v = *reinterpret_cast<const swu32x4*>(sw_lds + XA + sw_addr<C>((unsigned)(SW_G + P.halo + r), (unsigned)pc));
      *reinterpret_cast<swu32x4*>(outg + (unsigned)((t0 + r) * C + pc * 8)) = v;
    }
  }
}

static int sw_halo(int k, const int* dil, int nd) {
  int hsum = 0;
  for (int d = 0; d < nd; ++d) hsum += (k - 1) / 2 * (dil[d] + 1);
  return hsum;
}
static int sw_rows(int C) { return C == 64 ? SwGeo<64>::R : SwGeo<32>::R; }

bool resblock_sw_bf16_supported(int C, int k, const int* dil, int nd) {
  if (C != 32 && C != 64) return false;
  if (k < 3 || k % 2 == 0 || nd < 1 || nd > BV2_RBCL_MAX_D) return false;
  if (C == 32 && k % 4 != 3) return false;          // the four-tap ring's tail is three taps (k = 3, 7, 11, ...)
  for (int d = 0; d < nd; ++d)
    if (dil[d] < 1 || (k - 1) / 2 * dil[d] > SW_G) return false;
  return 4 * sw_halo(k, dil, nd) <= sw_rows(C);     // at least half of every tile is output
}

int64_t resblock_sw_bf16_w_elems(int C, int k, int nd) { return (int64_t)2 * nd * k * (C / 16) * (C / 32) * 512; }

template <int C>
static int launch_sw(hipStream_t stream, const RbClLaunch& L, int max_tiles) {
  const size_t lds = (size_t)2 * SwGeo<C>::TILE;
  auto kern = resblock_sw_bf16_kernel<C>;
  ensure_dyn_lds((const void*)kern, lds);
  hipLaunchKernelGGL(kern, dim3(max_tiles, L.B, L.nprob), dim3(64 * SW_NW), lds, stream, L);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_resblock_sw_bf16(hipStream_t stream, const RbClLaunch& L0) {
  RbClLaunch L = L0;
  if (L.nprob < 1 || L.nprob > 3 || L.B < 1 || L.L < 1 || (L.C != 32 && L.C != 64) || (int64_t)L.L * L.C >= (1ll << 31)) return -1;
  const int R = sw_rows(L.C);
  int max_tiles = 0;
  for (int i = 0; i < L.nprob; ++i) {
    if (!resblock_sw_bf16_supported(L.C, L.p[i].k, L.p[i].dil, L.nd)) return -1;
    L.p[i].halo = sw_halo(L.p[i].k, L.p[i].dil, L.nd);
    const int TT = R - 2 * L.p[i].halo;
    const int nt = (L.L + TT - 1) / TT;
    if (nt > max_tiles) max_tiles = nt;
  }
  return L.C == 64 ? launch_sw<64>(stream, L, max_tiles) : launch_sw<32>(stream, L, max_tiles);
}

}  // namespace bv2

// resblock_c16_bf16.hip — a WHOLE HiFi-GAN ResBlock1 (reference modules.py:296-309: for each dilation d,
// x = x + conv2(lrelu(conv1(lrelu(x), k, d)), k, 1)) of the C = 16 Generator stage (512 samples per latent frame) in ONE kernel,
// bf16 channels-last, on v_mfma_f32_16x16x32_bf16 (round 5; the 32x32x16 form is resblock_cl_bf16.hip).
//
// Why a second kernel for one width.  resblock_cl_bf16<C16> ran its 32-row MFMA block half empty (16 of the 32 output rows are zero
// padding), on [row][16 + 8 pad] LDS tiles (104 KB: ONE 8-wave workgroup per CU, whose six GEMM / epilogue / barrier phases run in
// lock-step): 1.61 ms per step at B = 32 — 0.10 of the MFMA roof and 0.10 of the HBM roof (profiles/r04_h_pmc_c3.json).  Here
//   * one MFMA is 16 output channels x 16 time steps x K = 32 = TWO taps x 16 input channels: no padding rows, half the matrix time;
//   * an LDS row is the 32 bytes of one time step, unpadded.  The B operand of lane (t = lane & 15, q = lane >> 4) is the 16 bytes
//     [(q & 1) * 16, +16) of row (t + (2u + (q >> 1)) * dil): one ds_read_b128 whose four hardware lane groups ({0-3, 12-15, 20-27},
//     ...: MI355X_MICROARCH.md §LDS) each touch 16 distinct 16-byte slots of the 256-byte bank row for ANY row offset — conflict-free
//     without padding or swizzle, every offset an immediate;
//   * two tiles of (1024 + 64) rows are 68 KB: TWO workgroups per CU (four waves per SIMD, <= 128 registers), so one workgroup's
//     epilogue / barrier runs under the other's GEMM — the halo share of a tile (what sank the half-size-tile experiment of round 3)
//     is unchanged;
//   * the bias is the accumulator's initial value (the C operand of a conv's first MFMA), the residual stays packed (exact bf16
//     values) in 16 registers, the conv's whole weight set (<= 6 units of 1 KB) lives in registers and is replaced by the NEXT
//     conv's, unit by unit, behind its last use.
// Rounding points are exactly those of the layer-wise bf16 path / oracle generator_bf16 (every tensor that was stored to HBM there
// is rounded to bf16 here at the same place); only the fp32 summation order differs (pairs of taps per MFMA).
//
// Weight stream (bv2_model.cpp, rb16_w_index): per branch [d][conv e][unit u < (k+1)/2][lane 64][8 bf16]; lane = c_out + 16 q,
// q = 2 (tap & 1) + c_in / 8, element c_in % 8, tap = 2u + (q >> 1) (zero where tap >= k).  Bias fp32 [2 nd][16].
#include <hip/hip_runtime.h>
#include "../bv2_kernels.h"

namespace bv2 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) bf16x8 R16GlobalFrag;   // explicit global address space: a FLAT load would also count on lgkmcnt

namespace {

__device__ __forceinline__ float r16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float r16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned r16_pack(float a, float b) {     // round-to-nearest-even (v_cvt_pk_bf16_f32)
  bf16x2 r;
  r[0] = (__bf16)a; r[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float r16_lrelu(float v, float slope) { return v < 0.f ? v * slope : v; }
__device__ __forceinline__ unsigned r16_act(unsigned u, float slope) {   // bf16 pair -> bf16(lrelu(.)) pair
  return r16_pack(r16_lrelu(r16_lo(u), slope), r16_lrelu(r16_hi(u), slope));
}

constexpr int R16_G = 32;            // guard rows on each side of the LDS tiles (dilated taps reach <= 30 rows outside)
constexpr int R16_NW = 8;            // waves per workgroup
constexpr int R16_NB = 8;            // 16-row blocks per wave (128 rows)
// blocks per GEMM pass: HB = 4 (two passes: 16 accumulator + 32 B-operand registers live) where the conv's weight set leaves room
// (KU <= 4 units), HB = 2 for the k = 9 / 11 branches (KU = 5 / 6: 24 weight registers)
template <int KU> struct R16Hb { static constexpr int v = KU <= 4 ? 4 : 2; };
constexpr int R16_R = 16 * R16_NB * R16_NW;   // 1024 rows per tile incl. halo
constexpr int R16_ROWS = R16_R + 2 * R16_G;
constexpr int R16_P = 16;            // elements per LDS row (32 bytes, unpadded)

extern __shared__ __attribute__((aligned(16))) unsigned short r16_lds[];
__device__ __forceinline__ bf16x8 r16_rd128(unsigned off) {
  return *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(r16_lds) + off);
}
__device__ __forceinline__ void r16_wr64(unsigned off, u32x2 v) {
  *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(r16_lds) + off) = v;
}
constexpr unsigned R16_TILE_BYTES = R16_ROWS * R16_P * 2;       // XA at byte 0, TA behind it
constexpr unsigned R16_BLK = 16 * R16_P * 2;                    // bytes between two 16-row blocks (512: an immediate offset)

// Per-lane constants, kept as FEW registers on purpose: every per-block address / time index is (one of these) + a compile-time
// constant.  (First build: the compiler hoisted eight LDS addresses, eight 64-bit global offsets and their predicates out of the
// dilation loop — 160 registers wanted, 45 spilled.)
struct R16Ctx {
  int tg0;                           // time step of this lane's row of block 0: tb + row0 + t
  int t0, tend, Lseq;                // rows [t0, tend) of the utterance are this tile's outputs
  unsigned own;                      // byte offset (inside a tile) of this lane's 8 output bytes of block 0: ((G + row0 + t) * 16 + 4 q) * 2
  unsigned rd;                       // byte offset of this lane's B-operand piece at tap offset 0: ((G + row0 + t) * 16 + (q & 1) * 8) * 2
  int qt;                            // q >> 1: which tap of a pair this lane feeds
  int q4;                            // 4 * q: first of this lane's 4 output channels
  float slope;
  unsigned wl;                       // lane * 16: byte offset of this lane's fragment inside a 1 KB weight unit
  int sum_mode;                      // stage hand-over: 0 = plain store, 1 = first branch (plain store into the sum tensor), 2 = add to the
  float sum_scale;                   // running sum, 3 = last branch (add, scale by 1/n): cl_bf16.h stage_mean's rounding points
};

// One conv of the block: acc = bias + sum_u W(u) x B(u) over the KU tap pairs, for the wave's 8 blocks in R16_NB / R16_HB passes.
//   MODE 1 (conv1): dst[row] = bf16(lrelu(bf16(acc)))                       (conv2's operand; zero outside [0, Lseq))
//   MODE 2 (conv2): x = bf16(acc + x) (zero outside); last pair: stored to HBM, else dst[row] = bf16(lrelu(x))
// While the last pass consumes unit u for the last time, W[u] is refilled with the NEXT conv's unit u (wnext), bv with its bias.
// src_tile / dst_tile: byte offset of the tile read / written (0 = XA, R16_TILE_BYTES = TA).
template <int KU, int MODE>
__device__ __forceinline__ void r16_conv(R16Ctx c, bf16x8 (&W)[KU], const uint16_t* wnext, f32x4& bv, const float* bias_next, int dil,
                                         int halfk, unsigned src_tile, unsigned dst_tile, unsigned (&xr)[R16_NB][2], bool last,
                                         uint16_t* outg) {
  // opaque to the optimiser: whatever is derived from these is recomputed per conv (a few adds) instead of living across the loop
  asm volatile("" : "+v"(c.tg0), "+v"(c.own), "+v"(c.rd));
  const unsigned ustep = (unsigned)(2 * dil) * (R16_P * 2);            // bytes between the rows of tap 2u and tap 2u + 2
  const unsigned rd0 = src_tile + c.rd + (unsigned)((c.qt - halfk) * dil * (R16_P * 2));
  const unsigned wr0 = dst_tile + c.own;
  constexpr int R16_HB = R16Hb<KU>::v;
#pragma unroll
  for (int h = 0; h < R16_NB / R16_HB; ++h) {
    const unsigned xb = rd0 + h * R16_HB * R16_BLK;
    f32x4 acc[R16_HB];
    bf16x8 bb[2][R16_HB];
#pragma unroll
    for (int nb = 0; nb < R16_HB; ++nb) bb[0][nb] = r16_rd128(xb + nb * R16_BLK);
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      if (u + 1 < KU) {
        const unsigned xn = xb + (u + 1) * ustep;
#pragma unroll
        for (int nb = 0; nb < R16_HB; ++nb) bb[(u & 1) ^ 1][nb] = r16_rd128(xn + nb * R16_BLK);
      }
#pragma unroll
      for (int nb = 0; nb < R16_HB; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[u], bb[u & 1][nb], u == 0 ? bv : acc[nb], 0, 0, 0);
      if (h == R16_NB / R16_HB - 1) {
        W[u] = *(const R16GlobalFrag*)(reinterpret_cast<const char*>(wnext + u * 512) + c.wl);
        // the NEXT conv's bias rides behind this conv's last use of its own (the C operand of the last pass's first MFMAs): loaded at the
        // top of a conv it was one exposed global round trip per conv in front of the first MFMA (ISA of the first build)
        if (u == 0) bv = *reinterpret_cast<const f32x4*>(bias_next + c.q4);
      }
      // pin the emitted order: the next unit's LDS reads first (they land under this unit's MFMAs), the MFMAs, then the weight refill
      if (u + 1 < KU) __builtin_amdgcn_sched_group_barrier(0x100, R16_HB, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, R16_HB, 0);
      if (h == R16_NB / R16_HB - 1) {
        if (u == 0) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);          // keep program order between units: nothing of unit u + 2 is hoisted above this point
    }
    // ---- epilogue of this pass: lane (t, q) holds channels [4q, 4q + 4) of time step t of each block
#pragma unroll
    for (int nb = 0; nb < R16_HB; ++nb) {
      const int ni = R16_HB * h + nb;
      const int tg = c.tg0 + 16 * ni;
      const bool inside = tg >= 0 && tg < c.Lseq;
      if (MODE == 1) {
        u32x2 o;
        o.x = r16_act(r16_pack(acc[nb][0], acc[nb][1]), c.slope);
        o.y = r16_act(r16_pack(acc[nb][2], acc[nb][3]), c.slope);
        if (!inside) o = u32x2{0u, 0u};
        r16_wr64(wr0 + ni * R16_BLK, o);
      } else {
        u32x2 x;
        x.x = r16_pack(acc[nb][0] + r16_lo(xr[ni][0]), acc[nb][1] + r16_hi(xr[ni][0]));
        x.y = r16_pack(acc[nb][2] + r16_lo(xr[ni][1]), acc[nb][3] + r16_hi(xr[ni][1]));
        if (!inside) x = u32x2{0u, 0u};
        xr[ni][0] = x.x; xr[ni][1] = x.y;
        if (last) {
          // 32-bit element offset inside the batch item (L * 16 < 2^31 is checked by the launcher)
          if (tg >= c.t0 && tg < c.tend) {
            u32x2* og = reinterpret_cast<u32x2*>(outg + (unsigned)(tg * 16 + c.q4));
            if (c.sum_mode >= 2) {
              // hand-over: the running sum of the earlier branches — this LANE's own store of the previous iteration (same tile origin for
              // every branch), read back from L2 — plus this branch's output
              const u32x2 pv = *og;
              float s0 = r16_lo(pv.x) + r16_lo(x.x), s1 = r16_hi(pv.x) + r16_hi(x.x), s2 = r16_lo(pv.y) + r16_lo(x.y), s3 = r16_hi(pv.y) + r16_hi(x.y);
              if (c.sum_mode == 3) { s0 *= c.sum_scale; s1 *= c.sum_scale; s2 *= c.sum_scale; s3 *= c.sum_scale; }
              x.x = r16_pack(s0, s1); x.y = r16_pack(s2, s3);
            }
            *og = x;
          }
        } else {
          u32x2 o;
          o.x = r16_act(x.x, c.slope); o.y = r16_act(x.y, c.slope);
          r16_wr64(wr0 + ni * R16_BLK, o);
        }
      }
    }
  }
}

template <int KU>
__device__ __forceinline__ void r16_block(const RbClLaunch& L, const RbClProb& P, int b, int t0, int TT, int Lseq, int halo, int sum_mode) {
  constexpr int NT = 64 * R16_NW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 15, q = lane >> 4, row0 = wid * 16 * R16_NB;
  R16Ctx c;
  c.tg0 = t0 - halo + row0 + t;
  c.t0 = t0; c.tend = t0 + TT < Lseq ? t0 + TT : Lseq; c.Lseq = Lseq;
  c.own = (unsigned)(((R16_G + row0 + t) * R16_P + 4 * q) * 2);
  c.rd = (unsigned)(((R16_G + row0 + t) * R16_P + (q & 1) * 8) * 2);
  c.qt = q >> 1; c.q4 = 4 * q;
  c.slope = L.slope; c.wl = (unsigned)lane * 16u;
  c.sum_mode = sum_mode; c.sum_scale = L.sum_scale;
  const int Lrow = L.L, nd = L.nd, halfk = (P.k - 1) / 2;
  const uint16_t* xg = P.x + (int64_t)b * Lrow * 16;
  uint16_t* outg = (L.sum_out ? L.sum_out : P.out) + (int64_t)b * Lrow * 16;

  // the first conv's weights go in flight before anything else
  bf16x8 W[KU];
#pragma unroll
  for (int u = 0; u < KU; ++u) W[u] = *(const R16GlobalFrag*)(reinterpret_cast<const char*>(P.w + u * 512) + c.wl);
  f32x4 bv = *reinterpret_cast<const f32x4*>(P.bias + c.q4);

  // ---- this wave's rows of x: residual registers (packed, exact bf16) and bf16(lrelu(x)) -> XA; rows outside [0, Lseq) are zero
  unsigned xr[R16_NB][2];
  {
    u32x2 v[R16_NB];
#pragma unroll
    for (int ni = 0; ni < R16_NB; ++ni) {
      const int tg = c.tg0 + 16 * ni;
      const int tc = tg < 0 ? 0 : (tg >= Lseq ? Lseq - 1 : tg);         // clamped: the loads are unconditional
      v[ni] = *reinterpret_cast<const u32x2*>(xg + (unsigned)(tc * 16 + c.q4));
    }
    // guard rows of both tiles (never written again): [0, G) and [R + G, R + 2G), 2 pieces of 16 bytes per row
    for (int p = tid; p < 4 * R16_G; p += NT) {
      const int gr = p >> 1, pc = p & 1;
      const int row = gr < R16_G ? gr : R16_R + gr;
      const unsigned off = (unsigned)((row * R16_P + pc * 8) * 2);
      *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(r16_lds) + off) = u32x4{0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(r16_lds) + R16_TILE_BYTES + off) = u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int ni = 0; ni < R16_NB; ++ni) {
      const int tg = c.tg0 + 16 * ni;
      const bool ok = tg >= 0 && tg < Lseq;
      u32x2 raw = ok ? v[ni] : u32x2{0u, 0u};
      xr[ni][0] = raw.x; xr[ni][1] = raw.y;
      u32x2 a;
      a.x = r16_act(raw.x, c.slope); a.y = r16_act(raw.y, c.slope);
      r16_wr64(c.own + ni * R16_BLK, a);
    }
  }
  __syncthreads();

  const int nconv = 2 * nd;
  for (int d = 0; d < nd; ++d) {
    const int c1 = 2 * d, c2 = 2 * d + 1, c3 = c2 + 1 < nconv ? c2 + 1 : c2;      // after the last conv: re-read it (unused)
    (void)c1;
    r16_conv<KU, 1>(c, W, P.w + (int64_t)c2 * KU * 512, bv, P.bias + c2 * 16, P.dil[d], halfk, 0u, R16_TILE_BYTES, xr, false, outg);
    __syncthreads();
    r16_conv<KU, 2>(c, W, P.w + (int64_t)c3 * KU * 512, bv, P.bias + c3 * 16, 1, halfk, R16_TILE_BYTES, 0u, xr, d + 1 == nd, outg);
    if (d + 1 < nd) __syncthreads();
  }
}

}  // namespace

__global__ void __launch_bounds__(64 * R16_NW, 4) resblock_c16_bf16_kernel(const RbClLaunch L) {
  // stage hand-over (L.sum_out): one workgroup runs the nprob branches of its tile one after the other — same tile origin (the widest halo) for
  // every branch, so a lane owns the same output elements in each — and leaves ONE tensor: x is read from HBM once (the later branches find it
  // in L2) and one output is written instead of one per branch
  const int nit = L.sum_out ? L.nprob : 1;
  for (int it = 0; it < nit; ++it) {
    const RbClProb& P = L.p[__builtin_amdgcn_readfirstlane(L.sum_out ? it : (int)blockIdx.z)];
    const int halo = L.sum_out ? L.halo_max : P.halo;
    const int TT = R16_R - 2 * halo;              // output rows per tile
    const int t0 = blockIdx.x * TT;
    if (t0 >= L.L) return;                        // branches with a smaller halo need fewer tiles
    const int b = blockIdx.y;
    int Lseq = L.L;
    if (L.lens) {
      const int64_t lv = L.lens[b] * L.len_mul;
      Lseq = __builtin_amdgcn_readfirstlane(lv < L.L ? (int)lv : L.L);
      if (t0 >= Lseq) return;                     // a tile wholly past the utterance: nobody reads its outputs
    }
    const int sm = !L.sum_out ? 0 : (it == 0 ? 1 : (it + 1 == nit ? 3 : 2));
    switch (P.k) {                                // wave-uniform: KU = (k + 1) / 2 tap pairs per conv
      case 3: r16_block<2>(L, P, b, t0, TT, Lseq, halo, sm); break;
      case 5: r16_block<3>(L, P, b, t0, TT, Lseq, halo, sm); break;
      case 7: r16_block<4>(L, P, b, t0, TT, Lseq, halo, sm); break;
      case 9: r16_block<5>(L, P, b, t0, TT, Lseq, halo, sm); break;
      default: r16_block<6>(L, P, b, t0, TT, Lseq, halo, sm); break;
    }
    if (it + 1 < nit) __syncthreads();            // the next branch re-stages x over the tiles
  }
}

static int r16_halo(int k, const int* dil, int nd) {
  int h = 0;
  for (int d = 0; d < nd; ++d) h += (k - 1) / 2 * (dil[d] + 1);
  return h;
}

bool resblock_c16_bf16_supported(int C, int k, const int* dil, int nd) {
  if (C != 16) return false;
  if (k < 3 || k > 11 || k % 2 == 0 || nd < 1 || nd > BV2_RBCL_MAX_D) return false;
  for (int d = 0; d < nd; ++d)
    if (dil[d] < 1 || ((k + 1) / 2) * dil[d] > R16_G) return false;     // the zero tap of the last pair reads (k+1)/2 * dil rows outside
  return 4 * r16_halo(k, dil, nd) <= R16_R;       // at least half of every tile is output
}

int launch_resblock_c16_bf16(hipStream_t stream, const RbClLaunch& L0) {
  RbClLaunch L = L0;
  if (L.nprob < 1 || L.nprob > 3 || L.B < 1 || L.L < 1 || L.C != 16 || (int64_t)L.L * 16 >= (1ll << 31)) return -1;
  int max_tiles = 0;
  L.halo_max = 0;
  for (int i = 0; i < L.nprob; ++i) {
    if (!resblock_c16_bf16_supported(L.C, L.p[i].k, L.p[i].dil, L.nd)) return -1;
    L.p[i].halo = r16_halo(L.p[i].k, L.p[i].dil, L.nd);
    if (L.p[i].halo > L.halo_max) L.halo_max = L.p[i].halo;
  }
  L.sum_scale = 1.f / (float)L.nprob;
  for (int i = 0; i < L.nprob; ++i) {
    const int TT = R16_R - 2 * (L.sum_out ? L.halo_max : L.p[i].halo);
    const int nt = (L.L + TT - 1) / TT;
    if (nt > max_tiles) max_tiles = nt;
  }
  const size_t lds = (size_t)2 * R16_ROWS * R16_P * 2;
  ensure_dyn_lds((const void*)resblock_c16_bf16_kernel, lds);
  hipLaunchKernelGGL(resblock_c16_bf16_kernel, dim3(max_tiles, L.B, L.sum_out ? 1 : L.nprob), dim3(64 * R16_NW), lds, stream, L);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace bv2

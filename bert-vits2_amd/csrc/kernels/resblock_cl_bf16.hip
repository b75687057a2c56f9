// resblock_cl_bf16.hip — a WHOLE HiFi-GAN ResBlock1 (reference modules.py:296-309: for each dilation d,
// x = x + conv2(lrelu(conv1(lrelu(x), k, d)), k, 1)) in ONE kernel, bf16 channels-last, for the narrow stages of the
// Generator (C = 32 / 16, i.e. 256 / 512 samples per latent frame).
//
// Why: at these widths the layer-wise convs move [B][98k..196k][C] activations with an arithmetic intensity of 60-120
// FLOP/B in bf16 — far below the machine balance (2.5 PF / 8 TB/s = 310 FLOP/B): profiles/r01_e shows the stage-3/4
// conv launches pinned at 2.5-2.9 TB/s of algorithmic traffic.  A ResBlock is 6 convs = 18 tensor passes layer by layer;
// here a workgroup stages one time tile (plus the halo the 6 convs consume: sum_d (k-1)/2*(d+1) <= 60 rows a side) once,
// runs all 6 convs LDS -> MFMA -> LDS, and writes the tile once: 2 passes.  The halo is recomputed (13-19 % extra MFMA
// work), which is cheap next to 9x less HBM traffic.
//
// Data flow per workgroup (NW waves, each owning NI blocks of 32 time steps; R = 32*NW*NI rows incl. halo):
//   XA [R+2G][C+8] bf16 : bf16(lrelu(x))  — conv1's B operand (activated ONCE per element, not once per tap)
//   TA [R+2G][C+8] bf16 : bf16(lrelu(bf16(conv1 + b1)))  — conv2's B operand; first holds the raw tile for the X registers
//   X  registers  fp32  : the wave's own rows of x (residual), exact bf16 values, carried across the 3 pairs
// Rows outside [0, L) are forced to zero after every conv (they are the NEXT conv's zero padding, not conv outputs).
// Rounding points are exactly those of the layer-wise bf16 path / oracle generator_bf16 (every tensor that was stored to
// HBM there is rounded to bf16 here at the same place), so both paths agree up to fp32 summation order.
// Weights: ONE contiguous bf16 fragment stream per (stage, branch) [d][conv][tap-major unit, padded to 8][lane][8] (+8 tail units),
// streamed global -> registers through an 8-deep ring that is never drained between the 6 convs.
#include <hip/hip_runtime.h>
#include "../bv2_kernels.h"

namespace bv2 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ float rb_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float rb_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned rb_pack(float a, float b) {     // round-to-nearest-even (v_cvt_pk_bf16_f32)
  bf16x2 r;
  r[0] = (__bf16)a; r[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float rb_lrelu(float v, float slope) { return v < 0.f ? v * slope : v; }

constexpr int RB_G = 32;          // guard rows on each side of the LDS tiles (dilated taps reach <= 25 rows outside)
// (Round 3, measured and not kept: 4 waves and half the rows per workgroup — R = 384 / 512, 72 / 55 KB of LDS — so that TWO
// independent workgroups share a CU and one's epilogue overlaps the other's GEMM.  PMC (profiles/r03_e_pmc_c3.json) has this kernel's
// MFMA pipe busy 0.27-0.28 of the time and its LDS array 0.25 (a fifth of it bank conflicts): neither is the bound, the lock-step
// GEMM / epilogue / barrier phases of the one resident workgroup are.  But at R = 384 only 264 of the rows are output for k = 11
// (648 of 768 now), and the extra halo GEMM work ate the overlap: 2.93-2.97 -> 3.02-3.03 ms per step in a same-box A/B of builds.)

struct Ring {
  bf16x8 ar[RBCL_PD];
  const uint16_t* wp;             // + lane*8; advances one unit (512 elements) per load
  __device__ __forceinline__ void load(int slot) {
    ar[slot] = *reinterpret_cast<const bf16x8*>(wp);
    wp += 512;
  }
};

// acc[ni] += sum_{u < U} W(u) x B(u, ni), units in TAP-MAJOR order u = j*G + s (tap j, 16-channel group s; the packer writes
// the stream in this order): the group index of ring slot i is the compile-time i % G, a tap is one pointer add, every LDS
// offset is an immediate.  The ring already holds this conv's first RBCL_PD units and is refilled from the contiguous stream
// (which continues into the next conv's units) as it is consumed; Upad = U rounded up to RBCL_PD.  (The first form of this
// loop recomputed (group, tap) and the LDS address per unit: 9.6 instructions per MFMA, which two waves per SIMD cannot
// issue under 32-cycle MFMAs.)
template <int NI, int G, int PITCH>
__device__ __forceinline__ void rb_gemm(f32x16 (&acc)[NI], Ring& ring, int U, int Upad, const unsigned short* xb, int tstep) {
  static_assert(RBCL_PD % G == 0, "ring slots map to fixed groups");
  bf16x8 bb[2][NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) bb[0][ni] = *reinterpret_cast<const bf16x8*>(xb + ni * 32 * PITCH);
  const unsigned short* xrow = xb;                // row of the current tap
  const int tap_step = tstep * PITCH;
  for (int u0 = 0; u0 < Upad; u0 += RBCL_PD) {
    const bool full = u0 + RBCL_PD <= U;          // wave-uniform: whole ring rounds run without per-unit guards
#pragma unroll
    for (int i = 0; i < RBCL_PD; ++i) {
      constexpr int dummy = 0; (void)dummy;
      const int s = i % G;
      const bool last_of_tap = s == G - 1;
      if (full || u0 + i < U) {
        // next unit: next group of this tap, or group 0 of the next tap (one tap past the end reads guard rows: unused)
        const unsigned short* xn = last_of_tap ? xrow + tap_step : xrow + (s + 1) * 16;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bb[(i & 1) ^ 1][ni] = *reinterpret_cast<const bf16x8*>(xn + ni * 32 * PITCH);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring.ar[i], bb[i & 1][ni], acc[ni], 0, 0, 0);
        if (last_of_tap) xrow += tap_step;
      }
      ring.load(i);
      __builtin_amdgcn_sched_group_barrier(0x100, NI, 0);   // next unit's LDS reads first, then the MFMAs, then the ring load
      __builtin_amdgcn_sched_group_barrier(0x008, NI, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_barrier(0);          // keep program order: the ring's vmcnt distances stay RBCL_PD - 1 units
    }
  }
}

}  // namespace

template <int C, int NW, int NI>
__global__ void __launch_bounds__(64 * NW) resblock_cl_bf16_kernel(const RbClLaunch L) {
  constexpr int NT = 64 * NW, R = 32 * NW * NI, PITCH = C + 8, ROWS = R + 2 * RB_G, NG = C / 8, PPR = C / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned short rb_lds[];
  unsigned short* XA = rb_lds;
  unsigned short* TA = rb_lds + ROWS * PITCH;
  const RbClProb& P = L.p[blockIdx.z];
  const int TT = R - 2 * P.halo;                  // output rows per tile
  const int t0 = blockIdx.x * TT;
  if (t0 >= L.L) return;                          // branches with a smaller halo need fewer tiles
  const int b = blockIdx.y;
  const int tb = t0 - P.halo;                     // time step of tile row 0
  const int Lrow = L.L, k = P.k, nd = L.nd;      // Lrow: rows per batch item in HBM; Lseq: valid rows of THIS item
  int Lseq = Lrow;
  if (L.lens) {
    const int64_t lv = L.lens[b] * L.len_mul;
    Lseq = lv < Lrow ? (int)lv : Lrow;
    if (t0 >= Lseq) return;                       // a tile wholly past the utterance: nobody reads its outputs
  }
  const float slope = L.slope;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const uint16_t* xg = P.x + (int64_t)b * Lrow * C;

  // weight ring: primed once, never drained between the convs
  Ring ring;
  ring.wp = P.w + lane * 8;
#pragma unroll
  for (int i = 0; i < RBCL_PD; ++i) { ring.load(i); __builtin_amdgcn_sched_barrier(0); }

  // ---- stage the tile: raw -> TA (for the X registers), bf16(lrelu) -> XA; rows outside [0, L) and the guards are zero
  for (int p = tid; p < 2 * RB_G * PPR; p += NT) {
    const int gr = p / PPR, cb = p - gr * PPR;
    const int row = gr < RB_G ? gr : R + gr;      // [0, G) and [R + G, R + 2G)
    *reinterpret_cast<u32x4*>(XA + row * PITCH + cb * 8) = u32x4{0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4*>(TA + row * PITCH + cb * 8) = u32x4{0u, 0u, 0u, 0u};
  }
  for (int base = 0; base < R * PPR; base += 4 * NT) {
    u32x4 v[4];
    int dst[4];
    bool ok[4], inb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int p = base + q * NT + tid;
      inb[q] = p < R * PPR;
      p = inb[q] ? p : R * PPR - 1;
      const int r = p / PPR, cb = p - r * PPR;
      const int t = tb + r;
      ok[q] = t >= 0 && t < Lseq;
      const int tc = t < 0 ? 0 : (t >= Lseq ? Lseq - 1 : t);
      dst[q] = (RB_G + r) * PITCH + cb * 8;
      v[q] = *reinterpret_cast<const u32x4*>(xg + (int64_t)tc * C + cb * 8);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!inb[q]) continue;
      u32x4 raw = ok[q] ? v[q] : u32x4{0u, 0u, 0u, 0u}, act;
#pragma unroll
      for (int w = 0; w < 4; ++w) act[w] = rb_pack(rb_lrelu(rb_lo(raw[w]), slope), rb_lrelu(rb_hi(raw[w]), slope));
      *reinterpret_cast<u32x4*>(TA + dst[q]) = raw;
      *reinterpret_cast<u32x4*>(XA + dst[q]) = act;
    }
  }
  __syncthreads();

  // ---- this wave's rows of x (the residual) into registers: lane = time step, group g = channels [8g + 4lh, +4)
  float xr[NI][NG][4];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int row = (wid * NI + ni) * 32 + l31;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const u32x2 u = *reinterpret_cast<const u32x2*>(TA + (RB_G + row) * PITCH + 8 * g + 4 * lh);
      xr[ni][g][0] = rb_lo(u.x); xr[ni][g][1] = rb_hi(u.x); xr[ni][g][2] = rb_lo(u.y); xr[ni][g][3] = rb_hi(u.y);
    }
  }
  __syncthreads();                                // every wave has its X before conv1 overwrites TA

  const int U = (C / 16) * k, Upad = (U + RBCL_PD - 1) / RBCL_PD * RBCL_PD;
  const int half = (k - 1) / 2;
  uint16_t* outg = P.out + (int64_t)b * Lrow * C;
  const int row0 = wid * NI * 32 + l31;           // this lane's row in block ni: row0 + 32*ni

  for (int d = 0; d < nd; ++d) {
    const int dil = P.dil[d];
    f32x16 acc[NI];
    // ---- conv1 (dilated): XA -> TA
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
    // this conv's bias (NG float4 per lane) goes in flight BEFORE the GEMM and lands under its MFMAs.  Loaded inside the (ni, g)
    // loop below, between LDS stores, every iteration waited for its own load: NI*NG serial round trips per conv, 6 convs per
    // workgroup, with every wave of the workgroup in its epilogue at the same time (ISA of round 2) — a large part of what looked
    // like an LDS-bandwidth bound on this kernel
    f32x4 bvv[NG];
    {
      const float* bias = P.bias + (2 * d) * 32;
#pragma unroll
      for (int g = 0; g < NG; ++g) bvv[g] = *reinterpret_cast<const f32x4*>(bias + 8 * g + 4 * lh);
    }
    rb_gemm<NI, C / 16, PITCH>(acc, ring, U, Upad, XA + (RB_G + row0 - half * dil) * PITCH + lh * 8, dil);
    {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int row = row0 + 32 * ni;
        const int t = tb + row;
        const bool inside = t >= 0 && t < Lseq;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const f32x4 bv = bvv[g];
          float v0 = acc[ni][4 * g] + bv.x, v1 = acc[ni][4 * g + 1] + bv.y, v2 = acc[ni][4 * g + 2] + bv.z,
                v3 = acc[ni][4 * g + 3] + bv.w;
          // t = bf16(conv1 + b1); conv2's operand = bf16(lrelu(t))
          const unsigned q0 = rb_pack(v0, v1), q1 = rb_pack(v2, v3);
          u32x2 o;
          o.x = rb_pack(rb_lrelu(rb_lo(q0), slope), rb_lrelu(rb_hi(q0), slope));
          o.y = rb_pack(rb_lrelu(rb_lo(q1), slope), rb_lrelu(rb_hi(q1), slope));
          if (!inside) o = u32x2{0u, 0u};
          *reinterpret_cast<u32x2*>(TA + (RB_G + row) * PITCH + 8 * g + 4 * lh) = o;
        }
      }
    }
    __syncthreads();
    // ---- conv2 (dil 1) + residual: TA -> X registers, XA (or HBM after the last pair)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
    {
      const float* bias = P.bias + (2 * d + 1) * 32;
#pragma unroll
      for (int g = 0; g < NG; ++g) bvv[g] = *reinterpret_cast<const f32x4*>(bias + 8 * g + 4 * lh);
    }
    rb_gemm<NI, C / 16, PITCH>(acc, ring, U, Upad, TA + (RB_G + row0 - half) * PITCH + lh * 8, 1);
    {
      const bool last = d + 1 == nd;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int row = row0 + 32 * ni;
        const int t = tb + row;
        const bool inside = t >= 0 && t < Lseq;
        const bool store = t >= t0 && t < t0 + TT && t < Lseq;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const f32x4 bv = bvv[g];
          const float v0 = acc[ni][4 * g] + bv.x + xr[ni][g][0], v1 = acc[ni][4 * g + 1] + bv.y + xr[ni][g][1],
                      v2 = acc[ni][4 * g + 2] + bv.z + xr[ni][g][2], v3 = acc[ni][4 * g + 3] + bv.w + xr[ni][g][3];
          u32x2 q;
          q.x = rb_pack(v0, v1); q.y = rb_pack(v2, v3);
          if (!inside) q = u32x2{0u, 0u};
          xr[ni][g][0] = rb_lo(q.x); xr[ni][g][1] = rb_hi(q.x); xr[ni][g][2] = rb_lo(q.y); xr[ni][g][3] = rb_hi(q.y);
          if (last) {
            if (store) *reinterpret_cast<u32x2*>(outg + (int64_t)t * C + 8 * g + 4 * lh) = q;
          } else {
            u32x2 o;
            o.x = rb_pack(rb_lrelu(xr[ni][g][0], slope), rb_lrelu(xr[ni][g][1], slope));
            o.y = rb_pack(rb_lrelu(xr[ni][g][2], slope), rb_lrelu(xr[ni][g][3], slope));
            *reinterpret_cast<u32x2*>(XA + (RB_G + row) * PITCH + 8 * g + 4 * lh) = o;
          }
        }
      }
    }
    if (d + 1 < nd) __syncthreads();
  }
}

// total halo the nd (conv1, conv2) pairs consume on each side of a tile
static int rb_halo(int k, const int* dil, int nd) {
  int h = 0;
  for (int d = 0; d < nd; ++d) h += (k - 1) / 2 * (dil[d] + 1);
  return h;
}

bool resblock_cl_bf16_supported(int C, int k, const int* dil, int nd) {
  if (C != 16 && C != 32) return false;
  if (k < 1 || k % 2 == 0 || nd < 1 || nd > BV2_RBCL_MAX_D) return false;
  for (int d = 0; d < nd; ++d)
    if (dil[d] < 1 || (k - 1) / 2 * dil[d] > RB_G) return false;
  const int R = C == 32 ? 768 : 1024;
  return 4 * rb_halo(k, dil, nd) <= R;            // at least half of every tile is output
}

int resblock_cl_bf16_units(int C, int k) { return ((C / 16) * k + RBCL_PD - 1) / RBCL_PD * RBCL_PD; }

template <int C, int NW, int NI>
static int launch_rb(hipStream_t stream, const RbClLaunch& L, int max_tiles) {
  constexpr int R = 32 * NW * NI;
  const size_t lds = (size_t)2 * (R + 2 * RB_G) * (C + 8) * 2;
  auto kern = resblock_cl_bf16_kernel<C, NW, NI>;
  ensure_dyn_lds((const void*)kern, lds);
  hipLaunchKernelGGL(kern, dim3(max_tiles, L.B, L.nprob), dim3(64 * NW), lds, stream, L);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_resblock_cl_bf16(hipStream_t stream, const RbClLaunch& L0) {
  RbClLaunch L = L0;
  if (L.nprob < 1 || L.nprob > 3 || L.B < 1 || L.L < 1) return -1;
  const int R = L.C == 32 ? 768 : 1024;
  int max_tiles = 0;
  for (int i = 0; i < L.nprob; ++i) {
    if (!resblock_cl_bf16_supported(L.C, L.p[i].k, L.p[i].dil, L.nd)) return -1;
    L.p[i].halo = rb_halo(L.p[i].k, L.p[i].dil, L.nd);
    const int TT = R - 2 * L.p[i].halo;
    const int nt = (L.L + TT - 1) / TT;
    if (nt > max_tiles) max_tiles = nt;
  }
  if (L.C == 32) return launch_rb<32, 8, 3>(stream, L, max_tiles);
  return launch_rb<16, 8, 4>(stream, L, max_tiles);
}

double resblock_cl_bf16_flops(const RbClLaunch& L) {
  double f = 0;
  for (int i = 0; i < L.nprob; ++i) f += 2.0 * 2 * L.nd * L.C * L.C * L.p[i].k * (double)L.L * L.B;
  return f;
}

double resblock_cl_bf16_bytes(const RbClLaunch& L) {   // x read once, out written once, weights once (hand-over: x and ONE output for all branches)
  double by = 0;
  for (int i = 0; i < L.nprob; ++i)
    by += 2.0 * ((L.sum_out && i ? 0.0 : 2.0) * L.C * (double)L.L * L.B + 2.0 * L.nd * L.C * L.C * L.p[i].k);
  return by;
}

}  // namespace bv2

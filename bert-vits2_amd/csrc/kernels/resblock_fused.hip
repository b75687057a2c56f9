// resblock_fused.hip — one (dilated conv, conv) pair of a HiFi-GAN ResBlock1 in ONE kernel, for the narrow stages of the
// Generator (C <= 32):   out = x + conv2(lrelu(conv1(lrelu(x), dil d) + b1), dil 1) + b2      (reference modules.py:296-309)
//
// Why: at C = 32 / 16 the layer-wise convolutions have an arithmetic intensity of 28-56 FLOP/B — they are bound by
// moving [C, 98k-196k] activations, not by the matrix core.  Keeping the intermediate activation of the
// pair in LDS removes one full write + read of it (5 tensor passes -> 3: read x, re-read x for the residual, write out),
// halves the launch count of these stages and doubles the MFMA work per byte.
//
// Tile: 224 output columns (7 x 32; 896 B = 7 cache lines, so tiles stay line-aligned).  The intermediate is computed on
// 256 columns (8 x 32) starting 16 columns early, which covers conv2's halo (k-1)/2 <= 16; x is staged on
// 256 + (k-1)*d <= 320 columns.  8 waves: in phase 1 wave w owns intermediate columns [32w, 32w+32), in phase 2 waves
// 0..6 own output columns [32w, 32w+32).  M = 32 rows (the whole channel dim; rows >= C meet zero weights).
// Operands: A (weights, fragment order, one contiguous 1 KB unit per (group, tap)) streams global -> registers through an
// 8-deep ring exactly as in the split-K kernel (all 8 waves read the same units: L1 hits); B comes from LDS.
// Zero padding: x outside [0, L) is staged as 0; the intermediate outside [0, L) is FORCED to 0 (it is conv2's padding,
// not conv1 applied to padding).
#include <hip/hip_runtime.h>
#include "../bv2_kernels.h"

namespace bv2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 rf_ld4(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float rf_ld(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

constexpr int RF_BN = 224;        // output columns per tile
constexpr int RF_TW = 256;        // intermediate columns per tile
constexpr int RF_LEAD = 16;       // intermediate starts this many columns before the outputs
constexpr int RF_XS = 5;          // 64-column strips of the staged x tile
// LDS pitches: both ≡ 16 (mod 32 banks).  The C = 16 path's B operand (rf_gemm16) is read by 16-lane groups that sit on ADJACENT
// rows (lanes 0-15 row r, lanes 16-31 row r + 1, serviced in one LDS cycle group): with a pitch ≡ 0 (the x tile's 320) or ≡ 4 (the
// intermediate's 260) the two rows share banks — PMC SQ_LDS_BANK_CONFLICT was 30 % of this kernel's LDS cycles
// (profiles/r03_e_pmc_c2.json).  The C = 32 path reads one row per lane group: any pitch.  (Same-box A/B of builds: 0.521-0.528 vs
// 0.517-0.521 ms per step — the conflicts were not what this kernel waits for; the layout stays because it is the right one.)
constexpr int RF_XW = RF_XS * 64; // staged columns of the x tile
constexpr int RF_XP = RF_XW + 16; // x tile pitch
constexpr int RF_TP = RF_TW + 16; // intermediate pitch
constexpr int RF_PD = 8;          // weight prefetch ring depth

// acc += sum over all (group, tap) units of W_unit x B(unit), B read from the LDS tile `bs` (pitch bp) at column
// col0 + tap * tstep; weights of unit u live at wp + lane offset + u * 1 KB.
// The weight ring lives OUTSIDE the GEMM: rf_prime puts a conv's first RF_PD units in flight before the phase that precedes it
// (conv1's before the x tile is staged, conv2's before conv1's epilogue and the barrier), so that the first MFMA of a GEMM never
// waits for a weight round trip of its own (ISA of round 2: the ring was primed behind the staging barrier).
struct RfRing { f32x4 ar[RF_PD]; int lu; };
__device__ __forceinline__ void rf_prime(RfRing& R, const float* wp, unsigned w_lane, int U) {
  R.lu = 0;
#pragma unroll
  for (int i = 0; i < RF_PD; ++i) {
    const int uc = R.lu < U ? R.lu : U - 1;
    R.ar[i] = rf_ld4(wp, w_lane + (unsigned)uc * 1024u);
    ++R.lu;
    __builtin_amdgcn_sched_barrier(0);
  }
}
__device__ __forceinline__ void rf_gemm(f32x16& acc, RfRing& R, const float* wp, unsigned w_lane, int groups, int k, const float* bs,
                                        int bp, int col0, int tstep, int lh) {
  const int U = groups * k;
  // two accumulators on alternate K steps: back-to-back MFMAs on ONE accumulator with anything issued in between stall the
  // pipe (~43 cycles each, MI355X_MICROARCH.md); summed on return
  f32x16 acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
  f32x4 (&ar)[RF_PD] = R.ar;
  int& lu = R.lu;
  auto load_unit = [&](int slot) __attribute__((always_inline)) {
    const int uc = lu < U ? lu : U - 1;                           // past the end: re-read the last unit, result unused
    ar[slot] = rf_ld4(wp, w_lane + (unsigned)uc * 1024u);
    ++lu;
  };
  // operands of unit u+1 are read from LDS before unit u's MFMAs are issued (software pipeline, order pinned below)
  int g = 0, j = 0;
  float bq[2][4];
  {
    const float* b = bs + lh * bp + col0;
    bq[0][0] = b[0]; bq[0][1] = b[2 * bp]; bq[0][2] = b[4 * bp]; bq[0][3] = b[6 * bp];
  }
  for (int u0 = 0; u0 < U; u0 += RF_PD) {
#pragma unroll
    for (int i = 0; i < RF_PD; ++i) {
      if (u0 + i < U) {
        int jn = j + 1, gn = g;
        if (jn == k) { jn = 0; ++gn; }
        const bool more = u0 + i + 1 < U;
        const float* b = bs + (8 * (more ? gn : g) + lh) * bp + col0 + (more ? jn : j) * tstep;
        bq[(i & 1) ^ 1][0] = b[0]; bq[(i & 1) ^ 1][1] = b[2 * bp]; bq[(i & 1) ^ 1][2] = b[4 * bp]; bq[(i & 1) ^ 1][3] = b[6 * bp];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[i].x, bq[i & 1][0], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[i].y, bq[i & 1][1], acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[i].z, bq[i & 1][2], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[i].w, bq[i & 1][3], acc2, 0, 0, 0);
        j = jn; g = gn;
      }
      load_unit(i);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
}

// C = 16: the 32-row MFMA would spend half of its rows on zero weights.  v_mfma_f32_16x16x4_f32 instead (16 rows x 16 columns, K = 4,
// 8 passes): a wave's 32 columns are two 16-column blocks (two independent accumulator chains), and ONE float4 per lane per tap
// feeds the tap's four K steps — lane (m = l & 15, kk = l >> 4) takes rows 0..15 of the (group kk >> 1, tap j, half kk & 1) piece of
// the ordinary fragment-ordered stream, i.e. channels 8(kk >> 1) + 2q + (kk & 1) for q = 0..3; MFMA q therefore pairs it with row
// 8(kk >> 1) + (kk & 1) + 2q of the LDS tile (any K-index <-> channel assignment works as long as A and B agree).  Half the MFMA
// cycles per tap of the 32x32x2 form, same number of LDS reads.   D layout: lane (n = l & 15, rg = l >> 4), register r -> row 4 rg + r.
__device__ __forceinline__ unsigned rf16_wlane(int k, int lane) {
  const int n = lane & 15, kk = lane >> 4;
  return 16u * (unsigned)(((kk >> 1) * k * 2 + (kk & 1)) * 32 + n);                       // + tap j * 1 KB
}
__device__ __forceinline__ void rf_gemm16(f32x4 (&acc)[2], RfRing& R, const float* wp, int k, const float* bs, int bp, int col0, int tstep,
                                          int lane) {
  const int n = lane & 15, kk = lane >> 4;
  const unsigned w_lane = rf16_wlane(k, lane);
  const float* b0 = bs + (8 * (kk >> 1) + (kk & 1)) * bp + col0 + n;
  f32x4 (&ar)[RF_PD] = R.ar;                      // primed by rf_prime(R, wp, rf16_wlane(k, lane), k): one unit per tap
  int& lu = R.lu;
  auto load_tap = [&](int slot) __attribute__((always_inline)) {
    const int uc = lu < k ? lu : k - 1;                             // past the end: re-read the last tap, result unused
    ar[slot] = rf_ld4(wp, w_lane + (unsigned)uc * 1024u);
    ++lu;
  };
  float bq[2][4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q) { bq[0][q][0] = b0[2 * q * bp]; bq[0][q][1] = b0[2 * q * bp + 16]; }
  for (int u0 = 0; u0 < k; u0 += RF_PD) {
#pragma unroll
    for (int i = 0; i < RF_PD; ++i) {
      if (u0 + i < k) {
        const float* b = b0 + (u0 + i + 1 < k ? u0 + i + 1 : u0 + i) * tstep;
#pragma unroll
        for (int q = 0; q < 4; ++q) { bq[(i & 1) ^ 1][q][0] = b[2 * q * bp]; bq[(i & 1) ^ 1][q][1] = b[2 * q * bp + 16]; }
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[i].x, bq[i & 1][0][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[i].x, bq[i & 1][0][1], acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[i].y, bq[i & 1][1][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[i].y, bq[i & 1][1][1], acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[i].z, bq[i & 1][2][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[i].z, bq[i & 1][2][1], acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[i].w, bq[i & 1][3][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[i].w, bq[i & 1][3][1], acc[1], 0, 0, 0);
      }
      load_tap(i);
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <bool C16>
__global__ void __launch_bounds__(512) resblock_fused_kernel(const FusedLaunch F) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                               // [C][RF_XP]   lrelu(x), zero outside [0, L)
  float* Tm = smem + 32 * RF_XP;                  // [C][RF_TP]   lrelu(conv1 + b1), zero outside [0, L)
  const FusedProb& P = F.p[blockIdx.z];
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;                    // timeline stamps (tools/timeline.py; F.dbg is null in the product)
  if (F.dbg) ts0 = __builtin_amdgcn_s_memtime();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * RF_BN;
  const int C = F.C, L = F.L, k = P.k, dil = P.dil;
  int Lv = L;                                     // valid length of this batch item (exact lengths), L stays the row stride
  if (F.lens) {
    const int64_t lv = F.lens[b] * F.len_mul;
    Lv = lv < L ? (int)lv : L;
    if (t0 >= Lv) return;
  }
  const int groups = (C + 7) / 8;
  const int h1 = ((k - 1) / 2) * dil, h2 = (k - 1) / 2;
  const float slope = F.slope;
  const float* xp = P.x + (int64_t)b * C * L;
  const unsigned w_lane = 16u * (unsigned)(lh * 32 + l31);
  RfRing ring;
  // conv1's first weight units fly while the x tile is staged
  if constexpr (C16) rf_prime(ring, P.w1, rf16_wlane(k, lane), k); else rf_prime(ring, P.w1, w_lane, groups * k);

  // ---- stage lrelu(x) for columns [t0 - LEAD - h1, t0 - LEAD - h1 + TW + 2*h1): wave w owns rows 4w..4w+3
  {
    const int tb = t0 - RF_LEAD - h1;
    const int XW = RF_TW + 2 * h1;
    float xr[4][RF_XS];
    unsigned tc[RF_XS];
    bool ok[RF_XS];
#pragma unroll
    for (int s = 0; s < RF_XS; ++s) {
      const int t = tb + lane + 64 * s;
      ok[s] = (lane + 64 * s < XW) && t >= 0 && t < Lv;
      tc[s] = 4u * (unsigned)(t < 0 ? 0 : (t >= Lv ? Lv - 1 : t));
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int row = wid * 4 + r;
      row = row < C ? row : C - 1;
#pragma unroll
      for (int s = 0; s < RF_XS; ++s) xr[r][s] = rf_ld(xp, (unsigned)row * 4u * (unsigned)L + tc[s]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int s = 0; s < RF_XS; ++s) {
        float v = xr[r][s];
        v = v < 0.f ? v * slope : v;
        Xs[(wid * 4 + r) * RF_XP + lane + 64 * s] = (ok[s] && wid * 4 + r < C) ? v : 0.f;
      }
  }
  __syncthreads();
  if (F.dbg) ts1 = __builtin_amdgcn_s_memtime();

  // ---- phase 1: intermediate columns [32*wid, 32*wid + 32)
  if constexpr (C16) {
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    rf_gemm16(acc, ring, P.w1, k, Xs, RF_XP, 32 * wid, dil, lane);
    if (wid < RF_BN / 32) rf_prime(ring, P.w2, rf16_wlane(k, lane), k);
    const int n = lane & 15, rg = lane >> 4;
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = P.b1[4 * rg + r];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int col = 32 * wid + 16 * nb + n;
      const int t = t0 - RF_LEAD + col;
      const bool tin = t >= 0 && t < Lv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[nb][r] + bv[r];
        v = v < 0.f ? v * slope : v;
        Tm[(4 * rg + r) * RF_TP + col] = tin ? v : 0.f;
      }
    }
  } else {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    rf_gemm(acc, ring, P.w1, w_lane, groups, k, Xs, RF_XP, 32 * wid + l31, dil, lh);
    if (wid < RF_BN / 32) rf_prime(ring, P.w2, w_lane, groups * k);   // conv2's first units fly under conv1's epilogue and the barrier
    // conv1's 16 bias rows into registers in ONE batch before the first LDS store: read as P.b1[row] inside the store loop, every
    // row re-loaded the pointer from the kernarg segment (the store in between may alias it, as far as the compiler knows) and
    // waited for its own load — 16 serial s_load + global_load round trips between the two GEMMs of every workgroup (ISA of
    // round 2).  (Loaded before the GEMM they would cost 16 registers across it: 140 in all, one workgroup per CU instead of two.)
    // Two batches of 8, not one of 16: 16 values in flight put the kernel at 129 registers — one over the 128 that let two workgroups
    // share a CU — and squeezing it back with __launch_bounds__(512, 4) spilled two registers to scratch, which costs far more than
    // it saves: a kernel with a scratch segment drains the queue at dispatch (config 2: 4.59 -> 4.84 ms per step with this one
    // kernel spilling, while the sum of kernel times FELL by 0.12 ms).
    const float* const b1p = P.b1;
    const int t = t0 - RF_LEAD + 32 * wid + l31;
    const bool tin = t >= 0 && t < Lv;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      float b1v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        int row = (r & 3) + 8 * ((r + 8 * hb) >> 2) + 4 * lh;
        row = row < C ? row : C - 1;
        b1v[r] = rf_ld(b1p, 4u * (unsigned)row);
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int rr = r + 8 * hb;
        const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
        if (row < C) {
          float v = acc[rr] + b1v[r];
          v = v < 0.f ? v * slope : v;
          Tm[row * RF_TP + 32 * wid + l31] = tin ? v : 0.f;
        }
      }
    }
  }
  __syncthreads();
  if (F.dbg) ts2 = __builtin_amdgcn_s_memtime();

  // ---- phase 2: output columns [32*wid, 32*wid + 32) of the tile, waves 0..6
  if (C16 && wid < RF_BN / 32) {
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    rf_gemm16(acc, ring, P.w2, k, Tm, RF_TP, RF_LEAD + 32 * wid - h2, 1, lane);
    const int n = lane & 15, rg = lane >> 4;
    float* const op = P.out + (int64_t)b * C * L;
    const float* const b2p = P.b2;
    float bv[4], xv[2][4];
    int tt[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = b2p[4 * rg + r];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      tt[nb] = t0 + 32 * wid + 16 * nb + n;
      const int tc = tt[nb] < L ? tt[nb] : L - 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) xv[nb][r] = rf_ld(xp, 4u * ((unsigned)(4 * rg + r) * (unsigned)L + (unsigned)tc));
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (tt[nb] < L) op[(unsigned)(4 * rg + r) * (unsigned)L + (unsigned)tt[nb]] = (acc[nb][r] + bv[r]) + xv[nb][r];
  }
  if (!C16 && wid < RF_BN / 32) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    rf_gemm(acc, ring, P.w2, w_lane, groups, k, Tm, RF_TP, RF_LEAD + 32 * wid + l31 - h2, 1, lh);
    const int t = t0 + 32 * wid + l31;
    if (t < L) {
      // pointers and the 16 bias / residual values into registers BEFORE the first store: P lives in the kernarg segment and is
      // re-loaded after every global store otherwise (see conv_mfma.hip's epilogue).  (Loading them before the GEMM instead — under
      // its MFMAs — costs 32 registers across it: 156 in all, one workgroup per CU instead of two.)
      float* const op = P.out + (int64_t)b * C * L;
      const float* const b2p = P.b2;
      float bv[16], xv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        row = row < C ? row : C - 1;
        bv[r] = b2p[row];
        xv[r] = rf_ld(xp, 4u * ((unsigned)row * (unsigned)L + (unsigned)t));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < C) op[(unsigned)row * (unsigned)L + (unsigned)t] = (acc[r] + bv[r]) + xv[r];   // conv + bias, then + x
      }
    }
  }
  if (F.dbg && tid == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long* d = F.dbg + 8ull * (((unsigned long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
    d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = __builtin_amdgcn_s_memtime();
    d[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    d[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    d[6] = (unsigned long long)k; d[7] = 1;
  }
}

bool resblock_fused_supported(int C, int k, int dil) {
  return C >= 8 && C <= 32 && C % 8 == 0 && k % 2 == 1 && (k - 1) / 2 <= RF_LEAD && RF_TW + (k - 1) * dil <= RF_XW &&
         RF_LEAD + RF_BN + (k - 1) / 2 <= RF_TW;
}

int launch_resblock_fused(hipStream_t stream, const FusedLaunch& F) {
  if (F.nprob < 1 || F.nprob > 3 || F.B < 1 || F.L < 1) return -1;
  for (int i = 0; i < F.nprob; ++i)
    if (!resblock_fused_supported(F.C, F.p[i].k, F.p[i].dil)) return -2;
  if ((int64_t)F.C * F.L >= (1ll << 29)) return -2;              // 32-bit byte offsets inside one batch item
  dim3 grid((F.L + RF_BN - 1) / RF_BN, F.B, F.nprob);
  const size_t lds = sizeof(float) * (size_t)(32 * RF_XP + 32 * RF_TP);
  auto kern = F.C == 16 ? resblock_fused_kernel<true> : resblock_fused_kernel<false>;
  ensure_dyn_lds((const void*)kern, lds);
  FusedLaunch Ft = F;
  {
    int ks = 0;
    for (int i = 0; i < F.nprob; ++i) ks |= (F.p[i].k & 255) << (8 * i);
    Ft.dbg = timeline_slice(grid.x, grid.y, grid.z, 88000 + F.C, ks, F.C, F.L);       // tile id 88xxx: fused ResBlock pair
  }
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, Ft);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

double resblock_fused_flops(const FusedLaunch& F) {
  double f = 0;
  for (int i = 0; i < F.nprob; ++i) f += 2.0 * (2.0 * F.C * F.C * F.p[i].k) * (double)F.L * F.B;
  return f;
}

double resblock_fused_bytes(const FusedLaunch& F) {   // x read once (+ once more for the residual), out written once
  double by = 0;
  for (int i = 0; i < F.nprob; ++i) by += 4.0 * (3.0 * F.C * (double)F.L * F.B + 2.0 * F.C * F.C * F.p[i].k);
  return by;
}

}  // namespace bv2

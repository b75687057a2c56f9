// conv_mfma.hip — conv1d (any k / dilation / padding, incl. 1x1 and the polyphase branches of ConvTranspose1d) as an
// implicit GEMM on the gfx950 fp32 matrix core (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains at the 157 TF vector rate).
//
//   GEMM view:  M = C_out (rows, from the packed weight),  N = time (columns, contiguous in HBM),  K = (tap j, C_in).
//
// Packed weight ("fragment order", written once by bv2_model.cpp):
//      Wp[m-tile = co/32][group g = ci/8][tap j][lh = ci&1][co%32][q = (ci%8)/2]
//   -> the four floats a lane needs as MFMA A operand (A[m = l&31][kk = l>>5]) for four consecutive K steps of one
//      8-channel group are ONE aligned float4 (one 16-byte global load per lane per 4 MFMAs, 1 KB per wave = one "unit"),
//      and the weight stream of one 32-row output tile is contiguous (sequential reads: no L2-channel camping).
//
// Two kernels:
//  * conv1d_mfma_kernel  — LDS-tiled, for problems with enough columns to fill the chip (the Generator, and every conv at
//    large batch): weights global -> register ring, X chunk in LDS re-used by all k taps, one barrier per C_in chunk
//    (details at the kernel).
//  * conv1d_splitk_kernel — for the small-N problems of the text encoder / flow / duration predictors at small batch
//    (N = T or T_y columns, a few hundred): 32x32 output tile per workgroup, the 4 waves split K (channel groups) and
//    reduce through LDS; optionally K is also split ACROSS workgroups into `ksplit` partial slabs that the consumer
//    (LayerNorm / embed kernel) sums.  Operands go global -> registers directly (4-deep prefetch ring), no barriers in
//    the main loop.  Workgroups that share a weight slice are placed on the same XCD (block b runs on XCD b % 8).
//
// Replaces (reference): every Conv1d of modules.ResBlock1 (modules.py:296-309), Generator.conv_pre / ups
// (models.py:539-545), attentions.FFN (attentions.py:438-446), q/k/v/o and all 1x1 projections, DurationPredictor
// convs (models.py:285-299), WN in/res_skip layers (modules.py:192-210).
#include <hip/hip_runtime.h>
#include "../bv2_kernels.h"

namespace bv2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: HIP's float4 struct defeats SROA here

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

// load base[byte_off]: wave-uniform base (SGPR pair) + 32-bit per-lane BYTE offset -> the `global_load v, v_off, s[base]`
// addressing form (one VGPR per address instead of a 64-bit pair)
__device__ __forceinline__ float ld_off(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ f32x4 ld_off4(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byte_off);
}

// ---------------------------------------------------------------------------------------------------------------
// conv1d_mfma_kernel — LDS-tiled implicit-GEMM conv for problems with enough columns to fill the chip (the Generator, and
// every conv at large batch).  Workgroup = 4 waves = WM x WN, wave tile = MI x NI blocks of 32x32, BM x BN outputs.
//   * weights never touch LDS: every wave streams the A fragments of its own MI m-tiles global -> registers through a ring
//     with one slot per channel group of the chunk (slot g: unit (g, tap j) is followed by (g, j+1), after the last tap by
//     (g, 0) of the next chunk — the ring is never drained);
//   * the X chunk [CK][BN + halo] is the only LDS tenant (double buffered, register-prefetched one chunk ahead; the
//     pre-activation / 3-way branch mean / input mask are applied once per element while it is written), re-used by all k
//     taps, and the workgroup synchronises ONCE PER CHUNK (GR*k units = 16*k MFMAs per wave);
//   * inside a chunk the units run tap-major (all groups of tap j, then tap j+1): LDS rows are immediate offsets, a tap is
//     one VGPR add, and the next unit's B operands are read while the current unit's MFMAs run (order pinned with
//     sched_group_barrier: 63 instructions per 16 MFMAs).
// (An earlier form staged the weights through LDS with a barrier every 16 MFMAs: 113 instructions per 16 MFMAs, 92 TF where
// this one reaches 103 TF on the C=128 stage at batch 1; tools/probe/mfma_probe.hip shows why — every VALU / LDS-dependent
// instruction between MFMAs costs issue slots the 64-cycle fp32 MFMA cannot hide at 1-3 waves per SIMD.)
// (Measured and not kept: __launch_bounds__(256, 5) for the 64x64 tile — 96 registers, five workgroups per CU instead of four,
// 6 spilled registers: the Generator at batch 1 went 2.819 -> 2.856 ms.)
template <int WM, int WN, int MI, int NI, int CK, int XS>
__global__ void __launch_bounds__(256) conv1d_mfma_kernel(const ConvLaunch L, const int mtiles, const int per_xcd, const int snake_n) {
  constexpr int BM = WM * MI * 32;
  constexpr int BN = WN * NI * 32;
  constexpr int XP = XS * 64;
  constexpr int RPW = CK / 4;
  constexpr int GR = CK / 8;                      // channel groups per chunk = ring depth (GR*k units per chunk: whole rings)
  static_assert(WM * WN == 4, "4 waves per workgroup");
  extern __shared__ __attribute__((aligned(16))) float smem[];

  // Placement.  Default: grid (time tiles, m-tiles x batch, problems) — dispatch order = problems in launch order (the host puts
  // the most expensive first), so when there are several times more workgroups than slots the long ones start first.
  // snake_n > 0 (every workgroup of the launch is resident at once, <= 4 per CU): dispatch is a plain round-robin — workgroup
  // p of an XCD lands on CU p % 32 — so WHICH problems meet on a CU is decided by the order alone.  With the problems sorted by
  // cost, round r takes its 32 workgroups in ascending order for even r and descending for odd r ("snake"): the CU that got an
  // expensive tile in one round gets a cheap one in the next (C2 stage 0, k = 11/7/3 on 576 workgroups: worst CU 21 -> 18 cost
  // units at a mean of 15.75; tools/timeline.py).
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
    bx = ((rest - by * per_xcd) << 3) | xcd;                    // decoded below exactly like the default form
  }
  const ConvProb P = L.p[pz];                    // BY VALUE: the descriptor's scalar loads are issued together, here (see the split-K kernel)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;                    // timeline stamps (tools/timeline.py; L.dbg is null in the product)
  if (L.dbg) ts0 = __builtin_amdgcn_s_memtime();
  const int b = by / mtiles;
  const int m0 = (by - b * mtiles) * BM;
  const int vt = per_xcd ? (bx & 7) * per_xcd + (bx >> 3) : bx;
  const int t0 = vt * BN;
  if (t0 >= L.L) return;
  if (m0 >= P.cout_pad) return;

  const int k = P.k, dil = P.dil, cin = P.cin, nsrc = P.nsrc;
  int Lin = P.Lin;
  if (L.lens) {
    const int64_t lv = L.lens[b] * L.len_mul;
    Lin = lv < Lin ? (int)lv : Lin;
    if (t0 >= Lin) return;
  }
  const float in_scale = P.in_scale, slope = P.slope;
  const bool lrelu = P.pre_act == PRE_LRELU;
  const float* const x0p = P.x[0] + (int64_t)b * P.x_bstride;
  const float* const x1p = P.x[1] ? P.x[1] + (int64_t)b * P.x_bstride : nullptr;
  const float* const x2p = P.x[2] ? P.x[2] + (int64_t)b * P.x_bstride : nullptr;
  const float* const maskp = P.in_mask ? P.in_mask + (int64_t)b * P.in_mask_bstride : nullptr;
  const int x_rstride = P.x_rstride;
  const int XW = BN + (k - 1) * dil;
  const int nchunks = P.cin_pad / CK;
  const int groups = P.cin_pad / 8;
  float* Xs = smem;                               // [nchunks > 1 ? 2 : 1][CK][XP]

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // ---- weight ring: unit (group g, tap j) of m-tile mt lives at byte offset (((mt*groups + g)*k + j)*64 + lh*32 + l31)*16
  f32x4 ar[GR][MI];
  // ring slot g always holds group g of the current chunk: unit (g, j) is followed in the slot by (g, j+1), and after the
  // last tap by (g, 0) of the next chunk.  One wave-uniform pointer per (slot, m-tile) + the per-lane offset.
  const float* wq[GR][MI];
  const unsigned wlane = 16u * (unsigned)(lh * 32 + l31);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    int mt = (m0 >> 5) + wm * MI + mi;
    mt = mt * 32 < P.w_ld ? mt : (m0 >> 5);       // rows beyond the allocation: any valid tile (results are dropped)
#pragma unroll
    for (int g = 0; g < GR; ++g) wq[g][mi] = P.w + ((int64_t)mt * groups + g) * k * 256;
  }
  auto load_unit = [&](int slot, int step_floats) __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      ar[slot][mi] = ld_off4(wq[slot][mi], wlane);
      wq[slot][mi] += step_floats;
    }
  };

  // ---- X prefetch (identical to v1)
  float xr[RPW][XS];
  float xm[XS];
  const int tbase = t0 - P.pad_left;
  unsigned tc[XS];
  float colsc[XS];
#pragma unroll
  for (int s = 0; s < XS; ++s) {
    const int t = tbase + lane + 64 * s;
    const bool tok = (lane + 64 * s < XW) && t >= 0 && t < Lin;
    colsc[s] = tok ? in_scale : 0.f;
    tc[s] = 4u * (unsigned)(t < 0 ? 0 : (t >= Lin ? Lin - 1 : t));
  }
  auto issue_x = [&](int c) __attribute__((always_inline)) {
    unsigned roff[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      int cg = c * CK + wid * RPW + r;
      cg = cg < cin ? cg : cin - 1;
      roff[r] = 4u * (unsigned)cg * (unsigned)x_rstride;
    }
    if (maskp) {
#pragma unroll
      for (int s = 0; s < XS; ++s) xm[s] = ld_off(maskp, tc[s]);
    } else {
#pragma unroll
      for (int s = 0; s < XS; ++s) xm[s] = 1.f;
    }
#pragma unroll
    for (int s = 0; s < XS; ++s)
#pragma unroll
      for (int r = 0; r < RPW; ++r) xr[r][s] = ld_off(x0p, roff[r] + tc[s]);
    if (nsrc > 1) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < XS; ++s)
#pragma unroll
        for (int r = 0; r < RPW; ++r) xr[r][s] += ld_off(x1p, roff[r] + tc[s]);
      if (nsrc > 2) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < XS; ++s)
#pragma unroll
          for (int r = 0; r < RPW; ++r) xr[r][s] += ld_off(x2p, roff[r] + tc[s]);
      }
    }
  };
  auto store_x = [&](int c, int buf) __attribute__((always_inline)) {
    float* dst = Xs + buf * (CK * XP) + (wid * RPW) * XP + lane;
    const int rows_ok = cin - (c * CK + wid * RPW);
#pragma unroll
    for (int s = 0; s < XS; ++s) {
      const float sc = colsc[s] * xm[s];
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        float v = xr[r][s];
        const float vn = v * slope;
        v = (lrelu && v < 0.f) ? vn : v;
        v *= sc;
        dst[r * XP + 64 * s] = r < rows_ok ? v : 0.f;
      }
    }
  };

  // ConvProb::omax (max |out| for the x3 form of conv_x6.hip): this wave's word of the slot, read now so that its round trip lands
  // under the prologue's loads (read in front of the atomic it was an exposed round trip at the end of every workgroup)
  unsigned* const omax_w = P.omax ? x3_slot_word(P.omax) : nullptr;
  unsigned omax_seen = 0xffffffffu;
  if (omax_w) omax_seen = *reinterpret_cast<volatile unsigned*>(omax_w);
  // prologue: X chunk 0 first (its latency is the long one), then prime the ring
  issue_x(0);
#pragma unroll
  for (int i = 0; i < GR; ++i) { load_unit(i, k == 1 ? (nchunks > 1 ? GR * 256 : 0) : 256); __builtin_amdgcn_sched_barrier(0); }
  store_x(0, 0);
  omax_seen = __builtin_amdgcn_readfirstlane(omax_seen);
  __syncthreads();
  if (L.dbg) ts1 = __builtin_amdgcn_s_memtime();

  // Unit order inside a chunk: tap-major — for every tap j the GR groups in turn (ring slot = group, a compile-time index;
  // rows 8g of the X chunk are immediate offsets, the tap's column shift is one VGPR add per tap).  Operands of the next
  // unit are read from LDS while the current unit's MFMAs run.
  const unsigned xlane = 4u * (unsigned)(lh * XP + wn * (NI * 32) + l31);
  const unsigned tap_step = 4u * (unsigned)dil;
  const char* const xs_bytes = reinterpret_cast<const char*>(Xs);
  const int last_step = ((GR - 1) * k + 1) * 256;                   // floats from (g, k-1) to (g, 0) of the next chunk
  for (int c = 0; c < nchunks; ++c) {
    const bool next_chunk = (c + 1) < nchunks;
    if (next_chunk) issue_x(c + 1);               // in flight under this chunk's MFMAs
    unsigned xcol = xlane + (unsigned)(c & 1) * (unsigned)(CK * XP * 4);
    // the units loaded during the LAST chunk's last tap have no successor: wrap the pointer back to the tile's first chunk
    // (valid memory, values unused) instead of branching around the load
    const int jump = next_chunk ? last_step : -(((nchunks - 1) * GR * k + (k - 1)) * 256);
    float bb[2][4][NI];
    {
      const char* xb = xs_bytes + xcol;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bb[0][q][ni] = *reinterpret_cast<const float*>(xb + 4 * ((2 * q) * XP + ni * 32));
    }
    for (int j = 0; j < k; ++j) {
      const unsigned xnext = (j + 1 < k) ? xcol + tap_step : xcol;   // the chunk's last unit re-reads itself (unused)
      // after consuming (g, j) the slot is refilled with (g, j+1); the unit loaded while j == k-1 is (g, 0) of the next chunk
      // and the one loaded while j == k-2 is (g, k-1): the pointer must jump AFTER that one
      const int step_after = (j + 2 == k || k == 1) ? jump : 256;
#pragma unroll
      for (int g = 0; g < GR; ++g) {
        {
          const char* xb = xs_bytes + (g + 1 < GR ? xcol : xnext);
          const int grow = g + 1 < GR ? 8 * (g + 1) : 0;
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              bb[(g & 1) ^ 1][q][ni] = *reinterpret_cast<const float*>(xb + 4 * ((grow + 2 * q) * XP + ni * 32));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const float a = q == 0 ? ar[g][mi].x : (q == 1 ? ar[g][mi].y : (q == 2 ? ar[g][mi].z : ar[g][mi].w));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb[g & 1][q][ni], acc[mi][ni], 0, 0, 0);
          }
        load_unit(g, step_after);
        // pin the emitted order: next unit's LDS reads FIRST (they land under this unit's MFMAs; left alone the scheduler
        // sinks them below the MFMAs and every unit then starts with an exposed LDS round trip), MFMAs, ring load
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * NI, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * MI * NI, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, MI, 0);
        __builtin_amdgcn_sched_barrier(0);        // keep program order: the ring's vmcnt distances stay GR - 1 units
      }
      xcol = xnext;
    }
    if (next_chunk) {
      store_x(c + 1, (c + 1) & 1);
      __syncthreads();
    }
  }

  if (L.dbg) ts2 = __builtin_amdgcn_s_memtime();
  // ---- epilogue.  Every ConvProb field is copied into a register FIRST: P lives in the kernarg segment, and as soon as the
  // kernel has stored to global memory the compiler must assume the kernarg may have changed, so reading P.* inside the store
  // loop re-loads it per row (measured: 164 scalar loads / 198 waits, 18-45k cycles per workgroup — a quarter of the kernel).
  // Offsets are 32-bit element offsets from wave-uniform 64-bit bases (a batch item is < 2^31 floats); the 16 residual / bias
  // loads of an accumulator tile are issued together, then the arithmetic, then the 16 stores.
  {
    const int cout = P.cout, Lout = L.L;
    const int act = P.act, mask_pre = P.mask_pre, mask_post = P.mask_post, res_mode = P.res_mode;
    const unsigned o_rs = (unsigned)P.out_rstride, o_ts = (unsigned)P.out_tstride, o_to = (unsigned)P.out_toff;
    float* const outb = P.out + (int64_t)b * P.out_bstride;
    const float* const resb = res_mode != RES_NONE ? P.res + (int64_t)b * P.res_bstride : nullptr;
    const float* const biasp = P.bias;
    const float* const bias2p = P.bias2 ? P.bias2 + (int64_t)b * P.bias2_bstride : nullptr;
    const float* const omaskp = P.out_mask ? P.out_mask + (int64_t)b * P.out_mask_bstride : nullptr;
    // bias vectors: a missing one reads a valid dummy address and is masked to +0.0 bit-wise, so that all 32 loads of a lane are
    // unconditional and in flight together.  As `biasp ? biasp[row] : 0` + `if (bias2p) v += bias2p[row]` every row was a pair of
    // loads followed by s_waitcnt vmcnt(0): 16 SERIAL memory round trips at the start of every workgroup's epilogue (ISA of round 2;
    // tools/timeline.py: epilogue 9-18k cycles of an 80-175k-cycle workgroup)
    const float* const b1p = biasp ? biasp : P.w;
    const float* const b2p = bias2p ? bias2p : P.w;
    const unsigned m_b1 = biasp ? 0xffffffffu : 0u, m_b2 = bias2p ? 0xffffffffu : 0u;
    float vmx = 0.f;                               // max |v| of everything stored, for the x3 form of conv_x6.hip that reads this tensor
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int row0 = m0 + wm * (MI * 32) + mi * 32 + 4 * lh;       // this lane's rows: row0 + (r & 3) + 8 * (r >> 2)
      float bs[16], bs2[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = row0 + (r & 3) + 8 * (r >> 2);
        row = row < cout ? row : cout - 1;                            // clamped: the load is unconditional, the store is not
        bs[r] = ld_off(b1p, m_b1 ? 4u * (unsigned)row : 0u);
        bs2[r] = ld_off(b2p, m_b2 ? 4u * (unsigned)row : 0u);
      }
      // (combined at their use below: the residual loads of the tile are issued first, one round trip for both)
      auto bsum = [&](int r) __attribute__((always_inline)) {
        return __uint_as_float(__float_as_uint(bs[r]) & m_b1) + __uint_as_float(__float_as_uint(bs2[r]) & m_b2);
      };
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int col = t0 + wn * (NI * 32) + ni * 32 + l31;
        const bool colok = col < Lout;
        const int colc = colok ? col : Lout - 1;
        const float om = omaskp ? omaskp[colc] : 1.f;
        if (act == ACT_GATE) {                     // fused_add_tanh_sigmoid_multiply: rows r and r + 8 of a lane are a (tanh, sigmoid) pair
          const unsigned goff0 = (unsigned)((row0 - 4 * lh) / 2 + 4 * lh) * o_rs + (unsigned)colc * o_ts + o_to;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            const float a = acc[mi][ni][r] + bsum(r), sg = acc[mi][ni][r + 8] + bsum(r + 8);
            const float v = tanhf(a) * (1.f / (1.f + expf(-sg)));
            if (colok && row0 + dr + 16 < cout) outb[goff0 + (unsigned)dr * o_rs] = v;
          }
          continue;
        }
        const unsigned off0 = (unsigned)row0 * o_rs + (unsigned)colc * o_ts + o_to;
        float rv[16];
        if (resb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            const unsigned off = row0 + dr < cout ? off0 + (unsigned)dr * o_rs : off0;
            rv[r] = ld_off(resb, 4u * off);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          float v = acc[mi][ni][r] + bsum(r);
          if (act == ACT_RELU) v = v < 0.f ? 0.f : v;     // a select: NaN stays NaN (torch.relu), v_max would return 0
          else if (act == ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
          if (mask_pre) v *= om;
          if (res_mode == RES_ADD) v += rv[r];
          else if (res_mode == RES_RSUB) v = rv[r] - v;
          if (mask_post) v *= om;
          if (colok && row0 + dr < cout) {
            outb[off0 + (unsigned)dr * o_rs] = v;
            vmx = fmaxf(vmx, fabsf(v));
          }
        }
      }
    }
    if (omax_w) x3_publish(omax_w, omax_seen, vmx, lane);
  }
  if (L.dbg && tid == 0) {
    __builtin_amdgcn_s_waitcnt(0);                                  // the epilogue's stores have been issued and acknowledged
    unsigned long long* d = L.dbg + 8ull * (snake_n > 0 ? (unsigned long long)blockIdx.x
                                                        : ((unsigned long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
    d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = __builtin_amdgcn_s_memtime();
    d[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);               // HW_ID
    d[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);              // XCC_ID
    d[6] = (unsigned long long)k; d[7] = 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// split-K kernel for small-N problems
constexpr int SK_PD = 8;          // prefetch ring depth (units of 4 MFMAs)


// LDSX (k > 1): the workgroup's X tile [its channel groups][32 + (k-1)*dil columns] is staged ONCE in LDS (mask / pre-activation /
// in_scale applied while it is written) and every tap reads it back shifted, instead of every wave re-loading each column per
// tap from global memory: tools/timeline.py showed the k = 5 FFN convs bound by the CU's L1 path (246 KB per workgroup, half
// of it the 5-fold re-read of X), their main loop at 3.7x the MFMA time.  Weights still stream global -> registers.
constexpr int SK_XP = 64;         // staged tile width (floats): 32 + (k-1)*dil must fit
constexpr int SK_RPW = 32;        // rows of the staged tile per wave at most
template <bool MASK, int NWV, bool LDSX>    // NWV waves per workgroup split K inside the workgroup
__global__ void __launch_bounds__(64 * NWV) conv1d_splitk_kernel(const ConvLaunch L, const int mtiles, const int ntiles,
                                                                 const int per_xcd, const int total) {
  extern __shared__ __attribute__((aligned(16))) float red_raw[];
  float (*red)[32][33] = reinterpret_cast<float (*)[32][33]>(red_raw);   // [NWV][32][33]; LDSX: aliases the X tile (barrier between)
  // XCD-aware placement: consecutive virtual ids (which share a weight slice) land on the same XCD
  const int bid = blockIdx.x;
  if (bid >= per_xcd * 8) {                                         // spare workgroups: the next launch's weights into this XCD's L2
    prefetch_tail(L.pf, (unsigned)bid, (unsigned)(bid - per_xcd * 8), threadIdx.x, 64 * NWV);
    return;
  }
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;                    // timeline stamps (tools/timeline.py; L.dbg is null in the product)
  if (L.dbg) ts0 = __builtin_amdgcn_s_memtime();
  const int v = (bid & 7) * per_xcd + (bid >> 3);
  if (v >= total) return;
  int rem = v;
  const int nt = rem % ntiles; rem /= ntiles;
  const int z = rem % L.ksplit; rem /= L.ksplit;
  const int mt = rem % mtiles; rem /= mtiles;
  const int b = rem % L.B; rem /= L.B;
  // The problem descriptor BY VALUE: every field's scalar load is issued here, together (one kernarg round trip).  Through a
  // reference the loads sat at their first uses — ~15 dependent s_load / s_waitcnt pairs spread over the prologue (ISA of round 2)
  // in front of the first global load of a kernel whose whole life is ~14k cycles.
  const ConvProb P = L.p[rem];
  const int m0 = mt * 32, t0 = nt * 32;
  if (m0 >= P.cout_pad) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps everything derived from it in SGPRs
  const int l31 = lane & 31, lh = lane >> 5;
  // problem fields hoisted into registers (P lives in the kernarg segment)
  const int k = P.k, dil = P.dil, cin = P.cin;
  int Lin = P.Lin;
  if (L.lens) {
    const int64_t lv = L.lens[b] * L.len_mul;
    Lin = lv < Lin ? (int)lv : Lin;
  }
  const int groups = P.cin_pad / 8;
  const float in_scale = P.in_scale, slope = P.slope;
  const bool lrelu = P.pre_act == PRE_LRELU;
  const float* const xp = P.x[0] + (int64_t)b * P.x_bstride;       // wave-uniform bases, 32-bit byte offsets per lane
  const float* const mp = MASK ? P.in_mask + (int64_t)b * P.in_mask_bstride : nullptr;
  const float* const wp = P.w;
  const unsigned x_rs4 = 4u * (unsigned)P.x_rstride;
  const unsigned w_tap = 1024u;                                      // bytes between consecutive taps of one group
  const unsigned w_unit = w_tap * (unsigned)k;                      // bytes between consecutive channel groups
  const unsigned w_lane = 16u * (unsigned)(lh * 32 + l31) + (unsigned)mt * (unsigned)groups * w_unit;   // this m-tile's stream
  // channel groups of this (slice z, wave wid): contiguous range, balanced
  // (32-bit: groups * slices < 2^31; the 64-bit form was three ~130-instruction software divisions in the prologue)
  const unsigned nsl = (unsigned)(L.ksplit * NWV), sl = (unsigned)(z * NWV + wid);
  const int g0 = (int)(((unsigned)groups * sl) / nsl), g1 = (int)(((unsigned)groups * (sl + 1u)) / nsl);
  const int U = (g1 - g0) * k;                   // units of (group, tap) = 4 MFMAs each

  f32x16 acc, acc2;                               // two accumulators on alternate K steps: see conv1d_mfma_kernel
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }

  const int tcol = t0 + l31 - P.pad_left;
  f32x4 ar[SK_PD];
  float br[SK_PD][4];
  float mr[SK_PD], vmr[SK_PD];                    // raw mask value / validity*in_scale of the unit's column

  // Loads are UNCONDITIONAL (also past the last unit: clamped to a valid address, result unused) and nothing here reads a
  // loaded value, so the compiler's vmcnt bookkeeping is exact and the ring really keeps SK_PD units in flight; padding
  // columns are zeroed by vmr, padded channels meet zero weights.
  int lg = g0, lj = 0;                            // next unit to LOAD
  const unsigned row_max = (unsigned)(cin - 1) * x_rs4;            // byte offset of the last real channel row
  const unsigned x_rs8 = 2u * x_rs4;
  auto group_off = [&](int g) __attribute__((always_inline)) {
    const int gc = g < groups ? g : groups - 1;
    return (unsigned)(8 * gc + lh) * x_rs4;
  };
  auto weight_off = [&](int g) __attribute__((always_inline)) {
    const int gc = g < groups ? g : groups - 1;
    return w_lane + (unsigned)gc * w_unit;
  };
  unsigned goff = group_off(lg), woff = weight_off(lg), joff = 0;   // joff = lj * w_tap
  auto load_unit = [&](int slot) __attribute__((always_inline)) {
    ar[slot] = ld_off4(wp, woff + joff);
    if (LDSX) {
      joff += w_tap;
      if (++lj == k) { lj = 0; joff = 0; ++lg; woff = weight_off(lg); }
      return;
    }
    const int t = tcol + lj * dil;
    const bool tok = t >= 0 && t < Lin;
    const unsigned tcl = 4u * (unsigned)(t < 0 ? 0 : (t >= Lin ? Lin - 1 : t));
    vmr[slot] = tok ? in_scale : 0.f;
    mr[slot] = MASK ? ld_off(mp, tcl) : 1.f;
    unsigned ro = goff;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned rc = ro < row_max ? ro : row_max;             // padded channels re-read the last real row (x 0 weight)
      br[slot][q] = ld_off(xp, rc + tcl);
      ro += x_rs8;
    }
    joff += w_tap;
    if (++lj == k) { lj = 0; joff = 0; ++lg; goff = group_off(lg); woff = weight_off(lg); }
  };

  // ---- epilogue operands first: every ConvProb field into a register (reading P.* after the first global store re-loads it
  // from the kernarg segment per row, see conv1d_mfma_kernel's epilogue), 32-bit element offsets from wave-uniform bases, and
  // the bias / residual values of this thread's output elements in flight BEFORE the main loop — loaded after the LDS reduction
  // they were one more exposed memory round trip in every one of the ~130 split-K launches of a batch-1 step
  const int col = t0 + (tid & 31);
  const bool colok = col < L.L;
  const int cout = P.cout, act = P.act, mask_pre = P.mask_pre, mask_post = P.mask_post, res_mode = P.res_mode;
  const unsigned o_rs = (unsigned)P.out_rstride, o_ts = (unsigned)P.out_tstride, o_to = (unsigned)P.out_toff;
  float* const outb = P.out + (int64_t)z * L.slab_stride + (int64_t)b * P.out_bstride;
  const float* const resb = (res_mode != RES_NONE && z == 0) ? P.res + (int64_t)b * P.res_bstride : nullptr;
  const float* const biasp = z == 0 ? P.bias : nullptr;
  const float* const bias2p = (z == 0 && P.bias2) ? P.bias2 + (int64_t)b * P.bias2_bstride : nullptr;
  const float om = (P.out_mask && colok) ? P.out_mask[(int64_t)b * P.out_mask_bstride + col] : 1.f;
  unsigned* const omaxp = P.omax ? x3_slot_word(P.omax) : nullptr;
  const unsigned coff = (unsigned)(colok ? col : 0) * o_ts + o_to;
  constexpr int RPP = 2 * NWV;                      // rows per pass (one element per thread per pass)
  float rvv[32 / RPP], bsv[32 / RPP], b2v[32 / RPP];
#pragma unroll
  for (int i = 0; i < 32 / RPP; ++i) {
    const int rl = (tid >> 5) + RPP * i;
    int row = m0 + rl;
    row = row < cout ? row : cout - 1;
    bsv[i] = biasp ? biasp[row] : 0.f;
    b2v[i] = bias2p ? bias2p[row] : 0.f;          // summed in the epilogue: an add here makes the load wait for itself
    rvv[i] = resb ? ld_off(resb, 4u * ((unsigned)row * o_rs + coff)) : 0.f;
  }
  // ---- LDSX: stage the workgroup's X tile.  Row rr = wid + NWV*i of the tile is channel 8*G0 + rr; lane = column.
  const int G0 = (int)(((unsigned)groups * (unsigned)(z * NWV)) / nsl);
  float* const Xs = red_raw;
  if (LDSX) {
    const int G1 = (int)(((unsigned)groups * (unsigned)((z + 1) * NWV)) / nsl);
    const int nrows = 8 * (G1 - G0);
    const int XW = 32 + (k - 1) * dil;
    const int t = t0 - P.pad_left + lane;
    const bool tok = lane < XW && t >= 0 && t < Lin;
    const unsigned tcl = 4u * (unsigned)(t < 0 ? 0 : (t >= Lin ? Lin - 1 : t));
    // the mask value is loaded UNCONDITIONALLY (clamped address) and only used after the X loads are in flight: inside the
    // `tok ? ... : 0` select it was a load + s_waitcnt vmcnt(0) of its own — one exposed memory round trip per masked launch
    const float mval = MASK ? ld_off(mp, tcl) : 1.f;
    float xv[SK_RPW];
#pragma unroll
    for (int i = 0; i < SK_RPW; ++i) {
      int row = 8 * G0 + wid + NWV * i;
      row = row < cin ? row : cin - 1;
      xv[i] = ld_off(xp, (unsigned)row * x_rs4 + tcl);
    }
    const float cs = tok ? in_scale * mval : 0.f;
#pragma unroll
    for (int i = 0; i < SK_RPW; ++i) {
      const int rr = wid + NWV * i;
      float x = xv[i];
      const float xn = x * slope;
      x = (lrelu && x < 0.f) ? xn : x;
      // unconditional: the tile is allocated for SK_RPW rows per wave, rows >= nrows are never read.  Behind `if (rr < nrows)` every
      // store was a basic block of its own and the compiler SANK the row's global load into it (load -> s_waitcnt vmcnt(0) -> store)
      Xs[rr * SK_XP + lane] = (rr < nrows && 8 * G0 + rr < cin) ? x * cs : 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < SK_PD; ++i) { load_unit(i); __builtin_amdgcn_sched_barrier(0); }
  if (LDSX) __syncthreads();
  if (L.dbg) ts1 = __builtin_amdgcn_s_memtime();
  if (LDSX) {
    // units run (group, tap) in the order the ring was loaded; the next unit's B operands are read while this unit's MFMAs run
    int ug = g0, uj = 0;
    float bq[2][4];
    {
      const float* xb = Xs + (8 * (ug - G0) + lh) * SK_XP + l31;
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[0][q] = xb[q * 2 * SK_XP];
    }
    for (int u0 = 0; u0 < U; u0 += SK_PD) {
#pragma unroll
      for (int i = 0; i < SK_PD; ++i) {
        if (u0 + i < U) {
          int jn = uj + 1, gn = ug;
          if (jn == k) { jn = 0; ++gn; }
          const bool more = u0 + i + 1 < U;
          const float* xb = Xs + (8 * ((more ? gn : ug) - G0) + lh) * SK_XP + l31 + (more ? jn : uj) * dil;
#pragma unroll
          for (int q = 0; q < 4; ++q) bq[(i & 1) ^ 1][q] = xb[q * 2 * SK_XP];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[i].x, bq[i & 1][0], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[i].y, bq[i & 1][1], acc2, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[i].z, bq[i & 1][2], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[i].w, bq[i & 1][3], acc2, 0, 0, 0);
          uj = jn; ug = gn;
        }
        load_unit(i);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();                              // every wave is done with the X tile before `red` overwrites it
  } else
  for (int u0 = 0; u0 < U; u0 += SK_PD) {
#pragma unroll
    for (int i = 0; i < SK_PD; ++i) {
      if (u0 + i < U) {
        float bq[4];
        const float cs = MASK ? vmr[i] * mr[i] : vmr[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float x = br[i][q];
          const float xn = x * slope;
          x = (lrelu && x < 0.f) ? xn : x;
          bq[q] = x * cs;
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[i].x, bq[0], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[i].y, bq[1], acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[i].z, bq[2], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[i].w, bq[3], acc2, 0, 0, 0);
      }
      load_unit(i);
      __builtin_amdgcn_sched_barrier(0);          // keep program order: the ring's vmcnt distances stay SK_PD - 1 units
    }
  }

  if (L.dbg) ts2 = __builtin_amdgcn_s_memtime();
  // reduce the waves' partial tiles through LDS
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wid][(r & 3) + 8 * (r >> 2) + 4 * lh][l31] = acc[r] + acc2[r];
  __syncthreads();
  // epilogue (its operands — ConvProb fields, bias, residual — were loaded before the main loop: "epilogue operands first" above)
  {
    float vmx = 0.f;
    if (act == ACT_GATE) {                          // rows rl and rl + 16 are a (tanh, sigmoid) pair: passes i and i + 16/RPP of this thread
      if constexpr (RPP <= 16) {
        float vs[32 / RPP];
#pragma unroll
        for (int i = 0; i < 32 / RPP; ++i) {
          const int rl = (tid >> 5) + RPP * i;
          float vv = 0.f;
#pragma unroll
          for (int w = 0; w < NWV; w += 4)
            vv += (red[w][rl][tid & 31] + red[w + 1][rl][tid & 31]) + (red[w + 2][rl][tid & 31] + red[w + 3][rl][tid & 31]);
          vs[i] = vv + (bsv[i] + b2v[i]);
        }
#pragma unroll
        for (int i = 0; i < 16 / RPP; ++i) {
          const int rl = (tid >> 5) + RPP * i;
          const float v = tanhf(vs[i]) * (1.f / (1.f + expf(-vs[i + 16 / RPP])));
          if (colok && m0 + rl + 16 < cout) outb[(unsigned)(m0 / 2 + rl) * o_rs + coff] = v;
        }
      }
    } else
#pragma unroll
    for (int i = 0; i < 32 / RPP; ++i) {
      const int rl = (tid >> 5) + RPP * i;
      const int row = m0 + rl;
      float vv = 0.f;
#pragma unroll
      for (int w = 0; w < NWV; w += 4)
        vv += (red[w][rl][tid & 31] + red[w + 1][rl][tid & 31]) + (red[w + 2][rl][tid & 31] + red[w + 3][rl][tid & 31]);
      vv += bsv[i] + b2v[i];
      if (act == ACT_RELU) vv = vv < 0.f ? 0.f : vv;             // host guarantees act == NONE when ksplit > 1
      else if (act == ACT_GELU) vv = 0.5f * vv * (1.0f + erff(vv * 0.70710678118654752440f));
      if (mask_pre) vv *= om;
      if (z == 0) {
        if (res_mode == RES_ADD) vv += rvv[i];
        else if (res_mode == RES_RSUB) vv = rvv[i] - vv;
      } else if (res_mode == RES_RSUB) {
        vv = -vv;
      }
      if (mask_post) vv *= om;
      if (colok && row < cout) {
        outb[(unsigned)row * o_rs + coff] = vv;
        vmx = fmaxf(vmx, fabsf(vv));
      }
    }
    if (omaxp) x3_publish(omaxp, 0u, vmx, lane);    // ksplit == 1 only (host): max |v| of the stored tensor, see conv1d_mfma_kernel
  }
  if (L.dbg && tid == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long* d = L.dbg + 8ull * blockIdx.x;
    d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = __builtin_amdgcn_s_memtime();
    d[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    d[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    d[6] = (unsigned long long)U; d[7] = 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
struct TileCfg { int id, bm, bn, xs; const char* name; };
static const TileCfg kTiles[] = {
    // order = preference of the auto picker (largest first)
    {TILE_128x128, 128, 128, 3, "conv1d_mfma<128x128>"}, {TILE_64x128, 64, 128, 3, "conv1d_mfma<64x128>"},
    {TILE_32x256, 32, 256, 5, "conv1d_mfma<32x256>"},    {TILE_64x64, 64, 64, 2, "conv1d_mfma<64x64>"},
    {TILE_32x128, 32, 128, 3, "conv1d_mfma<32x128>"},
};
// 128x64 (each wave 64 rows x 32 columns: every B fragment feeds two MFMAs): only reachable through the tuning hook, see below
static const TileCfg kTile128x64 = {TILE_128x64, 128, 64, 2, "conv1d_mfma<128x64>"};

// tuning experiments (tools/kbench.py drive these through bv2_test_set_tuning; 0 = the shipped heuristics)
static int g_tune_splitk_waves = 0, g_tune_force_ck = 0, g_tune_no_ldsx = 0, g_tune_no_snake = 0;
static long g_tune_tile_target = 0;
void conv_set_tuning(int splitk_waves, int force_ck, long tile_target) {
  g_tune_no_ldsx = splitk_waves >= 100 ? 1 : 0;                     // +100: the non-staged split-K form (A/B comparisons)
  g_tune_splitk_waves = splitk_waves % 100; g_tune_force_ck = force_ck % 100; g_tune_tile_target = tile_target;
  g_tune_no_snake = force_ck >= 100 ? 1 : 0;                        // force_ck + 100: plain z-major placement (A/B comparisons)
}

static unsigned long long* g_tl_buf = nullptr;
static long long g_tl_cap = 0, g_tl_off = 0;
struct TlMeta { long long off; int gx, gy, gz, tile, ks, cin, L; };
static TlMeta g_tl_meta[512];
static int g_tl_n = 0;
void conv_set_timeline(unsigned long long* dev_buf, long long capacity_u64) { g_tl_buf = dev_buf; g_tl_cap = capacity_u64; g_tl_off = 0; g_tl_n = 0; }
int conv_timeline_report(long long* meta, int max_launches) {
  int n = g_tl_n < max_launches ? g_tl_n : max_launches;
  for (int i = 0; i < n; ++i) {
    const TlMeta& m = g_tl_meta[i];
    long long* o = meta + 8 * i;
    o[0] = m.off; o[1] = m.gx; o[2] = m.gy; o[3] = m.gz; o[4] = m.tile; o[5] = m.ks; o[6] = m.cin; o[7] = m.L;
  }
  return n;
}
unsigned long long* timeline_slice(unsigned gx, unsigned gy, unsigned gz, int tile, int ks, int cin, int L) {
  if (!g_tl_buf) return nullptr;
  const long long need = 8ll * gx * gy * gz;
  if (g_tl_off + need > g_tl_cap || g_tl_n >= 512) return nullptr;
  unsigned long long* r = g_tl_buf + g_tl_off;
  g_tl_meta[g_tl_n++] = TlMeta{g_tl_off, (int)gx, (int)gy, (int)gz, tile, ks, cin, L};
  g_tl_off += need;
  return r;
}
static ConvLaunch with_timeline(const ConvLaunch& L, dim3 grid, int tile) {
  ConvLaunch r = L;
  if (!g_tl_buf) return r;
  int ks = 0;
  for (int i = 0; i < L.nprob && i < 3; ++i) ks |= (L.p[i].k & 255) << (8 * i);
  r.dbg = timeline_slice(grid.x, grid.y, grid.z, tile, ks, L.p[0].cin, L.L);
  return r;
}

template <int WM, int WN, int MI, int NI, int XS>
static int launch_variant(hipStream_t stream, const ConvLaunch& L0, int ck, int max_cout_pad, int max_extra, int max_chunks) {
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
  if (BN + max_extra > XS * 64) return -2;        // halo does not fit the staged tile
  const int mtiles = (max_cout_pad + BM - 1) / BM;
  const int ntx = (L0.L + BN - 1) / BN;
  const int per_xcd = (ntx % 8 == 0 || ntx >= 64) ? (ntx + 7) / 8 : 0;      // contiguous per-XCD ranges only if they balance
  dim3 grid(per_xcd ? per_xcd * 8 : ntx, mtiles * L0.B, L0.nprob);
  // cost-balanced ("snake") placement: problems of different cost, and the whole launch resident at once (<= 4 workgroups per
  // CU) so that dispatch is a plain round-robin over the CUs — see the kernel
  ConvLaunch Ls = L0;
  int snake_n = 0;
  if (per_xcd && L0.nprob > 1 && !g_tune_no_snake) {
    const long total = (long)per_xcd * 8 * mtiles * L0.B * L0.nprob;
    bool differ = false;
    for (int i = 1; i < L0.nprob; ++i)
      if (L0.p[i].k * L0.p[i].cin_pad != L0.p[0].k * L0.p[0].cin_pad) differ = true;
    if (differ && total > 256 && total <= 1024) {
      for (int i = 0; i < Ls.nprob; ++i)                 // most expensive problem first (insertion sort, <= 3 entries)
        for (int j = i; j > 0 && Ls.p[j].k * Ls.p[j].cin_pad > Ls.p[j - 1].k * Ls.p[j - 1].cin_pad; --j) {
          const ConvProb t = Ls.p[j]; Ls.p[j] = Ls.p[j - 1]; Ls.p[j - 1] = t;
        }
      snake_n = per_xcd * mtiles * L0.B * L0.nprob;
      grid = dim3(snake_n * 8, 1, 1);
    }
  }
  const ConvLaunch L = with_timeline(Ls, grid, BM * 1000 + BN);
  const int nxbuf = max_chunks > 1 ? 2 : 1;     // a single-chunk problem never re-stages its X tile
  // weights go global -> registers; LDS holds only the (double-buffered) X chunk
  if (ck == 32) {
    const size_t lds = sizeof(float) * (size_t)(nxbuf * 32 * XS * 64);
    auto kern = conv1d_mfma_kernel<WM, WN, MI, NI, 32, XS>;
    ensure_dyn_lds((const void*)kern, lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, L, mtiles, per_xcd, snake_n);
  } else {
    const size_t lds = sizeof(float) * (size_t)(nxbuf * 16 * XS * 64);
    auto kern = conv1d_mfma_kernel<WM, WN, MI, NI, 16, XS>;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, L, mtiles, per_xcd, snake_n);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Work split of the split-K kernel.  Measured on MI355X (tools/kbench.py, profiles/r01_kbench_*.txt): a wave is fastest
// with ~16 (group, tap) units of 4 MFMAs; more waves per workgroup than 8 never helps, and when the consumer can sum
// partial slabs, splitting K ACROSS workgroups (ksplit) beats adding waves.
constexpr int SK_UNITS_PER_WAVE = 16;

static int splitk_units(const ConvLaunch& L) {
  int units = 0;
  for (int i = 0; i < L.nprob; ++i) {
    const int u = (L.p[i].cin_pad / 8) * L.p[i].k;
    if (u > units) units = u;
  }
  return units;
}

static int splitk_waves(const ConvLaunch& L) {
  if (g_tune_splitk_waves == 4 || g_tune_splitk_waves == 8 || g_tune_splitk_waves == 16) return g_tune_splitk_waves;
  const int slices = (splitk_units(L) + SK_UNITS_PER_WAVE - 1) / SK_UNITS_PER_WAVE;   // K slices wanted in total
  const int per_wg = (slices + L.ksplit - 1) / L.ksplit;
  return per_wg > 4 ? 8 : 4;
}

// rows of the X tile a workgroup stages in the LDSX form: the largest channel-group range of any (problem, slice)
static int splitk_stage_rows(const ConvLaunch& L, int nw) {
  int rows = 0;
  for (int i = 0; i < L.nprob; ++i) {
    const int groups = L.p[i].cin_pad / 8, nsl = L.ksplit * nw;
    for (int z = 0; z < L.ksplit; ++z) {
      const int a = (int)(((int64_t)groups * (z * nw)) / nsl), b = (int)(((int64_t)groups * ((z + 1) * nw)) / nsl);
      rows = 8 * (b - a) > rows ? 8 * (b - a) : rows;
    }
  }
  return rows;
}

template <bool MASK, bool LDSX>
static void launch_splitk_nw(hipStream_t stream, const ConvLaunch& L, int nw, dim3 grid, int mtiles, int ntiles, int per_xcd,
                             int total, int stage_rows) {
  size_t lds = sizeof(float) * (size_t)nw * 32 * 33;
  if (LDSX) lds = sizeof(float) * (size_t)nw * SK_RPW * SK_XP;     // every wave stores its SK_RPW rows unconditionally (>= the reduce buffer)
  (void)stage_rows;
  if (nw == 16) {
    auto kern = conv1d_splitk_kernel<MASK, 16, LDSX>;
    ensure_dyn_lds((const void*)kern, lds);
    hipLaunchKernelGGL(kern, grid, dim3(1024), lds, stream, L, mtiles, ntiles, per_xcd, total);
  } else if (nw == 8) {
    auto kern = conv1d_splitk_kernel<MASK, 8, LDSX>;
    ensure_dyn_lds((const void*)kern, lds);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, L, mtiles, ntiles, per_xcd, total);
  } else {
    auto kern = conv1d_splitk_kernel<MASK, 4, LDSX>;
    ensure_dyn_lds((const void*)kern, lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, L, mtiles, ntiles, per_xcd, total);
  }
}

static int launch_splitk(hipStream_t stream, const ConvLaunch& L0, int max_cout_pad, const char** variant_name) {
  const int mtiles = max_cout_pad / 32, ntiles = (L0.L + 31) / 32;
  const ConvLaunch L = with_timeline(L0, dim3(((L0.nprob * L0.B * mtiles * L0.ksplit * ntiles + 7) / 8) * 8), 32032);
  const int total = L.nprob * L.B * mtiles * L.ksplit * ntiles;
  const int per_xcd = (total + 7) / 8;
  bool any_mask = false, all_mask = true;
  for (int i = 0; i < L.nprob; ++i) {
    if (L.p[i].nsrc != 1) return -2;              // multi-source inputs only exist on the LDS-tiled path
    if (L.p[i].in_mask) any_mask = true; else all_mask = false;
  }
  if (any_mask != all_mask) return -2;            // one launch = one mask mode
  int nw = splitk_waves(L);
  for (int i = 0; i < L.nprob; ++i)
    if (L.p[i].act == ACT_GATE && nw == 16) nw = 8;   // the gate pairs rows inside one thread: at most 16 rows per pass
  if (variant_name) *variant_name = nw == 16 ? "conv1d_splitk<32x32,16w>" : (nw == 8 ? "conv1d_splitk<32x32,8w>" : "conv1d_splitk<32x32,4w>");
  const dim3 grid(per_xcd * 8 + (L.pf.ptr && L.pf.bytes ? PF_BLOCKS : 0));
  // LDSX form: every problem has taps to re-use (k > 1), the staged tile fits (32 + (k-1)*dil <= SK_XP columns, <= SK_RPW rows
  // per wave) and a lane's 32-bit byte offsets reach every element
  bool ldsx = !g_tune_no_ldsx;
  const int stage_rows = splitk_stage_rows(L, nw);
  for (int i = 0; i < L.nprob; ++i)
    if (L.p[i].k < 2 || 32 + (L.p[i].k - 1) * L.p[i].dil > SK_XP) ldsx = false;
  if (stage_rows > SK_RPW * nw) ldsx = false;
  if (variant_name && ldsx) *variant_name = nw == 16 ? "conv1d_splitk_ldsx<32x32,16w>" : (nw == 8 ? "conv1d_splitk_ldsx<32x32,8w>" : "conv1d_splitk_ldsx<32x32,4w>");
  if (ldsx) {
    if (any_mask) launch_splitk_nw<true, true>(stream, L, nw, grid, mtiles, ntiles, per_xcd, total, stage_rows);
    else launch_splitk_nw<false, true>(stream, L, nw, grid, mtiles, ntiles, per_xcd, total, stage_rows);
  } else {
    if (any_mask) launch_splitk_nw<true, false>(stream, L, nw, grid, mtiles, ntiles, per_xcd, total, stage_rows);
    else launch_splitk_nw<false, false>(stream, L, nw, grid, mtiles, ntiles, per_xcd, total, stage_rows);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// K-split factor ACROSS workgroups for the split-K kernel (only when the consumer sums the slabs).
int conv_pick_ksplit(const ConvLaunch& L, int max_split) {
  if (max_split <= 1) return 1;
  int groups = 1 << 30;
  for (int i = 0; i < L.nprob; ++i) {
    if (L.p[i].cin_pad / 8 < groups) groups = L.p[i].cin_pad / 8;
    if (L.p[i].act != ACT_NONE) return 1;
  }
  const int slices = (splitk_units(L) + SK_UNITS_PER_WAVE - 1) / SK_UNITS_PER_WAVE;
  int ks = 1;
  while (ks < max_split && ks * 4 < slices && groups / (ks * 2 * 4) >= 1) ks *= 2;
  return ks;
}

bool conv_use_splitk(const ConvLaunch& L) {
  // small-N regime: even 32x128 tiles would leave most CUs idle
  int mt = 0;
  for (int i = 0; i < L.nprob; ++i)
    if (L.p[i].cout_pad / 32 > mt) mt = L.p[i].cout_pad / 32;
  for (int i = 0; i < L.nprob; ++i)
    if (L.p[i].nsrc != 1 || (L.p[i].in_mask != nullptr) != (L.p[0].in_mask != nullptr)) return false;
  const long cols = (long)L.B * L.L;
  const long tiles128 = (long)L.nprob * mt * ((L.L + 127) / 128) * L.B;
  // Round 5: a long-K conv on a medium batch (the text encoder's FFN conv_2 at B = 32: 768 x 3 taps -> 192 rows x 4 096 columns, 3.6 GFLOP)
  // fell on this side of the line with 192 "tiles" and ran as 6 144 four-wave split-K workgroups writing eight partial slabs: 76 us against
  // 41 us for its twin conv_1 on the LDS-tiled kernel.  Enough work AND enough 128-column tiles AND whole 64-column tiles per item: LDS-tiled.
  double flops = 0;
  for (int i = 0; i < L.nprob; ++i) flops += 2.0 * L.p[i].cout * L.p[i].cin * L.p[i].k * (double)cols;
  if (flops >= 2e9 && tiles128 >= 128 && L.L >= 64) return false;
  return cols <= 4096 && tiles128 < 512;
}

int launch_conv1d(hipStream_t stream, const ConvLaunch& L, int tile, const char** variant_name) {
  if (L.nprob < 1 || L.nprob > BV2_MAX_PROBS || L.B < 1 || L.L < 1) return -1;
  int max_cout_pad = 0, max_extra = 0, ck = 32, max_chunks = 1;
  for (int i = 0; i < L.nprob; ++i) {
    const ConvProb& p = L.p[i];
    if (p.cout_pad % 32 || p.cin_pad % 16 || p.k < 1 || p.dil < 1 || p.w_ld % 128 || p.w_ld < p.cout_pad) return -1;
    if (p.cout_pad > max_cout_pad) max_cout_pad = p.cout_pad;
    if ((p.k - 1) * p.dil > max_extra) max_extra = (p.k - 1) * p.dil;
    if (p.cin_pad % 32) ck = 16;
  }
  if (g_tune_force_ck == 16) ck = 16;
  for (int i = 0; i < L.nprob; ++i)
    if (L.p[i].cin_pad / ck > max_chunks) max_chunks = L.p[i].cin_pad / ck;
  const bool auto_sk = tile == TILE_AUTO && conv_use_splitk(L);
  if (auto_sk) tile = TILE_SPLITK;
  if (tile == TILE_SPLITK) {
    if (L.ksplit < 1 || L.ksplit > BV2_MAX_KSPLIT) return -1;
    return launch_splitk(stream, L, max_cout_pad, variant_name);
  }
  if (L.ksplit != 1) return -1;                   // the LDS-tiled kernel never splits K across workgroups
  // every problem carries its split-bf16 weight planes: the same conv on the bf16 matrix core (conv_x6.hip)
  if (tile >= TILE_X6) return launch_conv1d_x6(stream, L, tile, variant_name);
  if (tile == TILE_AUTO && g_tune_tile_target == 0 && conv_x6_supported(L)) return launch_conv1d_x6(stream, L, TILE_X6, variant_name);
  if (tile == TILE_AUTO) {
    // largest tile that still yields >= ~6 workgroups per CU (256 CUs; measured optimum, profiles/r01_c_*): the problems
    // of one launch differ in cost (k = 3 / 7 / 11 branches) and only 2-3 workgroups are resident per CU, so the
    // dispatcher balances them only if there are several times more workgroups than slots; per-tile prologue/epilogue
    // latency is also hidden by the co-resident workgroups.  conv_set_tuning overrides (tuning experiments).
    const long target = g_tune_tile_target > 0 ? g_tune_tile_target : 1536L;
    // tuning experiments (tools/tune_tiles.py): negative targets force one tile for one stage shape
    if (((g_tune_tile_target == -7 && max_cout_pad % 128 == 0) || (g_tune_tile_target == -8 && max_cout_pad == 128)) &&
        64 + max_extra <= 128) {
      if (variant_name) *variant_name = kTile128x64.name;
      return launch_variant<2, 2, 2, 1, 2>(stream, L, ck, max_cout_pad, max_extra, max_chunks);
    }
    if ((g_tune_tile_target == -9 && max_cout_pad == 64) || (g_tune_tile_target == -10 && max_cout_pad == 128)) {
      if (variant_name) *variant_name = "conv1d_mfma<64x128>";
      return launch_variant<2, 2, 1, 2, 3>(stream, L, ck, max_cout_pad, max_extra, max_chunks);
    }
    if (g_tune_tile_target == -11 && max_cout_pad == 128) {
      if (variant_name) *variant_name = "conv1d_mfma<128x128>";
      return launch_variant<2, 2, 2, 2, 3>(stream, L, ck, max_cout_pad, max_extra, max_chunks);
    }
    tile = TILE_32x128;
    double best_score = -1.0;
    bool reached = false;
    for (const TileCfg& t : kTiles) {
      if (t.bm > max_cout_pad && t.bm != 32) continue;
      if (max_cout_pad % t.bm && t.bm != 32) continue;
      if (t.bn + max_extra > t.xs * 64) continue;
      const long nb = (L.L + t.bn - 1) / t.bn;
      const long blocks = nb * ((max_cout_pad + t.bm - 1) / t.bm) * L.B * L.nprob;
      // tiles never span batch items: the fraction of a tile row that holds real columns (L = 128 under a 256-wide tile
      // leaves half of every workgroup idle)
      const double useful = (double)L.L / (double)(nb * t.bn);
      if (blocks >= target && useful >= 0.75) { tile = t.id; reached = true; break; }
      const double score = (double)(blocks < target ? blocks : target) * useful;
      if (score > best_score) { best_score = score; tile = t.id; }   // nothing reaches the target: most useful workgroups, largest first
    }
    (void)reached;
  }
  for (const TileCfg& t : kTiles)
    if (t.id == tile && variant_name) *variant_name = t.name;
  switch (tile) {
    case TILE_128x128: return launch_variant<2, 2, 2, 2, 3>(stream, L, ck, max_cout_pad, max_extra, max_chunks);
    case TILE_64x128:  return launch_variant<2, 2, 1, 2, 3>(stream, L, ck, max_cout_pad, max_extra, max_chunks);
    case TILE_64x64:   return launch_variant<2, 2, 1, 1, 2>(stream, L, ck, max_cout_pad, max_extra, max_chunks);
    case TILE_32x128:                             // halo > 64 columns (ResBlock2's k = 7, dilation 12: 72): the same tile on a 256-column staged chunk
      if (128 + max_extra > 3 * 64) return launch_variant<1, 4, 1, 1, 4>(stream, L, ck, max_cout_pad, max_extra, max_chunks);
      return launch_variant<1, 4, 1, 1, 3>(stream, L, ck, max_cout_pad, max_extra, max_chunks);
    case TILE_32x256:  return launch_variant<1, 4, 1, 2, 5>(stream, L, ck, max_cout_pad, max_extra, max_chunks);
    case TILE_128x64:  return launch_variant<2, 2, 2, 1, 2>(stream, L, ck, max_cout_pad, max_extra, max_chunks);
  }
  return -1;
}

double conv_flops(const ConvLaunch& L) {
  double f = 0;
  for (int i = 0; i < L.nprob; ++i) f += 2.0 * L.p[i].cout * L.p[i].cin * L.p[i].k * (double)L.L * L.B;
  return f;
}

double conv_bytes(const ConvLaunch& L) {   // each input read once, each output written once, weights once
  double by = 0;
  for (int i = 0; i < L.nprob; ++i) {
    const ConvProb& p = L.p[i];
    by += 4.0 * ((double)p.cin * p.nsrc * L.L * L.B + (double)p.cout * L.L * L.B * (p.res_mode ? 2 : 1) +
                 (double)p.cout * p.cin * p.k);
  }
  return by;
}

}  // namespace bv2

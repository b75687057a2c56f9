// gen_bf16.hip — the HiFi-GAN Generator (reference models.py:538-557, modules.py:296-309) in bf16 on the gfx950 matrix core
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate), activations CHANNELS-LAST [B][L][C] so that
//   * a tile of consecutive time steps is one contiguous HBM range: staging loads are 16 B per lane, fully coalesced;
//   * the MFMA B operand (8 consecutive input channels of one time step) is ONE ds_read_b128 from the LDS tile, and the
//     k taps / dilations of a conv are row shifts of the same tile (read once from HBM, used k times);
//   * a ConvTranspose1d is an ordinary conv with C_out' = stride*C_out (its u output rows per input step are contiguous);
//   * the D fragment (lane = time step, 4 consecutive output channels per register group) stores as 8-byte pieces of a row.
// BASELINE config 3 ("bf16 weights/activations, fp32 accumulate") for the 90 % of the path's FLOPs that live in the
// Generator; in bf16 its layer-wise arithmetic intensity (117-940 FLOP/B by stage) straddles the machine balance
// (2.5 PF / 6.3 TB/s = 400 FLOP/B): wide stages are MFMA-bound, narrow stages HBM-bound.
//
// conv_cl_bf16_kernel<WN, WM>: workgroup = WN x WM waves; wave (wn, wm) owns output channels [32*(cg*WN+wn), +32) and
// time steps [t0 + 128*wm, +128) (4 accumulator tiles of 32x32).  Prologue: the whole input tile — (WM*128 + (k-1)*dil)
// rows x C_in channels — goes HBM -> registers -> (mean of branches, leaky-ReLU, bf16 round) -> LDS, row pitch C_in + 8
// elements (= odd multiple of 16 B: conflict-free ds_read_b128 for 16 consecutive rows).  Main loop, NO barriers: the
// wave streams its weight fragments global -> registers through an 8-deep ring (1 KB units, L2-resident, contiguous per
// 32-channel output tile) and reads B fragments from LDS one unit ahead of the MFMAs.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include "../bv2_kernels.h"
#include "cl_bf16.h"

namespace bv2 {

// workgroup = WN x WM waves; wave (wn, wm) owns MI 32-channel output tiles starting at 32*MI*(cg*WN + wn) and NI 32-step
// time tiles starting at t0 + 32*NI*wm
// 4-wave workgroups (one wave per SIMD) with one 32x128 accumulator block per wave want THREE workgroups per CU (the LDS tile
// allows it): the second launch-bounds argument (waves per SIMD) keeps the register allocation at <= 168 — without it the
// staging batch below pushed the kernel to 178 registers and one workgroup per CU was lost
template <int WN, int WM, int MI, int NI, int G>   // G > 0: C_in = 16*G for every problem of the launch (tap-major GEMM)
__global__ void __launch_bounds__(64 * WN * WM, (WN * WM == 4 && MI * NI <= 2) ? 5 : ((WN * WM == 4 && MI * NI <= 4) ? 3 : 1))
conv_cl_bf16_kernel(const ClLaunch L, const int ngrp) {
  constexpr int NT = 64 * WN * WM;
  constexpr int WT = 32 * NI;
  constexpr int BT = WM * WT;
  constexpr int PD = MI == 1 ? CL_PD : CL_PD / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned short xs[];
  const ClProb& P = L.p[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wid % WN, wm = wid / WN;
  const int l31 = lane & 31, lh = lane >> 5;
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;                    // timeline stamps (tools/timeline.py; L.dbg is null in the product)
  if (L.dbg) ts0 = __builtin_amdgcn_s_memtime();
  const int b = blockIdx.y / ngrp;
  const int cg = blockIdx.y - b * ngrp;
  const int mt0 = (cg * WN + wn) * MI;
  const int t0 = blockIdx.x * BT;
  const int cin = P.cin, k = P.k, dil = P.dil;
  if (cg * WN * MI * 32 >= P.cout_pad) return;    // whole workgroup beyond this problem's channels (uniform: before the barrier)
  int Lin = P.Lin;
  if (L.lens) {                                   // exact lengths: this batch item's input ends at lens[b]*len_mul
    const int64_t lv = L.lens[b] * L.len_mul;
    Lin = lv < Lin ? (int)lv : Lin;
    if (t0 >= Lin) return;                        // a tile wholly past the utterance: nobody reads its outputs
  }
  const int pitch = cin + 8;
  cl_stage<NT>(xs, pitch, P.x[0] + (int64_t)b * P.x_bstride, P.x[1] ? P.x[1] + (int64_t)b * P.x_bstride : nullptr,
               P.x[2] ? P.x[2] + (int64_t)b * P.x_bstride : nullptr, P.nsrc, P.in_scale, P.pre_lrelu != 0, P.slope,
               t0 - P.pad_left, BT + (k - 1) * dil, cin, Lin, tid);
  __syncthreads();
  if (L.dbg) ts1 = __builtin_amdgcn_s_memtime();
  const bool active = mt0 * 32 < P.cout_pad;      // waves beyond this problem's channels still join the epilogue barriers

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  const int U = (cin >> 4) * k;
  // a wave whose second m-tile lies beyond cout_pad streams the first one twice (results of the copy are dropped)
  const int64_t mstride = (MI > 1 && (mt0 + 1) * 32 < P.cout_pad) ? (int64_t)U * 512 : 0;
  if (active) {
    // ConvTranspose1d as one conv: only the window taps of the phases this wave's output channels belong to (wave-uniform; the
    // packed stream keeps the other taps as zeros, they are stepped over — 1/3 of the units at k = 2u, 1/5 at k = 4u)
    int ja = 0, jb = k;
    if (P.ph_cout > 0) {
      const int c_lo = mt0 * 32, c_end = (mt0 + MI) * 32 < P.cout ? (mt0 + MI) * 32 : P.cout;
      const int ph0 = c_lo / P.ph_cout, ph1 = (c_end - 1) / P.ph_cout;
      ja = k; jb = 0;
      for (int ph = ph0; ph <= ph1; ++ph) {
        const int off = (int)((P.ph_offs >> (4 * ph)) & 15ull);
        ja = off < ja ? off : ja;
        jb = off + P.ph_ntaps > jb ? off + P.ph_ntaps : jb;
      }
      ja = __builtin_amdgcn_readfirstlane(ja); jb = __builtin_amdgcn_readfirstlane(jb);
    }
    const int kk = jb - ja;
    if constexpr (G > 0)
      cl_gemm_tm<MI, NI, G>(acc, P.w + ((int64_t)mt0 * U + ja) * 512, mstride, 16u * (unsigned)lane, kk,
                            xs + (wm * WT + l31 + ja * dil) * pitch + lh * 8, dil, k);
    else
      cl_gemm<MI, NI, PD>(acc, P.w + ((int64_t)mt0 * U + ja) * 512 + lane * 8, mstride, (cin >> 4) * kk, kk,
                          xs + (wm * WT + l31 + ja * dil) * pitch + lh * 8, pitch, dil, k);
  }

  if (L.dbg) ts2 = __builtin_amdgcn_s_memtime();
  // ---- epilogue through LDS.  In the D fragment a lane holds 4 consecutive channels of ONE time step, i.e. 8-byte pieces
  // 2*C bytes apart in HBM: stored (and, for the residual, loaded) directly, every wave instruction touches 32 different
  // rows.  Instead the workgroup's output tile [BT][WGC channels] is assembled in LDS (the input tile is dead by now) and
  // moved to / from HBM in 16-byte pieces along the rows: consecutive lanes hit consecutive addresses.
  constexpr int WGC = WN * MI * 32;               // channels per workgroup
  constexpr int OP = WGC + 8;                     // LDS pitch of the output tile (odd multiple of 16 B)
  const int cout = P.cout;
  const int ch0 = cg * WGC;
  const int wch = cout - ch0 < WGC ? cout - ch0 : WGC;             // valid channels of this workgroup (multiple of 8)
  const int rows = L.L - t0 < BT ? L.L - t0 : BT;
  const int ppr = wch >> 3;
  // The residual tile's loads are ALL issued here, before the barrier (they fly while the slower waves finish their GEMM), and
  // land in LDS after it; a piece loop that loads and stores one 16-byte piece per iteration serialises EPI_PIECES global round
  // trips (measured: +12k cycles per workgroup on every convs2 launch).
  constexpr int EPI_PIECES = (BT * (WGC / 8) + NT - 1) / NT;      // 16-byte pieces per thread of a full [BT][WGC] tile
  const uint16_t* const resp = P.res;
  u32x4 rv[EPI_PIECES];
  if (resp) {
    const uint16_t* rg = resp + (int64_t)b * P.res_bstride + (int64_t)t0 * cout + ch0;
    const int npc = rows * ppr;
#pragma unroll
    for (int i = 0; i < EPI_PIECES; ++i) {
      int p = tid + i * NT;
      p = p < npc ? p : npc - 1;
      const int r = p / ppr, c = p - r * ppr;
      rv[i] = *reinterpret_cast<const u32x4*>(rg + (int64_t)r * cout + c * 8);
    }
  }
  // The bias of this lane's channels: 4*MI float4 loads, issued HERE in one batch behind the residual's (both fly while the slower
  // waves finish their GEMM; nothing reads them before the barriers below).  Inside the accumulator loop — `if (P.bias) bv = *(P.bias
  // + co)` between LDS stores — every (time tile, channel group) iteration re-loaded the pointer from the kernarg segment (the LDS
  // store may alias it, for all the compiler knows) and waited for its own load: 16 serial scalar + vector round trips per wave in
  // EVERY workgroup's epilogue (ISA of round 2; tools/timeline.py had the epilogue at 17-25k cycles next to a 20k-cycle GEMM).
  // The per-batch bias (conv_pre's launch only) stays a load in the loop.
  // (Only for one accumulator row block per wave: with MI = 2 — the C = 64 stage's <1x8,2x2> tile — the 32 extra live registers
  // changed the allocation of the whole kernel, 167 -> 136, and that variant ran 20 % SLOWER in a same-box A/B of the two builds,
  // tools/ab_build.py; it keeps the loads in the loop.)
  constexpr bool HB = MI == 1;
  f32x4 bvv[MI][4];
  if constexpr (HB) {
    const float* const b1 = P.bias ? P.bias : reinterpret_cast<const float*>(P.w);     // no bias: valid dummy, masked below
    const bool has_b1 = P.bias != nullptr;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int co = cg * WGC + (wn * MI + mi) * 32 + 8 * g + 4 * lh;
        co = (co + 4 <= cout && has_b1) ? co : 0;
        bvv[mi][g] = *reinterpret_cast<const f32x4*>(b1 + co);
      }
  }
  const float bsel = P.bias != nullptr ? 1.f : 0.f;   // applied at the use: a select here would have to wait for the loads
  const float* const bias2p = P.bias2 ? P.bias2 + (int64_t)b * P.bias2_bstride : nullptr;
  __syncthreads();                                // every wave is done reading the input tile
  if (resp) {
    const int npc = rows * ppr;
#pragma unroll
    for (int i = 0; i < EPI_PIECES; ++i) {
      const int p = tid + i * NT;
      if (p < npc) {
        const int r = p / ppr, c = p - r * ppr;
        *reinterpret_cast<u32x4*>(xs + r * OP + c * 8) = rv[i];
      }
    }
    __syncthreads();
  }
  if (active) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      if (mi > 0 && (mt0 + mi) * 32 >= P.cout_pad) continue;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int tr = wm * WT + ni * 32 + l31;   // row inside the tile
        if (tr >= rows) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int cl = (wn * MI + mi) * 32 + 8 * g + 4 * lh;     // channel inside the workgroup's range
          const int co = ch0 + cl;
          if (co >= cout) continue;
          float v0 = acc[mi][ni][4 * g], v1 = acc[mi][ni][4 * g + 1], v2 = acc[mi][ni][4 * g + 2], v3 = acc[mi][ni][4 * g + 3];
          if constexpr (HB) {
            if (bsel != 0.f) { v0 += bvv[mi][g].x; v1 += bvv[mi][g].y; v2 += bvv[mi][g].z; v3 += bvv[mi][g].w; }
          } else if (P.bias) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(P.bias + co);
            v0 += bv.x; v1 += bv.y; v2 += bv.z; v3 += bv.w;
          }
          if (HB ? bias2p != nullptr : P.bias2 != nullptr) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>((HB ? bias2p : P.bias2 + (int64_t)b * P.bias2_bstride) + co);
            v0 += bv.x; v1 += bv.y; v2 += bv.z; v3 += bv.w;
          }
          unsigned short* slot = xs + tr * OP + cl;
          if (resp) {
            const u32x2 rr = *reinterpret_cast<const u32x2*>(slot);
            v0 += bf_lo(rr.x); v1 += bf_hi(rr.x); v2 += bf_lo(rr.y); v3 += bf_hi(rr.y);
          }
          u32x2 o;
          o.x = bf_pack(v0, v1); o.y = bf_pack(v2, v3);
          *reinterpret_cast<u32x2*>(slot) = o;
        }
      }
    }
  }
  __syncthreads();
  {
    uint16_t* og = P.out + (int64_t)b * P.out_bstride + (int64_t)t0 * cout + ch0;
    const int npc = rows * ppr;
    u32x4 ov[EPI_PIECES];
#pragma unroll
    for (int i = 0; i < EPI_PIECES; ++i) {       // all LDS reads first, then all stores
      int p = tid + i * NT;
      p = p < npc ? p : npc - 1;
      const int r = p / ppr, c = p - r * ppr;
      ov[i] = *reinterpret_cast<const u32x4*>(xs + r * OP + c * 8);
    }
#pragma unroll
    for (int i = 0; i < EPI_PIECES; ++i) {
      const int p = tid + i * NT;
      if (p < npc) {
        const int r = p / ppr, c = p - r * ppr;
        *reinterpret_cast<u32x4*>(og + (int64_t)r * cout + c * 8) = ov[i];
      }
    }
  }
  if (L.dbg && tid == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long* d = L.dbg + 8ull * (((unsigned long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
    d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = __builtin_amdgcn_s_memtime();
    d[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    d[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    d[6] = (unsigned long long)k; d[7] = 1;
  }
}

bool conv_cl_bf16_supported(int cin, int cout, int k, int dil) {
  if (cin < 16 || cin % 16 || cout < 8 || cout % 8 || k < 1 || dil < 1) return false;
  // narrowest workgroup tile (128 time steps) must fit the 160 KB LDS
  return (int64_t)(128 + (k - 1) * dil) * (cin + 8) * 2 <= 160 * 1024;
}

template <int WN, int WM, int MI, int NI, int G = 0>
static int launch_cl_variant(hipStream_t stream, const ClLaunch& L, int nt, size_t lds_rows_extra, int cin) {
  constexpr int BT = WM * 32 * NI;
  const size_t lds_in = (size_t)(BT + lds_rows_extra) * (size_t)(cin + 8) * 2;
  const size_t lds_out = (size_t)BT * (size_t)(WN * MI * 32 + 8) * 2;        // the epilogue re-uses the tile as [BT][channels]
  const size_t lds = lds_in > lds_out ? lds_in : lds_out;
  if (lds > 160 * 1024) return -2;
  const int ngrp = (nt + WN * MI - 1) / (WN * MI);
  dim3 grid((L.L + BT - 1) / BT, L.B * ngrp, L.nprob);
  auto kern = conv_cl_bf16_kernel<WN, WM, MI, NI, G>;
  ensure_dyn_lds((const void*)kern, lds);
  ClLaunch Lt = L;
  {
    int ks = 0;
    for (int i = 0; i < L.nprob && i < 3; ++i) ks |= (L.p[i].k & 255) << (8 * i);
    Lt.dbg = timeline_slice(grid.x, grid.y, grid.z, -(WN * 1000 + WM * 100 + MI * 10 + NI), ks, cin, L.L);   // negative tile id: bf16
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * WN * WM), lds, stream, Lt, ngrp);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// tuning experiments (bv2_test_set_variants, include/bv2_testing.h — no environment variables in the product path):
// spec "<nt>:<id>[,<nt>:<id>...]" forces variant <id> for launches with <nt> 32-channel tiles; generic = 1 forces the generic GEMM loop
static char g_cl_spec[128] = "";
static bool g_cl_generic = false;
void conv_cl_set_tuning(const char* spec, int generic) {
  std::strncpy(g_cl_spec, spec ? spec : "", sizeof(g_cl_spec) - 1);
  g_cl_spec[sizeof(g_cl_spec) - 1] = 0;
  g_cl_generic = generic != 0;
}
static int forced_variant(int nt) {
  const char* e = g_cl_spec;
  if (!*e) return -1;
  for (const char* p = e; *p;) {
    const int a = atoi(p);
    const char* c = strchr(p, ':');
    if (!c) break;
    const int v = atoi(c + 1);
    if (a == nt) return v;
    const char* n = strchr(c, ',');
    if (!n) break;
    p = n + 1;
  }
  return -1;
}

int launch_conv_cl_bf16(hipStream_t stream, const ClLaunch& L, const char** variant_name) {
  if (L.nprob < 1 || L.nprob > BV2_MAX_PROBS || L.B < 1 || L.L < 1) return -1;
  int nt = 0, extra = 0;
  const int cin = L.p[0].cin;
  for (int i = 0; i < L.nprob; ++i) {
    const ClProb& p = L.p[i];
    if (p.cin != cin || !conv_cl_bf16_supported(p.cin, p.cout, p.k, p.dil) || p.cout_pad % 32 || p.cout_pad < p.cout) return -1;
    if (p.cout_pad / 32 > nt) nt = p.cout_pad / 32;
    if ((p.k - 1) * p.dil > extra) extra = (p.k - 1) * p.dil;
  }
  // variant ids: 0 = 8x1 (8 waves x [32 ch x 128 t]), 1 = 4x1, 2 = 2x2, 3 = 1x4, then the 2 x NI register-blocked forms
  // 4 = 4x1 MI2 NI4 (4 waves x [64 ch x 128 t]), 5 = 2x2 MI2 NI4, 6 = 2x4 MI2 NI2, 7 = 1x4 MI2 NI4, 8 = 4x2 MI2 NI2, 9 = 1x8 MI2 NI2
  int v = forced_variant(nt);
  // measured at B=32 (profiles/r01_j_*): C=64 is fastest as 8 waves x [64 ch x 64 t] (2x2 register blocking: every B fragment
  // feeds two MFMAs), C >= 128 as one 32-channel tile per wave x 128 t (the 2 x NI forms lose more to fewer resident waves
  // than they gain in LDS traffic)
  bool generic = g_cl_generic;
  if (v < 0) {
    v = nt >= 8 ? 0 : (nt >= 3 ? 1 : (nt == 2 ? (cin == 64 ? 9 : 2) : 3));
    // The ConvTranspose1d launches (one or two taps per phase, three summed sources): staging and the output tile are 3/4 of a workgroup's
    // life (tools/timeline.py at B = 32: 15k + 12k ticks around a 7.8k-tick GEMM at C_in = 256), so what counts is how many workgroups a CU
    // holds, not the GEMM loop: the tap-major <8x1> needs 167 registers = ONE 8-wave workgroup per CU.  Per-launch HIP-event times of every
    // variant, same box (tools/ab_ups.py, profiles/r05_ab_ups_variants.txt): C_in = 256: <4x2,2x2> generic (124 registers, two workgroups)
    // 242 -> 217 us; C_in = 64 (a 1 x 1 conv, HBM-bound): <2x2> generic 202 -> 158 us.
    if (L.ups && !generic) {
      if (nt == 32 && cin == 256) { v = 8; generic = true; }
      else if (nt == 2 && cin == 64) { v = 2; generic = true; }
    }
  }
  static const char* names[] = {"conv_cl_bf16<8x1>", "conv_cl_bf16<4x1>", "conv_cl_bf16<2x2>", "conv_cl_bf16<1x4>",
                                "conv_cl_bf16<4x1,2x4>", "conv_cl_bf16<2x2,2x4>", "conv_cl_bf16<2x4,2x2>", "conv_cl_bf16<1x4,2x4>",
                                "conv_cl_bf16<4x2,2x2>", "conv_cl_bf16<1x8,2x2>", "conv_cl_bf16<4x1,t64>", "conv_cl_bf16<8x1,t64>"};
  int r = -1;
  for (int attempt = 0; attempt < 2; ++attempt) {
    switch (v) {
      // C_in known at compile time for the ResBlock widths: tap-major GEMM (conv_cl_set_tuning can force the generic loop)
      case 0: r = (cin == 256 && !generic) ? launch_cl_variant<8, 1, 1, 4, 16>(stream, L, nt, extra, cin)
                                           : launch_cl_variant<8, 1, 1, 4>(stream, L, nt, extra, cin); break;
      case 1: r = (cin == 128 && !generic) ? launch_cl_variant<4, 1, 1, 4, 8>(stream, L, nt, extra, cin)
                                           : launch_cl_variant<4, 1, 1, 4>(stream, L, nt, extra, cin); break;
      case 2: r = (cin == 64 && !generic) ? launch_cl_variant<2, 2, 1, 4, 4>(stream, L, nt, extra, cin)
                                          : launch_cl_variant<2, 2, 1, 4>(stream, L, nt, extra, cin); break;
      case 3: r = launch_cl_variant<1, 4, 1, 4>(stream, L, nt, extra, cin); break;
      case 4: r = (cin == 256 && !generic) ? launch_cl_variant<4, 1, 2, 4, 16>(stream, L, nt, extra, cin)
                                           : launch_cl_variant<4, 1, 2, 4>(stream, L, nt, extra, cin); break;
      case 5: r = (cin == 128 && !generic) ? launch_cl_variant<2, 2, 2, 4, 8>(stream, L, nt, extra, cin)
                                           : launch_cl_variant<2, 2, 2, 4>(stream, L, nt, extra, cin); break;
      case 6: r = (cin == 128 && !generic) ? launch_cl_variant<2, 4, 2, 2, 8>(stream, L, nt, extra, cin)
                                           : launch_cl_variant<2, 4, 2, 2>(stream, L, nt, extra, cin); break;
      case 7: r = (cin == 64 && !generic) ? launch_cl_variant<1, 4, 2, 4, 4>(stream, L, nt, extra, cin)
                                          : launch_cl_variant<1, 4, 2, 4>(stream, L, nt, extra, cin); break;
      case 8: r = (cin == 256 && !generic) ? launch_cl_variant<4, 2, 2, 2, 16>(stream, L, nt, extra, cin)
                                           : launch_cl_variant<4, 2, 2, 2>(stream, L, nt, extra, cin); break;
      case 9: r = (cin == 64 && !generic) ? launch_cl_variant<1, 8, 2, 2, 4>(stream, L, nt, extra, cin)
                                          : launch_cl_variant<1, 8, 2, 2>(stream, L, nt, extra, cin); break;
      // 64-step time tiles: half the MFMA work per workgroup but 31 KB of LDS and <= 96 registers -> five workgroups per CU
      case 10: r = (cin == 128 && !generic) ? launch_cl_variant<4, 1, 1, 2, 8>(stream, L, nt, extra, cin)
                                            : launch_cl_variant<4, 1, 1, 2>(stream, L, nt, extra, cin); break;
      case 11: r = (cin == 256 && !generic) ? launch_cl_variant<8, 1, 1, 2, 16>(stream, L, nt, extra, cin)
                                            : launch_cl_variant<8, 1, 1, 2>(stream, L, nt, extra, cin); break;
      default: return -1;
    }
    if (variant_name) *variant_name = names[v];
    if (r != -2) break;
    v = 1;                                        // tile did not fit the LDS: the narrowest-in-time variant
  }
  return r;
}

double conv_cl_bytes(const ClLaunch& L) {   // each input read once, each output written once (+ residual read), weights once
  double by = 0;
  for (int i = 0; i < L.nprob; ++i) {
    const ClProb& p = L.p[i];
    by += 2.0 * ((double)p.cin * p.nsrc * L.L * L.B + (double)p.cout * L.L * L.B * (p.res ? 2 : 1) +
                 (double)p.cout * p.cin * p.k);
  }
  return by;
}

// ---------------------------------------------------------------------------------------------------------------
// (z * y_mask)[:, :, :L] fp32 [B][C][Ty] -> bf16 channels-last [B][L][C]   (the input of dec.conv_pre, models.py:1073)
__global__ void __launch_bounds__(256) cast_cl_kernel(const float* z, int z_rstride, int64_t z_bstride, const float* mask,
                                                      int mask_bstride, uint16_t* out, int C, int L) {
  const int b = blockIdx.z, c8 = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= L) return;
  const float m = mask ? mask[(int64_t)b * mask_bstride + t] : 1.f;
  const float* src = z + (int64_t)b * z_bstride + (int64_t)(c8 * 8) * z_rstride + t;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = src[(int64_t)e * z_rstride] * m;
  u32x4 o;
  o.x = bf_pack(v[0], v[1]); o.y = bf_pack(v[2], v[3]); o.z = bf_pack(v[4], v[5]); o.w = bf_pack(v[6], v[7]);
  *reinterpret_cast<u32x4*>(out + ((int64_t)b * L + t) * C + c8 * 8) = o;
}

int launch_cast_cl(hipStream_t stream, const float* z, int z_rstride, int64_t z_bstride, const float* mask, int mask_bstride,
                   uint16_t* out, int B, int C, int L) {
  if (C % 8 || B < 1 || L < 1) return -1;
  dim3 grid((L + 255) / 256, C / 8, B);
  hipLaunchKernelGGL(cast_cl_kernel, grid, dim3(256), 0, stream, z, z_rstride, z_bstride, mask, mask_bstride, out, C, L);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// debug taps: bf16 channels-last [B][L][C] -> fp32 [B][C][L]
__global__ void __launch_bounds__(256) uncast_cl_kernel(const uint16_t* x, float* out, int C, int L) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= L) return;
  out[((int64_t)b * C + c) * L + t] = __uint_as_float((unsigned)x[((int64_t)b * L + t) * C + c] << 16);
}

int launch_uncast_cl(hipStream_t stream, const uint16_t* x, float* out, int B, int C, int L) {
  dim3 grid((L + 255) / 256, C, B);
  hipLaunchKernelGGL(uncast_cl_kernel, grid, dim3(256), 0, stream, x, out, C, L);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------------------------------------------
// conv_post + tanh (reference models.py:553-555: leaky_relu default slope 0.01, Conv1d(C, 1, 7, bias=False), tanh) on the
// mean of the last stage's branches; fp32 arithmetic on bf16 inputs, fp32 waveform out.
constexpr int CP_TS = 256;
__global__ void __launch_bounds__(CP_TS) conv_post_cl_kernel(const ConvPostClArgs A) {
  extern __shared__ float cps[];                  // [C*k] weights, then [(CP_TS + k - 1)][C + 1] activations
  const int C = A.C, k = A.k, pad = (k - 1) / 2;
  float* ws = cps;
  float* ms = cps + C * k;
  const int mp = C + 1;
  const int b = blockIdx.y, t0 = blockIdx.x * CP_TS, tid = threadIdx.x;
  int Lv = A.L;
  if (A.lens) {
    const int64_t lv = A.lens[b] * A.len_mul;
    Lv = lv < Lv ? (int)lv : Lv;
  }
  for (int i = tid; i < C * k; i += CP_TS) ws[i] = A.w[i];
  const int rows = CP_TS + k - 1, ppr = C >> 3;
  // every load of the tile in flight before the first is used: (rows * ppr / CP_TS) <= 3 pieces per thread x 3 sources.  The loop form
  // (load the sources of one piece, sum, store to LDS, next piece) ran three dependent HBM round trips per workgroup: 3.1 TB/s on a
  // launch that does nothing but stream 0.6 GB (B = 32)
  constexpr int PQ = 3;
  for (int base = 0; base < rows * ppr; base += PQ * CP_TS) {      // one pass for C = 16, k = 7 (524 pieces of 768)
  u32x4 u[PQ][3];
  int rr[PQ], cbq[PQ];
  bool okq[PQ], inq[PQ];
#pragma unroll
  for (int q = 0; q < PQ; ++q) {
    int p = base + tid + q * CP_TS;
    inq[q] = p < rows * ppr;
    p = inq[q] ? p : rows * ppr - 1;
    const int r = p / ppr, cb = p - r * ppr;
    const int t = t0 - pad + r;
    okq[q] = t >= 0 && t < Lv;
    const int tc = t < 0 ? 0 : (t >= Lv ? Lv - 1 : t);
    const int64_t off = ((int64_t)b * A.L + tc) * C + cb * 8;
    rr[q] = r; cbq[q] = cb;
#pragma unroll
    for (int s = 0; s < 3; ++s) u[q][s] = *reinterpret_cast<const u32x4*>(A.x[s < A.nsrc ? s : 0] + off);
  }
#pragma unroll
  for (int q = 0; q < PQ; ++q) {
    if (!inq[q]) continue;
    float v[8];
#pragma unroll
    for (int w = 0; w < 4; ++w) {                 // the last stage's branch mean with the hand-over's rounding points (cl_bf16.h stage_mean)
      v[2 * w] = stage_mean(bf_lo(u[q][0][w]), bf_lo(u[q][1][w]), bf_lo(u[q][2][w]), A.nsrc, A.in_scale);
      v[2 * w + 1] = stage_mean(bf_hi(u[q][0][w]), bf_hi(u[q][1][w]), bf_hi(u[q][2][w]), A.nsrc, A.in_scale);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = v[e];
      a = a < 0.f ? a * A.slope : a;
      v[e] = okq[q] ? a : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) ms[rr[q] * mp + cbq[q] * 8 + e] = v[e];
  }
  }
  __syncthreads();
  const int t = t0 + tid;
  if (t >= A.L) return;
  float acc = 0.f;
  for (int c = 0; c < C; ++c)
    for (int j = 0; j < k; ++j) acc += ws[c * k + j] * ms[(tid + j) * mp + c];
  A.out[(int64_t)b * A.L + t] = tanhf(acc);
}

// The same for C = 16 (every released config: upsample_initial_channel 512 halved five times; models.py:463 conv_post = Conv1d(ch, 1, 7)),
// row-wise: thread r owns ONE input row (16 channels = 32 contiguous bytes per source, loaded straight into registers: no staged tile),
// forms the K products p_j = sum_c w[c][j] * x[r][c] of its row with every tap, and output t is sum_j p_j of row t - pad + j: K floats per
// thread through LDS instead of the C*K = 112 scalar LDS reads per output of the generic kernel above, which ran this 0.3 GB stream at
// 1.9 TB/s (167 us at B = 32, 1 % of the step).  A workgroup of 256 rows produces 256 - (K - 1) outputs.
template <int K>
__global__ void __launch_bounds__(CP_TS) conv_post_cl16_kernel(const ConvPostClArgs A) {
  constexpr int C = 16, OUTS = CP_TS - (K - 1), pad = (K - 1) / 2;
  __shared__ __attribute__((aligned(16))) float ws[K * C];          // [tap][channel]: a tap's 16 weights are four broadcast ds_read_b128
  __shared__ float ps[K][CP_TS];
  const int b = blockIdx.y, t0 = blockIdx.x * OUTS, tid = threadIdx.x;
  int Lv = A.L;
  if (A.lens) {
    const int64_t lv = A.lens[b] * A.len_mul;
    Lv = lv < Lv ? (int)lv : Lv;
  }
  const int t = t0 - pad + tid;
  const bool ok = t >= 0 && t < Lv;
  const int tc = t < 0 ? 0 : (t >= Lv ? Lv - 1 : t);               // clamped: the loads are unconditional
  const int64_t off = ((int64_t)b * A.L + tc) * C;
  u32x4 u[3][2];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    u[s][0] = *reinterpret_cast<const u32x4*>(A.x[s < A.nsrc ? s : 0] + off);
    u[s][1] = *reinterpret_cast<const u32x4*>(A.x[s < A.nsrc ? s : 0] + off + 8);
  }
  if (tid < K * C) ws[(tid % K) * C + tid / K] = A.w[tid];          // A.w is [c][k]
  float x[C];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float lo = stage_mean(bf_lo(u[0][h][w]), bf_lo(u[1][h][w]), bf_lo(u[2][h][w]), A.nsrc, A.in_scale);
      float hi = stage_mean(bf_hi(u[0][h][w]), bf_hi(u[1][h][w]), bf_hi(u[2][h][w]), A.nsrc, A.in_scale);
      lo = lo < 0.f ? lo * A.slope : lo; hi = hi < 0.f ? hi * A.slope : hi;
      x[8 * h + 2 * w] = ok ? lo : 0.f; x[8 * h + 2 * w + 1] = ok ? hi : 0.f;
    }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < K; ++j) {
    float p = 0.f;
#pragma unroll
    for (int c4 = 0; c4 < C / 4; ++c4) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(ws + j * C + 4 * c4);
      p += wv.x * x[4 * c4]; p += wv.y * x[4 * c4 + 1]; p += wv.z * x[4 * c4 + 2]; p += wv.w * x[4 * c4 + 3];
    }
    ps[j][tid] = p;
  }
  __syncthreads();
  const int to = t0 + tid;
  if (tid >= OUTS || to >= A.L) return;
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < K; ++j) acc += ps[j][tid + j];
  A.out[(int64_t)b * A.L + to] = tanhf(acc);
}

int launch_conv_post_cl(hipStream_t stream, const ConvPostClArgs& a) {
  if (a.C % 8 || a.C > 64 || a.k < 1 || a.k > 15 || a.nsrc < 1 || a.nsrc > 3) return -1;
  if (a.C == 16 && a.k == 7 && !a.generic) {
    dim3 grid((a.L + CP_TS - 7) / (CP_TS - 6), a.B);
    hipLaunchKernelGGL(conv_post_cl16_kernel<7>, grid, dim3(CP_TS), 0, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  dim3 grid((a.L + CP_TS - 1) / CP_TS, a.B);
  const size_t lds = sizeof(float) * ((size_t)a.C * a.k + (size_t)(CP_TS + a.k - 1) * (a.C + 1));
  hipLaunchKernelGGL(conv_post_cl_kernel, grid, dim3(CP_TS), lds, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace bv2

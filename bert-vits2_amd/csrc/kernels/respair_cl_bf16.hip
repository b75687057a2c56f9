// respair_cl_bf16.hip — one (dilated conv, conv) PAIR of HiFi-GAN's ResBlock1 (reference modules.py:296-309:
//     xt = c1(lrelu(x)); xt = c2(lrelu(xt)); x = xt + x)
// in ONE launch for the wide bf16 Generator stages (C = 64 / 128 / 256), channels-last [B][L][C], fp32 accumulate.
//
// Why.  Layer by layer (gen_bf16.hip) the pair is two launches and five tensor passes — x in, t out | t in, x in (residual), out —
// and at B = 32 the k = 3 branch of C = 128 (126 FLOP per HBM byte) and the whole C = 64 stage (180) sit BELOW the machine balance
// (2.5 PF / 8 TB/s = 312): the wide stages were pinned at 0.33 (C = 128), 0.23 (C = 64) of the bf16 MFMA peak with the workgroup
// spending more of its life staging its input tile and assembling its output tile than multiplying (tools/timeline.py: prologue
// + epilogue >= the GEMM for k <= 7).  Here the intermediate never leaves the CU:
//   1. stage x rows [t0 - p2 - p1, + HT + (k-1) d) x C  (HBM -> registers -> bf16(lrelu) -> LDS), ONCE for both convs;
//   2. conv1 (k taps, dilation d) for the HT = 128 / 256 rows of t the tile's outputs need (k - 1 of them are halo, recomputed by
//      the neighbour: 2-8 % extra MFMA work), operands LDS x global weight-fragment ring, no barrier in the loop;
//   3. t = bf16(acc + b1), h = bf16(lrelu(t)) — the two rounding points of the layer-wise path — zero outside [0, L) (conv2's zero
//      padding), written to LDS OVER the x tile (dead by then): one tile's worth of LDS, three workgroups per CU at C <= 128;
//   4. conv2 (k taps, dilation 1) on h;
//   5. out = bf16(acc + b2 + x) with the raw residual rows re-read from L2 (they were fetched in step 1) and the output tile
//      assembled in LDS and moved in 16-byte row pieces, as in gen_bf16.hip.
// Two tensor passes per pair instead of five, one prologue and one HBM epilogue instead of two, the second GEMM's weight ring primed
// under the first one's epilogue.  Bit-identical to the layer-wise path (same unit order in both GEMMs, same rounding points):
// tests/test_respair_gpu.py compares the two with torch.equal.
// out must not alias x (a tile's halo rows are another tile's outputs): the host ping-pongs between two buffers per branch.
#include <hip/hip_runtime.h>
#include "../bv2_kernels.h"
#include "cl_bf16.h"

namespace bv2 {

// workgroup = WN x WM waves: wave (wn, wm) owns channels [32 wn, +32) (WN = C / 32: every output channel of conv1 is needed by conv2)
// and rows [32 NI wm, +32 NI) of the HT = 32 NI WM rows of t / out the tile computes
template <int WN, int WM, int NI, int G>
__global__ void __launch_bounds__(64 * WN * WM, WN * WM == 4 ? 3 : 1)
respair_cl_bf16_kernel(const RpClLaunch L, const int per_xcd, const int mix) {
  constexpr int NT = 64 * WN * WM;
  constexpr int C = 16 * G;
  constexpr int WT = 32 * NI;
  constexpr int HT = WM * WT;
  constexpr int PITCH = C + 8;                    // odd multiple of 16 B: conflict-free ds_read_b128 over 16 consecutive rows
  static_assert(WN * 32 == C, "the workgroup owns every channel");
  extern __shared__ __attribute__((aligned(16))) unsigned short xs[];
  // mix: the problems of the launch (the k = 11 / 7 / 3 branches) interleaved in dispatch order — consecutive workgroups of an XCD
  // cycle through them, so every CU holds an MFMA-bound k = 11 tile next to an HBM-bound k = 3 tile instead of the launch running
  // as three phases that each leave one of the two resources idle
  int bx = blockIdx.x, pz = blockIdx.z;
  if (mix) {
    const int q = bx >> 3;
    pz = q % mix;
    bx = ((q / mix) << 3) | (bx & 7);
  }
  // stage hand-over (L.sum_out): this workgroup runs ALL branches of its tile, p[0] first, and accumulates them into one tensor
  const int nit = L.sum_out ? L.nprob : 1;
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;         // timeline stamps (tools/timeline.py; L.dbg is null in the product)
  if (L.dbg) ts0 = __builtin_amdgcn_s_memtime();
  int kdbg = 0;
  for (int it = 0; it < nit; ++it) {
  // everything per-lane is re-derived from an opaque copy of the thread index inside the loop: hoisted out of it, the staging / epilogue
  // addresses of all phases stayed alive across the whole body and the kernel spilled
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wid % WN, wm = wid / WN;
  const int l31 = lane & 31, lh = lane >> 5;
  // (readfirstlane: a `return` under a loaded length makes the loop divergent to the compiler, and a divergent index turns P into 14 VGPRs)
  const RpClProb P = L.p[__builtin_amdgcn_readfirstlane(L.sum_out ? it : pz)];    // by value: one kernarg round trip
  const int k = P.k, dil = P.dil;
  kdbg = k;
  const int BT = HT - ((L.sum_out ? L.kmax : k) - 1);   // output rows per tile (hand-over: the widest branch's, so every branch has the same tiles)
  // consecutive time tiles share (k-1)(d+1) halo rows: each XCD gets a contiguous range of them, so the re-read hits ITS L2
  const int vt = per_xcd ? (bx & 7) * per_xcd + (bx >> 3) : bx;
  const int t0 = vt * BT;
  if (t0 >= L.L) return;
  const int b = blockIdx.y;
  int Lin = L.L;
  if (L.lens) {                                   // exact lengths: this batch item ends at lens[b]*len_mul
    const int64_t lv = L.lens[b] * L.len_mul;
    Lin = __builtin_amdgcn_readfirstlane(lv < Lin ? (int)lv : Lin);
    if (t0 >= Lin) return;                        // a tile wholly past the utterance: nobody reads its outputs
  }
  const int p2 = (k - 1) / 2, p1 = p2 * dil;
  const int64_t bstride = (int64_t)L.L * C;
  const uint16_t* const xg = P.x + (int64_t)b * bstride;
  const unsigned wlane = 16u * (unsigned)lane;

  // conv1's ring goes out first: its G loads land under the staging
  bf16x8 ar[G];
  const uint16_t* wq[G];
  cl_tm_prime<G>(ar, wq, P.w1 + (int64_t)wn * G * k * 512, wlane, k);
  cl_stage<NT>(xs, PITCH, xg, nullptr, nullptr, 1, 1.f, true, L.slope, t0 - p2 - p1, HT + (k - 1) * dil, C, Lin, tid);
  __syncthreads();
  if (L.dbg) ts1 = __builtin_amdgcn_s_memtime();

  f32x16 acc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
  const unsigned short* const xlane = xs + (wm * WT + l31) * PITCH + lh * 8;
  cl_tm_run<NI, G>(acc, ar, wq, wlane, k, xlane, dil);
  if (L.dbg) ts2 = __builtin_amdgcn_s_memtime();

  // conv2's ring and both bias vectors: in flight under the barrier and the h epilogue
  cl_tm_prime<G>(ar, wq, P.w2 + (int64_t)wn * G * k * 512, wlane, k);
  f32x4 bv1[4], bv2v[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bv1[g] = *reinterpret_cast<const f32x4*>(P.b1 + wn * 32 + 8 * g + 4 * lh);
    bv2v[g] = *reinterpret_cast<const f32x4*>(P.b2 + wn * 32 + 8 * g + 4 * lh);
  }
  __syncthreads();                                // every wave is done reading the x tile
  {
    const float slope = L.slope;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int r = wm * WT + ni * 32 + l31;      // row of t: time t0 - p2 + r
      const int th = t0 - p2 + r;
      const bool inside = th >= 0 && th < Lin;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v0 = acc[ni][4 * g] + bv1[g].x, v1 = acc[ni][4 * g + 1] + bv1[g].y, v2 = acc[ni][4 * g + 2] + bv1[g].z,
                    v3 = acc[ni][4 * g + 3] + bv1[g].w;
        // t = bf16(conv1 + b1) is what the layer-wise path stores; conv2's operand is bf16(lrelu(t))
        const unsigned q0 = bf_pack(v0, v1), q1 = bf_pack(v2, v3);
        float a0 = bf_lo(q0), a1 = bf_hi(q0), a2 = bf_lo(q1), a3 = bf_hi(q1);
        a0 = a0 < 0.f ? a0 * slope : a0; a1 = a1 < 0.f ? a1 * slope : a1;
        a2 = a2 < 0.f ? a2 * slope : a2; a3 = a3 < 0.f ? a3 * slope : a3;
        u32x2 o;
        o.x = bf_pack(a0, a1); o.y = bf_pack(a2, a3);
        if (!inside) o = u32x2{0u, 0u};
        *reinterpret_cast<u32x2*>(xs + r * PITCH + wn * 32 + 8 * g + 4 * lh) = o;
      }
#pragma unroll
      for (int r16 = 0; r16 < 16; ++r16) acc[ni][r16] = 0.f;
    }
  }
  __syncthreads();
  // out row r = time t0 + r needs h rows r .. r + k - 1
  cl_tm_run<NI, G>(acc, ar, wq, wlane, k, xlane, 1);
  if (L.dbg) ts3 = __builtin_amdgcn_s_memtime();

  // ---- epilogue: out = bf16(acc + b2 + x), the tile [BT][C] assembled in LDS, 16-byte row pieces to / from HBM (gen_bf16.hip)
  constexpr int OP = C + 8;
  constexpr int PPR = C / 8;
  constexpr int EPI_PIECES = (HT * PPR + NT - 1) / NT;
  int rows = L.L - t0 < BT ? L.L - t0 : BT;
  const int npc = rows * PPR;
  u32x4 rv[EPI_PIECES];
  {
    const uint16_t* rg = xg + (int64_t)t0 * C;
#pragma unroll
    for (int i = 0; i < EPI_PIECES; ++i) {
      int p = tid + i * NT;
      p = p < npc ? p : npc - 1;
      rv[i] = *reinterpret_cast<const u32x4*>(rg + p * 8);        // a row is C contiguous elements: piece p of the tile is at p * 8
    }
  }
  __syncthreads();                                // every wave is done reading h
#pragma unroll
  for (int i = 0; i < EPI_PIECES; ++i) {
    const int p = tid + i * NT;
    if (p < npc) {
      const int r = p / PPR, c = p - r * PPR;
      *reinterpret_cast<u32x4*>(xs + r * OP + c * 8) = rv[i];
    }
  }
  __syncthreads();
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int tr = wm * WT + ni * 32 + l31;
    if (tr >= rows) continue;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      unsigned short* slot = xs + tr * OP + wn * 32 + 8 * g + 4 * lh;
      const u32x2 rr = *reinterpret_cast<const u32x2*>(slot);
      const float v0 = acc[ni][4 * g] + bv2v[g].x + bf_lo(rr.x), v1 = acc[ni][4 * g + 1] + bv2v[g].y + bf_hi(rr.x),
                  v2 = acc[ni][4 * g + 2] + bv2v[g].z + bf_lo(rr.y), v3 = acc[ni][4 * g + 3] + bv2v[g].w + bf_hi(rr.y);
      u32x2 o;
      o.x = bf_pack(v0, v1); o.y = bf_pack(v2, v3);
      *reinterpret_cast<u32x2*>(slot) = o;
    }
  }
  __syncthreads();
  {
    uint16_t* og = (L.sum_out ? L.sum_out : P.out) + (int64_t)b * bstride + (int64_t)t0 * C;
    u32x4 ov[EPI_PIECES];
#pragma unroll
    for (int i = 0; i < EPI_PIECES; ++i) {        // all LDS reads first, then all stores
      int p = tid + i * NT;
      p = p < npc ? p : npc - 1;
      const int r = p / PPR, c = p - r * PPR;
      ov[i] = *reinterpret_cast<const u32x4*>(xs + r * OP + c * 8);
    }
    if (it > 0) {
      // hand-over: the running sum of the branches before this one — the very pieces THIS thread stored an iteration ago (same piece ->
      // same thread in every iteration), read back from L2 — plus this branch, rounded as cl_bf16.h stage_mean says.  Four pieces at a time:
      // with all of them in flight next to ov[] the kernel spilled
      const bool lastb = it + 1 == nit;
      const float sc = L.sum_scale;
#pragma unroll
      for (int i0 = 0; i0 < EPI_PIECES; i0 += 4) {
        u32x4 pv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int p = tid + (i0 + i) * NT;
          p = p < npc ? p : npc - 1;
          if (i0 + i < EPI_PIECES) pv[i] = *reinterpret_cast<const u32x4*>(og + p * 8);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i0 + i < EPI_PIECES) ov[i0 + i] = stage_accum(ov[i0 + i], pv[i], lastb, sc);
      }
    }
#pragma unroll
    for (int i = 0; i < EPI_PIECES; ++i) {
      const int p = tid + i * NT;
      if (p < npc) *reinterpret_cast<u32x4*>(og + p * 8) = ov[i];
    }
  }
  if (it + 1 < nit) __syncthreads();              // the next branch stages its x tile over this output tile
  }
  if (L.dbg && threadIdx.x == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long* d = L.dbg + 8ull * (((unsigned long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
    d[0] = ts0; d[1] = ts1; d[2] = ts3; d[3] = __builtin_amdgcn_s_memtime();
    d[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    d[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    d[6] = (unsigned long long)kdbg | ((ts2 - ts1) << 16);        // taps | ticks of conv1's GEMM
    d[7] = 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Second form: 64-channel x 128-row wave tiles (MI = 2, NI = 4) on an UNPADDED, XOR-swizzled LDS tile.
//
// Why.  tools/probe/mfma_bf16_probe.hip (profiles/r04_mfma_bf16_probe.txt), full-range random operands: v_mfma_f32_32x32x16_bf16 with
// its operands in registers sustains 0.68-0.73 of the 2.5 PF peak — the chip clocks down to 1.8 GHz under it (0.94-0.99 at 2.4 GHz
// with ZERO operands: the guide's figure) — and with the operand traffic of the form above, one ds_read_b128 per MFMA (a 32-channel
// wave reads 1 KB of B operand for every MFMA) + one global load per 4, 0.45-0.52: that form sat at 0.38-0.40 end to end, MFMA pipe
// 0.58 busy at 1.66 GHz (profiles/r04_b_pmc_c3.json), and neither more overlap (problem interleaving: no change) nor fewer phases
// move a power-bound kernel.  What does is fewer operand bytes per MFMA: a wave that owns TWO 32-channel row blocks uses every B
// fragment twice (one ds_read_b128 per 2 MFMAs + one global load per 4: 0.57-0.59 in the probe).  128 accumulator registers per wave
// mean two waves per SIMD, i.e. bigger tiles per workgroup; they fit because the tile is stored without row padding — 16-byte
// piece p of row r lives at piece p ^ (r & 15) (p ^ ((r >> 1) & 7) for the 128-byte rows of C = 64): conflict-free for ds_read_b128 over
// any 16 consecutive rows, 6 % smaller than the padded pitch — (256 + 50) rows x 256 B = 78 KB at C = 128: two workgroups per CU.
//   C = 64 : 1 x 4 waves, HT = 512 rows,  72 KB, 2 workgroups / CU          C = 256: 4 x 2 waves, HT = 256, 157 KB, 1 workgroup / CU
//   C = 128: 2 x 2 waves, HT = 256 rows,  78 KB, 2 workgroups / CU          (k - 1 halo rows of HT: 2-4 % recomputed instead of 8 %)
// Same unit order (tap-major over the 16-channel groups) and rounding points as the first form: bit-identical to the
// layer-wise path.
namespace {

template <int C>
__device__ __forceinline__ unsigned rp2_sw(unsigned r) { return C == 64 ? ((r >> 1) & 7u) : (r & 15u); }

// byte address of 16-byte piece p of tile row r
template <int C>
__device__ __forceinline__ unsigned rp2_addr(unsigned r, unsigned p) { return r * (2u * C) + ((p ^ rp2_sw<C>(r)) << 4); }

// rows [tb, tb + rows) x C channels: HBM -> bf16(lrelu) -> swizzled LDS; rows outside [0, Lin) are zero.  Every load of a batch is in
// flight before the first is used (QB x 16 B per thread).
template <int C, int NT, int QB>
__device__ __forceinline__ void rp2_stage(unsigned char* lds, const uint16_t* xg, float slope, int tb, int rows, int Lin, int tid) {
  constexpr int PPR = C / 8;
  const int total = rows * PPR;
  for (int base = 0; base < total; base += QB * NT) {
    u32x4 v[QB];
    unsigned dst[QB];
    unsigned okm = 0, inm = 0;
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      int p = base + q * NT + tid;
      const bool inb = p < total;
      p = inb ? p : total - 1;
      const int r = p / PPR, cb = p - r * PPR;
      const int t = tb + r;
      const bool ok = inb && t >= 0 && t < Lin;
      const int tc = t < 0 ? 0 : (t >= Lin ? Lin - 1 : t);           // clamped: the loads are unconditional
      dst[q] = rp2_addr<C>((unsigned)r, (unsigned)cb);
      okm |= ok ? (1u << q) : 0u;
      inm |= inb ? (1u << q) : 0u;
      v[q] = *reinterpret_cast<const u32x4*>(xg + (int64_t)tc * C + cb * 8);
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      u32x4 o;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        float a = bf_lo(v[q][w]), b = bf_hi(v[q][w]);
        a = a < 0.f ? a * slope : a; b = b < 0.f ? b * slope : b;
        o[w] = bf_pack(a, b);
      }
      if (!((okm >> q) & 1u)) o = u32x4{0u, 0u, 0u, 0u};
      if ((inm >> q) & 1u) *reinterpret_cast<u32x4*>(lds + dst[q]) = o;
    }
  }
}

// Weight ring: RS = 4 slots x 2 row blocks.  The units of a GEMM run in tap-major order u = j * G + s (tap j, 16-channel group s) — the
// order of the first form and of the layer-wise kernel, so the fp32 sums are bit-identical — in blocks of four: block q holds groups
// 4 (q % NP) .. + 3 of tap q / NP (NP = G / 4), slot i its i-th unit.  A slot is refilled right behind its MFMAs with the same slot of
// the NEXT block: 4 units = 32 MFMAs = 1024 matrix-pipe cycles ahead of its use.  wq[i][mi] points at the unit the slot loads next.
constexpr int RP2_RS = 4;

// element offset from block q to block q + 1 of a slot's stream (unit (s, j) of a row block starts at (s * k + j) * 512)
template <int NP>
__device__ __forceinline__ int rp2_step(int sb, int k) {
  return sb + 1 < NP ? RP2_RS * k * 512 : 512 - (NP - 1) * RP2_RS * k * 512;
}

template <int NP>
__device__ __forceinline__ void rp2_prime(bf16x8 (&ar)[RP2_RS][2], const uint16_t* (&wq)[RP2_RS][2], const uint16_t* w0, const uint16_t* w1,
                                          unsigned wlane_bytes, int k) {
  const int step = rp2_step<NP>(0, k);             // block 0 -> block 1 (k >= 3: a second block always exists)
#pragma unroll
  for (int i = 0; i < RP2_RS; ++i) {
    wq[i][0] = w0 + (int64_t)i * k * 512;
    wq[i][1] = w1 + (int64_t)i * k * 512;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      ar[i][mi] = *(const GlobalFrag*)(reinterpret_cast<const char*>(wq[i][mi]) + wlane_bytes);
      wq[i][mi] += step;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// acc[mi][ni] += sum over (tap j, group s) W(mi; s, j) x B(s, j; ni).  B(s, j; ni) = channels [16 s + 8 lh, +8) of tile row
// row0 + 32 ni + j * tstep (row0: this lane's first row = a multiple of 32 + l31), read from the swizzled tile.
template <int C>
__device__ __forceinline__ void rp2_run(f32x16 (&acc)[2][4], bf16x8 (&ar)[RP2_RS][2], const uint16_t* (&wq)[RP2_RS][2],
                                        unsigned wlane_bytes, int k, const unsigned char* lds, unsigned row0, int tstep, unsigned lh) {
  constexpr int G = C / 16, NP = G / RP2_RS;
  constexpr unsigned ROWB = 2u * C;
  static_assert(G % RP2_RS == 0, "blocks of four groups");
  bf16x8 bb[2][4];
  // this lane's byte offset inside a row for group s: ((2 s) << 4) ^ y, y = (lh ^ swizzle(row)) << 4; the row's swizzle changes with
  // the tap only (32 ni and the wave's first row are multiples of 32)
  unsigned rowj = row0 * ROWB, yj = (lh ^ rp2_sw<C>(row0)) << 4, rj = row0;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) bb[0][ni] = *reinterpret_cast<const bf16x8*>(lds + rowj + yj + ni * 32 * ROWB);
  const int nblk = k * NP;
  int q = 0;
  for (int j = 0; j < k; ++j) {
    const bool last_tap = j + 1 == k;
    const unsigned rn = last_tap ? rj : rj + (unsigned)tstep;       // after the last tap: re-read the same rows (unused)
    const unsigned rown = rn * ROWB;
    const unsigned yn = (lh ^ rp2_sw<C>(rn)) << 4;
#pragma unroll
    for (int sb = 0; sb < NP; ++sb, ++q) {
      // the slots reload block q + 1 now; afterwards their pointers move on to block q + 2 (or stay on the last block: valid, unused)
      const int sb2 = (sb + 1) % NP;
      const int step = q + 2 < nblk ? rp2_step<NP>(sb2, k) : 0;
#pragma unroll
      for (int i = 0; i < RP2_RS; ++i) {
        {
          const bool wrap = sb + 1 == NP && i + 1 == RP2_RS;          // next unit: group 0 of the next tap
          const unsigned sn = wrap ? 0u : (unsigned)(sb * RP2_RS + i + 1);
          const unsigned an = (wrap ? rown : rowj) + (((2u * sn) << 4) ^ (wrap ? yn : yj));
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) bb[(i & 1) ^ 1][ni] = *reinterpret_cast<const bf16x8*>(lds + an + ni * 32 * ROWB);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[i][mi], bb[i & 1][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          ar[i][mi] = *(const GlobalFrag*)(reinterpret_cast<const char*>(wq[i][mi]) + wlane_bytes);
          wq[i][mi] += step;
        }
        // emitted order: one memory instruction behind each of the first six MFMAs (4 LDS reads of the next unit's B, 2 ring loads)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    rowj = rown; yj = yn; rj = rn;
  }
}

}  // namespace

template <int WN, int WM, int G, bool SUM>         // SUM: the stage hand-over form (L.sum_out), a kernel of its own so that the plain form keeps its registers
__global__ void __launch_bounds__(64 * WN * WM, WN * WM <= 4 ? 2 : 1)
respair2_cl_bf16_kernel(const RpClLaunch L, const int per_xcd, const int mix) {
  constexpr int NT = 64 * WN * WM;
  constexpr int C = 16 * G;
  constexpr int HT = WM * 128;
  constexpr int NP = G / RP2_RS;
  constexpr unsigned ROWB = 2u * C;
  constexpr int PPR = C / 8;
  static_assert(WN * 64 == C, "the workgroup owns every channel");
  extern __shared__ __attribute__((aligned(16))) unsigned char xsb[];
  int bx = blockIdx.x, pz = blockIdx.z;
  if (!SUM && mix) {
    const int q = bx >> 3;
    pz = q % mix;
    bx = ((q / mix) << 3) | (bx & 7);
  }
  const int nit = SUM ? L.nprob : 1;             // stage hand-over: see the first form
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
  const bool dbg = !SUM && L.dbg != nullptr;     // (no timeline in the hand-over form: its stamps cost the scalar registers the loop needs)
  if (dbg) ts0 = __builtin_amdgcn_s_memtime();
  int kdbg = 0;
  for (int it = 0; it < nit; ++it) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));                   // per-lane values re-derived inside the loop (see the first form)
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wid % WN, wm = wid / WN;
  const unsigned l31 = lane & 31, lh = lane >> 5;
  const RpClProb P = L.p[__builtin_amdgcn_readfirstlane(SUM ? it : pz)];
  const int k = P.k, dil = P.dil;
  kdbg = k;
  const int BT = HT - ((SUM ? L.kmax : k) - 1);
  const int vt = per_xcd ? (bx & 7) * per_xcd + (bx >> 3) : bx;
  const int t0 = vt * BT;
  if (t0 >= L.L) return;
  const int b = blockIdx.y;
  int Lin = L.L;
  if (L.lens) {
    const int64_t lv = L.lens[b] * L.len_mul;
    Lin = __builtin_amdgcn_readfirstlane(lv < Lin ? (int)lv : Lin);
    if (t0 >= Lin) return;
  }
  const int p2 = (k - 1) / 2, p1 = p2 * dil;
  const int64_t bstride = (int64_t)L.L * C;
  const uint16_t* const xg = P.x + (int64_t)b * bstride;
  const unsigned wlane = 16u * (unsigned)lane;
  const int64_t mstream = (int64_t)G * k * 512;  // elements of one 32-row block's weight stream

  bf16x8 ar[RP2_RS][2];
  const uint16_t* wq[RP2_RS][2];
  rp2_prime<NP>(ar, wq, P.w1 + (2 * wn) * mstream, P.w1 + (2 * wn + 1) * mstream, wlane, k);
  rp2_stage<C, NT, 20>(xsb, xg, L.slope, t0 - p2 - p1, HT + (k - 1) * dil, Lin, tid);
  __syncthreads();
  if (dbg) ts1 = __builtin_amdgcn_s_memtime();

  f32x16 acc[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  const unsigned row0 = (unsigned)wm * 128u + l31;
  rp2_run<C>(acc, ar, wq, wlane, k, xsb, row0, dil, lh);
  if (dbg) ts2 = __builtin_amdgcn_s_memtime();

  __syncthreads();                                // every wave is done reading the x tile
  {
    const float slope = L.slope;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      f32x4 bv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const f32x4*>(P.b1 + wn * 64 + mi * 32 + 8 * g + 4 * lh);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const unsigned r = row0 + 32u * ni;       // row of t: time t0 - p2 + r
        const int th = t0 - p2 + (int)r;
        const bool inside = th >= 0 && th < Lin;
        const unsigned rb = r * ROWB + lh * 8u, sw = rp2_sw<C>(r);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float v0 = acc[mi][ni][4 * g] + bv[g].x, v1 = acc[mi][ni][4 * g + 1] + bv[g].y, v2 = acc[mi][ni][4 * g + 2] + bv[g].z,
                      v3 = acc[mi][ni][4 * g + 3] + bv[g].w;
          const unsigned q0 = bf_pack(v0, v1), q1 = bf_pack(v2, v3);     // t = bf16(conv1 + b1); conv2's operand = bf16(lrelu(t))
          float a0 = bf_lo(q0), a1 = bf_hi(q0), a2 = bf_lo(q1), a3 = bf_hi(q1);
          a0 = a0 < 0.f ? a0 * slope : a0; a1 = a1 < 0.f ? a1 * slope : a1;
          a2 = a2 < 0.f ? a2 * slope : a2; a3 = a3 < 0.f ? a3 * slope : a3;
          u32x2 o;
          o.x = bf_pack(a0, a1); o.y = bf_pack(a2, a3);
          if (!inside) o = u32x2{0u, 0u};
          const unsigned p = (unsigned)(wn * 8 + mi * 4 + g);
          *reinterpret_cast<u32x2*>(xsb + rb + ((p ^ sw) << 4)) = o;
          acc[mi][ni][4 * g] = 0.f; acc[mi][ni][4 * g + 1] = 0.f; acc[mi][ni][4 * g + 2] = 0.f; acc[mi][ni][4 * g + 3] = 0.f;
        }
      }
    }
  }
  // conv2's ring: in flight across the barrier (primed before the h epilogue it cost 32 registers there: spills)
  rp2_prime<NP>(ar, wq, P.w2 + (2 * wn) * mstream, P.w2 + (2 * wn + 1) * mstream, wlane, k);
  __syncthreads();
  rp2_run<C>(acc, ar, wq, wlane, k, xsb, row0, 1, lh);
  if (dbg) ts3 = __builtin_amdgcn_s_memtime();

  // ---- epilogue: out = bf16(acc + b2 + x): residual rows L2 -> registers -> LDS (swizzled), fragment add in place, rows back out.
  // The residual goes through in two halves — the first in flight across the barrier, the second behind it — so that at most
  // EPI_PIECES / 2 pieces are live next to the 128 accumulator registers (all 16 at once spilled)
  // (the piece -> LDS address arithmetic below is the staging's: without the opaque copy of tid the compiler keeps the staging's
  // addresses alive across both GEMMs for re-use here — 10+ registers next to 128 accumulators: spills)
  int tide = tid;
  unsigned row0e = row0, lhe = lh;
  asm volatile("" : "+v"(tide), "+v"(row0e), "+v"(lhe));
  constexpr int EPI_PIECES = (HT * PPR + NT - 1) / NT;
  constexpr int EH = EPI_PIECES / 2;
  static_assert(EPI_PIECES % 2 == 0, "two halves");
  const int rows = L.L - t0 < BT ? L.L - t0 : BT;
  const int npc = rows * PPR;
  {
    const uint16_t* rg = xg + (int64_t)t0 * C;
    u32x4 ra[EH], rb2[EH];
#pragma unroll
    for (int i = 0; i < EH; ++i) {
      int p = tide + i * NT;
      p = p < npc ? p : npc - 1;
      ra[i] = *reinterpret_cast<const u32x4*>(rg + p * 8);
    }
    __syncthreads();                              // every wave is done reading h
#pragma unroll
    for (int i = 0; i < EH; ++i) {
      int p = tide + (i + EH) * NT;
      p = p < npc ? p : npc - 1;
      rb2[i] = *reinterpret_cast<const u32x4*>(rg + p * 8);
    }
#pragma unroll
    for (int i = 0; i < EH; ++i) {
      const int p = tide + i * NT;
      if (p < npc) {
        const int r = p / PPR, c = p - r * PPR;
        *reinterpret_cast<u32x4*>(xsb + rp2_addr<C>((unsigned)r, (unsigned)c)) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < EH; ++i) {
      const int p = tide + (i + EH) * NT;
      if (p < npc) {
        const int r = p / PPR, c = p - r * PPR;
        *reinterpret_cast<u32x4*>(xsb + rp2_addr<C>((unsigned)r, (unsigned)c)) = rb2[i];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    f32x4 bv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const f32x4*>(P.b2 + wn * 64 + mi * 32 + 8 * g + 4 * lh);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const unsigned r = row0e + 32u * ni;        // (opaque copy: the h epilogue's slot addresses must not stay alive across conv2)
      if ((int)r >= rows) continue;
      const unsigned rb = r * ROWB + lhe * 8u, sw = rp2_sw<C>(r);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const unsigned p = (unsigned)(wn * 8 + mi * 4 + g);
        unsigned char* slot = xsb + rb + ((p ^ sw) << 4);
        const u32x2 rr = *reinterpret_cast<const u32x2*>(slot);
        const float v0 = acc[mi][ni][4 * g] + bv[g].x + bf_lo(rr.x), v1 = acc[mi][ni][4 * g + 1] + bv[g].y + bf_hi(rr.x),
                    v2 = acc[mi][ni][4 * g + 2] + bv[g].z + bf_lo(rr.y), v3 = acc[mi][ni][4 * g + 3] + bv[g].w + bf_hi(rr.y);
        u32x2 o;
        o.x = bf_pack(v0, v1); o.y = bf_pack(v2, v3);
        *reinterpret_cast<u32x2*>(slot) = o;
      }
    }
  }
  __syncthreads();
  {
    // (another opaque copy of the thread index: the residual loads above use the same piece offsets, and kept alive for re-use here — 64-bit
    // each — they spilled in the hand-over form)
    int tids = threadIdx.x;
    asm volatile("" : "+v"(tids));
    uint16_t* og = (SUM ? L.sum_out : P.out) + (int64_t)b * bstride + (int64_t)t0 * C;
    if (SUM && it > 0) {
      // hand-over: running sum (this THREAD's own stores of the previous iteration: same piece -> same thread every iteration, read back
      // from L2) + this branch, rounded as cl_bf16.h stage_mean says.  Four pieces at a time: all sixteen in flight spilled
      const bool lastb = it + 1 == nit;
      const float sc = L.sum_scale;
#pragma unroll
      for (int i0 = 0; i0 < EPI_PIECES; i0 += 4) {
        u32x4 pv[4], o4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int p = tids + (i0 + i) * NT;
          p = p < npc ? p : npc - 1;
          pv[i] = *reinterpret_cast<const u32x4*>(og + p * 8);
          const int r = p / PPR, c = p - r * PPR;
          o4[i] = *reinterpret_cast<const u32x4*>(xsb + rp2_addr<C>((unsigned)r, (unsigned)c));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int p = tids + (i0 + i) * NT;
          if (p < npc) *reinterpret_cast<u32x4*>(og + p * 8) = stage_accum(o4[i], pv[i], lastb, sc);
        }
      }
    } else {
      u32x4 ov[EPI_PIECES];
#pragma unroll
      for (int i = 0; i < EPI_PIECES; ++i) {
        int p = tids + i * NT;
        p = p < npc ? p : npc - 1;
        const int r = p / PPR, c = p - r * PPR;
        ov[i] = *reinterpret_cast<const u32x4*>(xsb + rp2_addr<C>((unsigned)r, (unsigned)c));
      }
#pragma unroll
      for (int i = 0; i < EPI_PIECES; ++i) {
        const int p = tids + i * NT;
        if (p < npc) *reinterpret_cast<u32x4*>(og + p * 8) = ov[i];
      }
    }
  }
  if (SUM && it + 1 < nit) __syncthreads();              // the next branch stages its x tile over this output tile
  }
  if (dbg && threadIdx.x == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long* d = L.dbg + 8ull * (((unsigned long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
    d[0] = ts0; d[1] = ts1; d[2] = ts3; d[3] = __builtin_amdgcn_s_memtime();
    d[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    d[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    d[6] = (unsigned long long)kdbg | ((ts2 - ts1) << 16);
    d[7] = 1;
  }
}

bool respair_cl_bf16_supported(int C, int k, int dil) {
  if (C != 32 && C != 64 && C != 128 && C != 256) return false;
  if (k < 3 || k % 2 == 0 || dil < 1) return false;
  if (C == 32) return (int64_t)(512 + (k - 1) * dil) * (C + 8) * 2 <= 64 * 1024;   // first form only: one wave owns all 32 channels
  const int HT = C == 64 ? 256 : 128;             // the smaller of the two forms' tiles
  if (HT - (k - 1) < HT / 2) return false;        // at least half of conv1's rows are outputs
  const int HT2 = C == 64 ? 512 : 256;            // second form: (HT2 + (k-1) dil) rows of 2 C bytes
  return (int64_t)(HT2 + (k - 1) * dil) * C * 2 <= 160 * 1024 && (int64_t)(HT + (k - 1) * dil) * (C + 8) * 2 <= 160 * 1024;
}

template <int WN, int WM, int NI, int G>
static int launch_rp(hipStream_t stream, const RpClLaunch& L0) {
  constexpr int HT = WM * NI * 32, C = 16 * G;
  int ntx = 0, extra = 0;
  int kmax = 0;
  for (int i = 0; i < L0.nprob; ++i) kmax = L0.p[i].k > kmax ? L0.p[i].k : kmax;
  for (int i = 0; i < L0.nprob; ++i) {
    const int BT = HT - ((L0.sum_out ? kmax : L0.p[i].k) - 1);
    const int n = (L0.L + BT - 1) / BT;
    ntx = n > ntx ? n : ntx;
    const int e = (L0.p[i].k - 1) * L0.p[i].dil;
    extra = e > extra ? e : extra;
  }
  const int per_xcd = ntx >= 16 ? (ntx + 7) / 8 : 0;
  const size_t lds = (size_t)(HT + extra) * (C + 8) * 2;
  const int mix = (per_xcd && L0.nprob > 1 && L0.mix && !L0.sum_out) ? L0.nprob : 0;
  dim3 grid(per_xcd ? per_xcd * 8 : ntx, L0.B, L0.sum_out ? 1 : L0.nprob);
  if (mix) grid = dim3(per_xcd * 8 * L0.nprob, L0.B, 1);
  auto kern = respair_cl_bf16_kernel<WN, WM, NI, G>;
  ensure_dyn_lds((const void*)kern, lds);
  RpClLaunch Lt = L0;
  Lt.kmax = kmax; Lt.sum_scale = 1.f / (float)L0.nprob;
  if (Lt.dbg == nullptr) {
    int ks = 0;
    for (int i = 0; i < L0.nprob && i < 3; ++i) ks |= (L0.p[i].k & 255) << (8 * i);
    Lt.dbg = timeline_slice(grid.x, grid.y, grid.z, -(90000 + C), ks, C, L0.L);
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * WN * WM), lds, stream, Lt, per_xcd, mix);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int WN, int WM, int G>
static int launch_rp2(hipStream_t stream, const RpClLaunch& L0) {
  constexpr int HT = WM * 128, C = 16 * G;
  int ntx = 0, extra = 0;
  int kmax = 0;
  for (int i = 0; i < L0.nprob; ++i) kmax = L0.p[i].k > kmax ? L0.p[i].k : kmax;
  for (int i = 0; i < L0.nprob; ++i) {
    const int BT = HT - ((L0.sum_out ? kmax : L0.p[i].k) - 1);
    const int n = (L0.L + BT - 1) / BT;
    ntx = n > ntx ? n : ntx;
    const int e = (L0.p[i].k - 1) * L0.p[i].dil;
    extra = e > extra ? e : extra;
  }
  const int per_xcd = ntx >= 16 ? (ntx + 7) / 8 : 0;
  const size_t lds = (size_t)(HT + extra) * C * 2;
  const int mix = (per_xcd && L0.nprob > 1 && L0.mix && !L0.sum_out) ? L0.nprob : 0;
  dim3 grid(per_xcd ? per_xcd * 8 : ntx, L0.B, L0.sum_out ? 1 : L0.nprob);
  if (mix) grid = dim3(per_xcd * 8 * L0.nprob, L0.B, 1);
  auto kern = L0.sum_out ? respair2_cl_bf16_kernel<WN, WM, G, true> : respair2_cl_bf16_kernel<WN, WM, G, false>;
  ensure_dyn_lds((const void*)kern, lds);
  RpClLaunch Lt = L0;
  Lt.kmax = kmax; Lt.sum_scale = 1.f / (float)L0.nprob;
  if (Lt.dbg == nullptr) {
    int ks = 0;
    for (int i = 0; i < L0.nprob && i < 3; ++i) ks |= (L0.p[i].k & 255) << (8 * i);
    Lt.dbg = timeline_slice(grid.x, grid.y, grid.z, -(90000 + C), ks, C, L0.L);
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * WN * WM), lds, stream, Lt, per_xcd, mix);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_respair_cl_bf16(hipStream_t stream, const RpClLaunch& L, const char** variant_name) {
  if (L.nprob < 1 || L.nprob > 3 || L.B < 1 || L.L < 1) return -1;
  for (int i = 0; i < L.nprob; ++i) {
    const RpClProb& p = L.p[i];
    if (!respair_cl_bf16_supported(L.C, p.k, p.dil) || !p.x || !p.w1 || !p.w2 || !p.b1 || !p.b2) return -1;
    if (L.sum_out ? p.x == L.sum_out : (!p.out || p.x == p.out)) return -1;
  }
  if (L.form >= 1 && L.C >= 64) {
    switch (L.C) {
      case 64:
        if (variant_name) *variant_name = "respair_cl_bf16<64,64x128>";
        return launch_rp2<1, 4, 4>(stream, L);
      case 128:
        if (variant_name) *variant_name = "respair_cl_bf16<128,64x128>";
        return launch_rp2<2, 2, 8>(stream, L);
      case 256:
        if (variant_name) *variant_name = "respair_cl_bf16<256,64x128>";
        return launch_rp2<4, 2, 16>(stream, L);
    }
    return -1;
  }
  switch (L.C) {
    case 32:                                      // HBM-bound: 4 waves side by side in time, 512 rows of t per tile, 45 KB
      if (variant_name) *variant_name = "respair_cl_bf16<32>";
      return launch_rp<1, 4, 4, 2>(stream, L);
    case 64:
      if (variant_name) *variant_name = "respair_cl_bf16<64>";
      return launch_rp<2, 2, 4, 4>(stream, L);
    case 128:
      if (variant_name) *variant_name = "respair_cl_bf16<128>";
      return launch_rp<4, 1, 4, 8>(stream, L);
    case 256:
      if (variant_name) *variant_name = "respair_cl_bf16<256>";
      return launch_rp<8, 1, 4, 16>(stream, L);
  }
  return -1;
}

double respair_cl_bf16_flops(const RpClLaunch& L) {   // both convs, no halo recompute counted
  double f = 0;
  for (int i = 0; i < L.nprob; ++i) f += 2.0 * 2.0 * L.C * L.C * L.p[i].k * (double)L.L * L.B;
  return f;
}

double respair_cl_bf16_bytes(const RpClLaunch& L) {   // x read once, out written once (hand-over: ONE output for all branches), both weight sets once
  double by = 0;
  for (int i = 0; i < L.nprob; ++i)
    by += 2.0 * ((L.sum_out && i ? 1.0 : 2.0) * L.C * (double)L.L * L.B + 2.0 * L.C * L.C * L.p[i].k);
  return by;
}

}  // namespace bv2

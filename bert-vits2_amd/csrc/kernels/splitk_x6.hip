// splitk_x6.hip — the split-K conv kernel of the small-N regime (conv_mfma.hip conv1d_splitk_kernel: 32 x 32 output tile, K split over
// the workgroup's waves and, into partial slabs, over workgroups) with its products on the bf16 matrix core from exact three-way bf16
// splits of both operands — conv_x6.hip's arithmetic (fp32 operands, fp32 results, six of the nine cross products accumulated in fp32),
// for the FFN convolutions of the Encoder stacks at batch 1 (reference attentions.py:438-446: conv_1 192 -> 768, conv_2 768 -> 192,
// k = 3 in the text encoder, k = 5 in the flow): 47 launches of a 192-launch step.  (VERDICT r4 #3a.)
//
// Why.  tools/timeline_bert.py / tools/timeline.py: a split-K launch is ONE wave of workgroups, so a workgroup's life is the launch, and
// its K loop is the fp32 matrix pipe: 480 v_mfma_f32_32x32x2_f32 per tile (K = 960) are 7.7k cycles of a CU — 15.4k on the 32 CUs that
// get two of the 288 tiles — in a 31k-cycle launch; prefetching the weights into L2 does not shorten it (profiles/r05_ab_prefetch_c2.txt).
// On v_mfma_f32_32x32x16_bf16 the same tile is 60 units x 6 MFMAs x 32 cycles = 2.9k cycles.
//
// MEASURED (profiles/r05_ab_splitk_x6_not_kept.txt): with the planes packed for all 44 FFN convs (+195 MB of blob) config 2 goes 3.611 /
// 3.596 -> 3.584 / 3.588 ms: conv_1 21.1 -> 20.3 us, conv_2 18.1 -> 17.5 us per launch under the per-launch event pass.  tools/timeline
// says why: a split-K launch at batch 1 is 10.2k ticks of prologue (first bytes of x — produced on other XCDs — and of the weights),
// 10.3k of K loop, 1.9k of epilogue, behind a ~3 us launch gap; only the loop shrinks, and not to the MFMA ratio (its LDS reads and
// ring waits stay).  So the product does NOT pack the planes for these convs and TILE_AUTO never meets a problem with w6 in the
// small-N regime; the kernel is reachable through TILE_SPLITK_X6 (tests/test_splitk_x6_gpu.py) and costs nothing where it is not used.
//
// Differences to the fp32 kernel:
//   * weights are the three bf16 planes the packer writes for conv_x6.hip (x6_w_index: a (16-channel group, tap) unit of a 32-row tile =
//     3 KB contiguous), streamed global -> registers through a ring of S6_PD units;
//   * the workgroup's X tile ([its channels][32 + (k-1) dil columns]) is loaded once (lane = column: coalesced), mask / in_scale /
//     pre-activation applied, split into its three planes and written channels-last to LDS (row pitch = odd multiple of 16 bytes:
//     conflict-free ds_read_b128), so a unit's B operand is one ds_read_b128 per plane and taps are row shifts;
//   * K is split over the waves in whole 16-channel groups (the K of one bf16 MFMA); 12 / 8 / 6 / 4 waves by what divides the slice.
// Epilogue (LDS reduction of the waves' partials, bias / per-batch bias, ReLU / GELU, masks, residual, partial slabs) = the fp32 kernel's.
#include <hip/hip_runtime.h>
#include <math.h>
#include "../bv2_kernels.h"

namespace bv2 {

namespace {

typedef __bf16 s6bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 s6bf16x2 __attribute__((ext_vector_type(2)));
typedef float s6f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned s6u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) s6bf16x8 S6GlobalFrag;

constexpr int S6_PD = 4;          // weight ring depth (units of 6 MFMAs, 3 x 16 bytes per lane each)
constexpr int S6_MAXO = 4;        // channel octets a wave stages at most
constexpr int S6_UNIT = 3 * 512;  // elements of one (group, tap) unit: 3 planes x 64 lanes x 8

__device__ __forceinline__ float s6_ld(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ unsigned s6_pack(float a, float b) {     // round-to-nearest-even (v_cvt_pk_bf16_f32)
  s6bf16x2 r;
  r[0] = (__bf16)a; r[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float s6_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float s6_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
// the three planes of a pair of values (conv_x6.hip store_x: plane 1 saturates at the largest bf16, so FLT_MAX splits into finite planes)
__device__ __forceinline__ void s6_split2(float a, float b, unsigned& u1, unsigned& u2, unsigned& u3) {
  constexpr float M = 3.38953139e38f;
  u1 = s6_pack(__builtin_amdgcn_fmed3f(a, -M, M), __builtin_amdgcn_fmed3f(b, -M, M));
  a -= s6_lo(u1); b -= s6_hi(u1);
  u2 = s6_pack(a, b);
  a -= s6_lo(u2); b -= s6_hi(u2);
  u3 = s6_pack(a, b);
}

}  // namespace

template <bool MASK, int NWV>
__global__ void __launch_bounds__(64 * NWV) conv1d_splitk_x6_kernel(const ConvLaunch L, const int mtiles, const int ntiles, const int per_xcd,
                                                                    const int total, const int pitch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s6_raw[];
  unsigned short* const Xs = reinterpret_cast<unsigned short*>(s6_raw);                // [3 planes][XW columns][pitch] bf16
  float (*red)[32][33] = reinterpret_cast<float (*)[32][33]>(s6_raw);                  // [NWV][32][33]: aliases the X tile (barrier between)
  // XCD-aware placement (conv_mfma.hip): consecutive virtual ids (which share a weight slice) land on the same XCD
  const int bid = blockIdx.x;
  const int v = (bid & 7) * per_xcd + (bid >> 3);
  if (v >= total) return;
  int rem = v;
  const int nt = rem % ntiles; rem /= ntiles;
  const int z = rem % L.ksplit; rem /= L.ksplit;
  const int mt = rem % mtiles; rem /= mtiles;
  const int b = rem % L.B;
  const ConvProb P = L.p[0];                        // BY VALUE: one kernarg round trip (see conv1d_splitk_kernel)
  const int m0 = mt * 32, t0 = nt * 32;
  if (m0 >= P.cout_pad) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned l31 = lane & 31, lh = lane >> 5;
  const int k = P.k, dil = P.dil, cin = P.cin;
  int Lin = P.Lin;
  if (L.lens) {
    const int64_t lv = L.lens[b] * L.len_mul;
    Lin = lv < Lin ? (int)lv : Lin;
  }
  const int groups = cin / 16;                      // 16-channel groups = K steps of one bf16 MFMA
  const float in_scale = P.in_scale, slope = P.slope;
  const bool lrelu = P.pre_act == PRE_LRELU;
  const float* const xp = P.x[0] + (int64_t)b * P.x_bstride;
  const float* const mp = MASK ? P.in_mask + (int64_t)b * P.in_mask_bstride : nullptr;
  const unsigned x_rs4 = 4u * (unsigned)P.x_rstride;
  // this workgroup's K slice [G0, G1) and this wave's share [g0, g1) of it, in 16-channel groups (balanced)
  const int G0 = (int)(((unsigned)groups * (unsigned)z) / (unsigned)L.ksplit);
  const int G1 = (int)(((unsigned)groups * (unsigned)(z + 1)) / (unsigned)L.ksplit);
  const int g0 = G0 + (int)(((unsigned)(G1 - G0) * (unsigned)wid) / (unsigned)NWV);
  const int g1 = G0 + (int)(((unsigned)(G1 - G0) * (unsigned)(wid + 1)) / (unsigned)NWV);
  const int U = (g1 - g0) * k;

  // ---- weight ring: unit (g, j) of m-tile mt = 3 planes x 1 KB at w6 + ((mt * groups + g) * k + j) * S6_UNIT
  s6bf16x8 ar[S6_PD][3];
  const uint16_t* wq = P.w6 + ((int64_t)mt * groups + g0) * k * S6_UNIT;       // wave-uniform; units of a wave are contiguous
  const unsigned wlane = 16u * (unsigned)lane;
  int lu = 0;
  auto load_unit = [&](int slot) __attribute__((always_inline)) {
    const int uc = lu < U ? lu : (U > 0 ? U - 1 : 0);                          // past the end: re-read the last unit (unused)
    const char* base = reinterpret_cast<const char*>(wq + (int64_t)uc * S6_UNIT) + wlane;
#pragma unroll
    for (int p = 0; p < 3; ++p) ar[slot][p] = *(const S6GlobalFrag*)(base + 1024 * p);
    ++lu;
  };
#pragma unroll
  for (int i = 0; i < S6_PD; ++i) { load_unit(i); __builtin_amdgcn_sched_barrier(0); }

  // ---- epilogue operands first (see conv1d_splitk_kernel): every field, the bias / residual values of this thread's outputs
  const int col = t0 + (tid & 31);
  const bool colok = col < L.L;
  const int cout = P.cout, act = P.act, mask_pre = P.mask_pre, mask_post = P.mask_post, res_mode = P.res_mode;
  const unsigned o_rs = (unsigned)P.out_rstride, o_ts = (unsigned)P.out_tstride, o_to = (unsigned)P.out_toff;
  float* const outb = P.out + (int64_t)z * L.slab_stride + (int64_t)b * P.out_bstride;
  const float* const resb = (res_mode != RES_NONE && z == 0) ? P.res + (int64_t)b * P.res_bstride : nullptr;
  const float* const biasp = z == 0 ? P.bias : nullptr;
  const float* const bias2p = (z == 0 && P.bias2) ? P.bias2 + (int64_t)b * P.bias2_bstride : nullptr;
  const float om = (P.out_mask && colok) ? P.out_mask[(int64_t)b * P.out_mask_bstride + col] : 1.f;
  const unsigned coff = (unsigned)(colok ? col : 0) * o_ts + o_to;
  constexpr int RPP = 2 * NWV;                      // rows per pass (one element per thread per pass)
  constexpr int NPASS = (32 + RPP - 1) / RPP;
  float rvv[NPASS], bsv[NPASS], b2v[NPASS];
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    const int rl = (tid >> 5) + RPP * i;
    int row = m0 + (rl < 32 ? rl : 31);
    row = row < cout ? row : cout - 1;
    bsv[i] = biasp ? biasp[row] : 0.f;
    b2v[i] = bias2p ? bias2p[row] : 0.f;
    rvv[i] = resb ? s6_ld(resb, 4u * ((unsigned)row * o_rs + coff)) : 0.f;
  }

  // ---- stage the X tile: wave `wid` takes channel octets wid, wid + NWV, ... of the slice; lane = column
  const int XW = 32 + (k - 1) * dil;
  const int plane = XW * pitch;                     // elements per plane
  {
    const int noct = 2 * (G1 - G0);
    const int t = t0 - P.pad_left + lane;
    const bool tok = lane < XW && t >= 0 && t < Lin;
    const unsigned tcl = 4u * (unsigned)(t < 0 ? 0 : (t >= Lin ? Lin - 1 : t));
    const float mval = MASK ? s6_ld(mp, tcl) : 1.f;
    float xv[S6_MAXO][8];
#pragma unroll
    for (int i = 0; i < S6_MAXO; ++i) {
      int o = wid + NWV * i;
      o = o < noct ? o : noct - 1;                  // clamped: the loads are unconditional
      const unsigned row0 = (unsigned)(16 * G0 + 8 * o) * x_rs4 + tcl;
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[i][e] = s6_ld(xp, row0 + (unsigned)e * x_rs4);
    }
    const float cs = tok ? in_scale * mval : 0.f;
#pragma unroll
    for (int i = 0; i < S6_MAXO; ++i) {
      const int o = wid + NWV * i;
      s6u32x4 q1, q2, q3;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        float a = xv[i][2 * w], bq = xv[i][2 * w + 1];
        const float an = a * slope, bn = bq * slope;
        a = (lrelu && a < 0.f) ? an : a;
        bq = (lrelu && bq < 0.f) ? bn : bq;
        a *= cs; bq *= cs;
        unsigned u1, u2, u3;
        s6_split2(a, bq, u1, u2, u3);
        q1[w] = u1; q2[w] = u2; q3[w] = u3;
      }
      if (o < noct && lane < XW) {
        unsigned short* dst = Xs + lane * pitch + 8 * o;
        *reinterpret_cast<s6u32x4*>(dst) = q1;
        *reinterpret_cast<s6u32x4*>(dst + plane) = q2;
        *reinterpret_cast<s6u32x4*>(dst + 2 * plane) = q3;
      }
    }
  }
  __syncthreads();

  // ---- main loop: units (g, j) of this wave in the order the ring was loaded; six cross products per unit, smallest terms first
  s6f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  {
    const unsigned short* const xlane = Xs + l31 * pitch + 16 * (g0 - G0) + 8 * lh;
    int ug = 0, uj = 0;                             // group (relative to g0) and tap of the current unit
    s6bf16x8 bb[2][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) bb[0][p] = *reinterpret_cast<const s6bf16x8*>(xlane + p * plane);
    for (int u0 = 0; u0 < U; u0 += S6_PD) {
#pragma unroll
      for (int i = 0; i < S6_PD; ++i) {
        if (u0 + i < U) {
          int jn = uj + 1, gn = ug;
          if (jn == k) { jn = 0; ++gn; }
          const bool more = u0 + i + 1 < U;
          const unsigned short* xn = xlane + (more ? gn : ug) * 16 + (more ? jn : uj) * dil * pitch;
#pragma unroll
          for (int p = 0; p < 3; ++p) bb[(i & 1) ^ 1][p] = *reinterpret_cast<const s6bf16x8*>(xn + p * plane);
#define S6_PROD(WP, XP) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[i][WP], bb[i & 1][XP], acc, 0, 0, 0);
          S6_PROD(2, 0) S6_PROD(1, 1) S6_PROD(0, 2) S6_PROD(1, 0) S6_PROD(0, 1) S6_PROD(0, 0)
#undef S6_PROD
          uj = jn; ug = gn;
        }
        load_unit(i);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);   // the next unit's B planes first, then the six MFMAs, then the ring refill
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __syncthreads();                                  // every wave is done with the X tile before `red` overwrites it

  // ---- reduce the waves' partial tiles through LDS, then the fp32 kernel's epilogue
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wid][(r & 3) + 8 * (r >> 2) + 4 * lh][l31] = acc[r];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    const int rl = (tid >> 5) + RPP * i;
    if (rl >= 32) break;
    const int row = m0 + rl;
    float vv = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) vv += red[w][rl][tid & 31];
    vv += bsv[i] + b2v[i];
    if (act == ACT_RELU) vv = vv < 0.f ? 0.f : vv;               // a select: NaN stays NaN (host guarantees act == NONE when ksplit > 1)
    else if (act == ACT_GELU) vv = 0.5f * vv * (1.0f + erff(vv * 0.70710678118654752440f));
    if (mask_pre) vv *= om;
    if (res_mode == RES_ADD) vv += rvv[i];
    else if (res_mode == RES_RSUB) vv = rvv[i] - vv;             // (slabs z > 0 carry no residual: rvv = 0)
    if (mask_post) vv *= om;
    if (colok && row < cout) outb[(unsigned)row * o_rs + coff] = vv;
  }
}

// Which launches take this kernel: one problem, planes present, whole 16-channel groups, ReLU / GELU / no activation (no gate),
// a staged tile of at most 64 columns, and a K slice whose octets fit S6_MAXO per wave at the wave count picked below.
static int s6_pick_waves(int slice_groups) {
  const int cand[4] = {12, 8, 6, 4};
  for (int c : cand)
    if (slice_groups % c == 0 && 2 * slice_groups <= S6_MAXO * c) return c;
  for (int c : cand)
    if (slice_groups >= c && 2 * slice_groups <= S6_MAXO * c) return c;
  return 0;
}

bool splitk_x6_supported(const ConvLaunch& L) {
  if (L.nprob != 1 || L.ksplit < 1) return false;
  const ConvProb& p = L.p[0];
  if (!p.w6 || p.nsrc != 1 || p.cin % 16 || p.cin != p.cin_pad || p.cout_pad % 32 || p.k < 1 || p.dil < 1) return false;
  if (32 + (p.k - 1) * p.dil > 64) return false;
  if (p.act != ACT_NONE && p.act != ACT_RELU && p.act != ACT_GELU) return false;
  const int groups = p.cin / 16;
  if (groups < L.ksplit * 4) return false;
  const int slice = (groups + L.ksplit - 1) / L.ksplit;           // the largest slice
  if ((int64_t)p.cin * p.x_rstride >= (1ll << 29)) return false;  // 32-bit byte offsets inside a batch item
  return s6_pick_waves(slice) > 0;
}

template <bool MASK, int NWV>
static void launch_s6(hipStream_t stream, const ConvLaunch& L, dim3 grid, int mtiles, int ntiles, int per_xcd, int total, int pitch, size_t lds) {
  auto kern = conv1d_splitk_x6_kernel<MASK, NWV>;
  ensure_dyn_lds((const void*)kern, lds);
  hipLaunchKernelGGL(kern, grid, dim3(64 * NWV), lds, stream, L, mtiles, ntiles, per_xcd, total, pitch);
}

int launch_splitk_x6(hipStream_t stream, const ConvLaunch& L, const char** variant_name) {
  if (!splitk_x6_supported(L)) return -2;
  const ConvProb& p = L.p[0];
  const int mtiles = p.cout_pad / 32, ntiles = (L.L + 31) / 32;
  const int total = L.B * mtiles * L.ksplit * ntiles;
  const int per_xcd = (total + 7) / 8;
  const int groups = p.cin / 16, slice = (groups + L.ksplit - 1) / L.ksplit;
  const int nw = s6_pick_waves(slice);
  const int XW = 32 + (p.k - 1) * p.dil;
  const int pitch = 16 * slice + 8;                                // bf16 elements per staged column: an odd multiple of 16 bytes
  size_t lds = (size_t)3 * XW * pitch * 2;
  const size_t red = sizeof(float) * (size_t)nw * 32 * 33;
  if (red > lds) lds = red;
  if (lds > 160 * 1024) return -2;
  const dim3 grid(per_xcd * 8);
  const bool mask = p.in_mask != nullptr;
  if (variant_name) *variant_name = nw == 12 ? "conv1d_splitk_x6<32x32,12w>" : (nw == 8 ? "conv1d_splitk_x6<32x32,8w>" :
                                    (nw == 6 ? "conv1d_splitk_x6<32x32,6w>" : "conv1d_splitk_x6<32x32,4w>"));
#define S6_GO(NW_)                                                                                            \
  if (mask) launch_s6<true, NW_>(stream, L, grid, mtiles, ntiles, per_xcd, total, pitch, lds);                \
  else launch_s6<false, NW_>(stream, L, grid, mtiles, ntiles, per_xcd, total, pitch, lds);
  switch (nw) {
    case 12: S6_GO(12) break;
    case 8: S6_GO(8) break;
    case 6: S6_GO(6) break;
    default: S6_GO(4) break;
  }
#undef S6_GO
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace bv2

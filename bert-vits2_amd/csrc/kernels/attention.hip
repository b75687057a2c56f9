// attention.hip — MultiHeadAttention.attention with windowed relative positions (reference attentions.py:273-322 and the
// skew helpers :324-395), flash-style on the fp32 matrix core, never materialising the [B,H,T,T] score tensor.
//
// What the reference does with zero-padded [2T-1]-wide relative matmuls + pad/reshape skew tricks is, mathematically,
//    logit[i][j] = q_i·k_j/sqrt(d) + (|j-i| <= W ? q_i·Ek[j-i+W]/sqrt(d) : 0);   masked pairs are SET to -1e4
//    out[i]      = sum_j p[i][j] v_j  +  sum_{|r|<=W, 0<=i+r<T} p[i][i+r] Ev[r+W]
// (verified bit-for-bit against the reference in oracle/).  So only 2W+1 = 9 extra dot products
// per query are needed instead of the reference's padded matmuls (which triple its attention FLOPs) — and those nine
// are not even computed here: q_i·Ek[r]/sqrt(d) is linear in the layer input, so the fused q/k/v projection emits
// them as 2W+1 extra output rows per head (weights Ek·Wq/sqrt(d), folded at pack time).
//
// Mapping (latency-first: at batch 1 the whole op is ~0.1 GFLOP): one workgroup = 32 queries of one (batch, head); it
// has one wave per 32-key tile (up to 8 waves, round-robin beyond), flash-decoding style, merged through LDS.
//   S^T tile  [32 keys x 32 queries] = K^T·Q : A[m=key][kk=c] = k[c][j] and B[kk=c][n=query] = q[c][i] are both natural
//             row reads of the [C][T] layout (time contiguous) — no transposes anywhere.
//   D layout: lane holds ONE query column (l&31) and 16 key rows -> row max / row sum are in-lane + one shfl_xor(32).
//   O^T tile  [D x 32 queries] += V·P^T with the K-steps taken in the D layout's own row order: step r pairs the keys
//             rowmap(r,0) / rowmap(r,1), so the B operand is the probability register S[r] itself (no LDS round trip)
//             and the A operand v[c][rowmap(r,lh)] comes as aligned float4s (4 consecutive keys) straight from global.
//
// Key split (A.ksplit > 1, fused conv_o form only; batch 1 leaves a launch at T_y/32 x H = 24 workgroups on a 256-CU part, each
// walking 12 key tiles): the key tiles of a (head, query tile) are dealt to `ksplit` workgroups.  Workgroup r normalises over ITS keys
// only (running max m_r, sum l_r), pushes that partial through conv_o — the projection is linear — into slab h*ksplit + r, and
// writes (m_r, l_r) per query.  The LayerNorm that sums the slabs anyway weighs slab (h, r) by
//     w_r = l_r e^{m_r - M} / sum_r' l_r' e^{m_r' - M},   M = max_r m_r
// — exactly the flash-decoding merge, with no extra launch and no cross-workgroup hand-over inside the kernel (an agent-scope fence
// costs more than a launch on this 8-XCD part: tools/probe/grid_barrier.hip).  Relative-value band terms belong to the workgroup that
// owns the key; bias and residual move to the LayerNorm.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <math.h>
#include "../bv2_kernels.h"

namespace bv2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// wave-uniform base (SGPR pair) + 32-bit per-lane BYTE offset: one VGPR per address instead of a 64-bit pair
__device__ __forceinline__ float ld_off(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ f32x4 ld_off4(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byte_off);
}

constexpr int AQ = 32;          // queries per workgroup
constexpr int AK = 32;          // keys per tile
constexpr int AMAXW = 8;        // max window
constexpr int ANS = 4;          // merge slots

// F16 = true (bv2_set_flow_dtype(BV2_F16)): the two matrix products run on v_mfma_f32_32x32x16_f16 — 12 MFMAs per key tile
// instead of 96 fp32 ones.  Nothing else changes: K / V / Q are still read as fp32 and rounded to fp16 in registers, the
// logits, softmax, running max / sum, output accumulators and the merge stay fp32.  The operand fragments need no data
// movement: an fp16 MFMA takes 8 K-indices per lane, and any K-index <-> channel (or key) assignment works as long as both
// operands use the same one, so
//   S^T step t (16 channels): A = the lane's kreg[8t .. 8t+7] (channels 2(8t+e)+lh), B = the same channels of Q, which the
//                             staging loop stores in LDS as fp16 in exactly that order (one ds_read_b128 per step);
//   O^T step s (16 keys):     B = the lane's S registers 8s .. 8s+7 (its own D-layout rows), A = vreg[m][2s], vreg[m][2s+1]
//                             (the float4s that hold exactly those keys).
// KV16 = true (F16 only, round 6; A.kh / A.vh): K and V arrive as fp16 from the q/k/v projection's epilogue (enc_f16.hip HcProb::k16 / v16) instead of
// as fp32 rows of `qkv` — K channels-last [B][ld][HD] so that the 8 K-indices of lane (key, lh) at step t are the 16 bytes [16t + 8lh, +8) of the
// key's row (6 loads of 16 B per key tile at D = 96 instead of 48 dword loads + 48 conversions), V channel-major [B][HD][ld] (the two groups of
// four consecutive keys a step needs are two 8-byte loads: half the bytes of the fp32 float4s).  The values are the same fp16 numbers (the
// rounding moved from this kernel's registers to the projection's epilogue); only the K-index <-> channel assignment of the S^T MFMAs differs.
template <int DT, int NW, bool F16, bool ONE, bool KV16 = false>   // D = 32*DT head channels, NW waves; ONE: at most one key tile per wave (no loop)
__global__ void __launch_bounds__(64 * NW) attention_kernel(const AttnArgs A) {
  static_assert(!KV16 || F16, "fp16 K / V only feed the fp16 matrix products");
  constexpr int D = 32 * DT;
  constexpr int NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps everything derived from it in SGPRs
  const int l31 = lane & 31, lh = lane >> 5;
  const int KS = A.ksplit > 1 ? A.ksplit : 1;
  int b = blockIdx.z, h = blockIdx.y, bx = blockIdx.x;
  if (A.xcd_b) {                                             // batch item -> XCD affinity (bv2_kernels.h xcd_decode)
    int r;
    if (!xcd_decode(blockIdx.x, A.xcd_per, A.B, b, r)) return;
    bx = r % A.xcd_gx; h = r / A.xcd_gx;
  }
  const int kr = KS > 1 ? bx % KS : 0;                       // this workgroup's key range
  const int i0 = (KS > 1 ? bx / KS : bx) * AQ;
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0;  // timeline stamps (tools/timeline.py; A.dbg is null in the product)
  if (A.dbg) ts0 = __builtin_amdgcn_s_memtime();
  const int T = A.T, W = A.W, NR = 2 * W + 1, ld = A.ld;
  const int HD = A.H * D;
  const int R = 3 * HD + A.H * NR;

  float* Os = smem;                                 // [ANS][D][AQ] partial outputs
  float* Qs = Os + ANS * D * AQ;                    // [D][AQ] query tile (pre-scaled by the projection)
  float* Ev = Qs + D * AQ;                          // [2*AMAXW+1][D] relative-value embedding
  float* Sb = Ev + (2 * AMAXW + 1) * D;             // [NR][AQ] raw band logits -> band probabilities
  float* Mw = Sb + (2 * AMAXW + 1) * AQ;            // [NW][AQ]
  float* Lw = Mw + NW * AQ;                         // [NW][AQ]
  float* Qe = Lw + NW * AQ;                         // [NR][AQ] relative-key logits q_i.Ek[r]/sqrt(d) of this query tile

  const float* base = A.qkv + (int64_t)b * R * ld;
  const float* qp = base + (int64_t)(h * D) * ld;
  const float* kp = qp + (int64_t)HD * ld;
  const float* vp = kp + (int64_t)HD * ld;
  const float* qe = base + (int64_t)(3 * HD + h * NR) * ld;
  const float* mp = A.mask + (int64_t)b * T;

  const int iq = i0 + l31;                          // this lane's query
  const bool iok = iq < T;
  const int ntiles_all = (T + AK - 1) / AK;
  const int kt0 = (ntiles_all * kr) / KS, kt1 = (ntiles_all * (kr + 1)) / KS;   // key tiles [kt0, kt1) are this workgroup's
  const int ntiles = kt1 - kt0;

  // ---- everything this wave's first key tile needs goes in flight at once (one memory round trip):
  //      K tile -> D/2 registers + key mask now; the V tile (D/8 float4) is issued as soon as the K registers are consumed
  float kreg[KV16 ? 1 : D / 2];
  f32x4 vreg[KV16 ? 1 : DT][4];
  u32x4 kq[KV16 ? D / 16 : 1];                      // KV16: step t's 8 K-indices of this lane's key, packed fp16
  u32x2 vq[KV16 ? DT : 1][2][2];                    // KV16: [m][step][keys ja.. / ja+8..] of channel m*32 + l31, packed fp16
  const uint16_t* const khp = KV16 ? A.kh + (int64_t)b * ld * HD + h * D : nullptr;
  const uint16_t* const vhp = KV16 ? A.vh + ((int64_t)b * HD + h * D) * ld : nullptr;
  float mkey = 0.f;
  const unsigned koff = 4u * (unsigned)(lh * ld + l31);         // per-lane byte offsets, shared by every load of a tile
  const unsigned voff = 4u * (unsigned)(l31 * ld + 4 * lh);
  auto issue_k = [&](int j0) __attribute__((always_inline)) {
    if constexpr (KV16) {
      const uint16_t* kr_ = khp + (int64_t)(j0 + l31) * HD + 8 * lh;     // rows up to ld exist (ld = T rounded up to 32)
#pragma unroll
      for (int t = 0; t < D / 16; ++t) kq[t] = *reinterpret_cast<const u32x4*>(kr_ + 16 * t);
    } else {
#pragma unroll
    for (int s = 0; s < D / 2; ++s) kreg[s] = ld_off(kp, koff + 4u * (unsigned)(j0 + 2 * s * ld));
    }
    const int jm = j0 + l31;
    mkey = mp[jm < T ? jm : T - 1];
  };
  auto issue_v = [&](int j0) __attribute__((always_inline)) {
    if constexpr (KV16) {
#pragma unroll
      for (int m = 0; m < DT; ++m) {
        const uint16_t* vr_ = vhp + (int64_t)(m * 32 + l31) * ld + j0 + 4 * lh;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          vq[m][s2][0] = *reinterpret_cast<const u32x2*>(vr_ + 16 * s2);
          vq[m][s2][1] = *reinterpret_cast<const u32x2*>(vr_ + 16 * s2 + 8);
        }
      }
    } else {
#pragma unroll
    for (int m = 0; m < DT; ++m)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) vreg[m][g4] = ld_off4(vp, voff + 4u * (unsigned)((m * 32) * ld + j0 + 8 * g4));
    }
  };
  const bool have_tile = wid < ntiles;
  if (have_tile) {
    issue_k((kt0 + wid) * AK);
  }
  // the query's mask value: loaded unconditionally (clamped) and selected after the staging barrier — as `iok ? mp[iq] : 0` the load sat
  // in a branch that ended with s_waitcnt vmcnt(0), i.e. the whole K tile's round trip BEFORE the query loads were even issued
  const float mi_raw = mp[iok ? iq : T - 1];
  // query tile and Ev -> LDS (all threads)
  constexpr int QP = D + 8;                         // fp16 row pitch of the transposed query tile (odd multiple of 16 B)
  _Float16* Qh = reinterpret_cast<_Float16*>(Qs);   // F16: [AQ][QP], row i holds Q[2(8t+e)+lh][i] at t*16 + lh*8 + e
  {
    // all of a thread's query loads in flight at once (the piece-per-iteration loop took D*AQ/NT serial L2 round trips: 9-12k
    // cycles before the first MFMA, tools/timeline.py)
    constexpr int QPT = (D * AQ + NT - 1) / NT;
    float qv[QPT];
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
      int e = tid + q * NT;
      e = e < D * AQ ? e : D * AQ - 1;
      qv[q] = qp[(e >> 5) * ld + i0 + (e & 31)];    // columns beyond T: finite-or-not garbage, dead columns below
    }
    // ... and so are the relative-value table and the relative-key logit rows of the tile (ISA of round 2: each `for (e = tid; ...)
    // LDS[e] = global[e]` loop iteration was a load -> s_waitcnt vmcnt(0) -> ds_write of its own, four serial round trips behind the
    // query tile's: most of the 7k-cycle staging phase of a workgroup that lives 36-60k cycles)
    constexpr int EVPT = ((2 * AMAXW + 1) * D + NT - 1) / NT, QEPT = ((2 * AMAXW + 1) * AQ + NT - 1) / NT;
    float evv[EVPT], qev[QEPT];
#pragma unroll
    for (int q = 0; q < EVPT; ++q) {
      int e = tid + q * NT;
      e = e < NR * D ? e : NR * D - 1;
      evv[q] = A.erv[e];
    }
#pragma unroll
    for (int q = 0; q < QEPT; ++q) {
      int e = tid + q * NT;
      e = e < NR * AQ ? e : NR * AQ - 1;
      const int ic = i0 + (e & 31) < T ? i0 + (e & 31) : T - 1;
      qev[q] = qe[(e >> 5) * ld + ic];
    }
#pragma unroll
    for (int q = 0; q < EVPT; ++q) {
      const int e = tid + q * NT;
      if (e < NR * D) Ev[e] = evv[q];
    }
#pragma unroll
    for (int q = 0; q < QEPT; ++q) {
      const int e = tid + q * NT;
      if (e < NR * AQ) Qe[e] = (i0 + (e & 31) < T) ? qev[q] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
      const int e = tid + q * NT;
      if (e >= D * AQ) continue;
      const int c = e >> 5, i = e & 31;
      if (F16 && KV16) {
        Qh[i * QP + c] = (_Float16)((i0 + i < T) ? qv[q] : 0.f);       // natural order: K-index 16t + 8lh + e is channel 16t + 8lh + e
      } else if (F16) {
        const int h2 = c & 1, u = c >> 1;           // c = 2u + lh, u = 8t + e
        Qh[i * QP + (u >> 3) * 16 + h2 * 8 + (u & 7)] = (_Float16)((i0 + i < T) ? qv[q] : 0.f);
      } else {
        Qs[e] = qv[q];
      }
    }
  }
  // (the 2W+1 relative-key logit rows of the tile's queries are staged once: read from global inside the softmax loop they were a
  // dependent load per in-band element — up to 9 serial round trips in the wave that owns the diagonal tile, which every other
  // wave of the workgroup then waited for at the merge barrier: tools/timeline.py, 12-16k cycles of a 62k-cycle workgroup)
  __syncthreads();
  if (A.dbg) ts1 = __builtin_amdgcn_s_memtime();
  const float mi = iok ? mi_raw : 0.f;

  f32x16 O[DT];
#pragma unroll
  for (int m = 0; m < DT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[m][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

#pragma unroll 1
  for (int kt = wid; kt < ntiles; kt += (ONE ? (1 << 20) : NW)) {    // ONE: the body runs at most once (ntiles <= NW)
    const int j0 = (kt0 + kt) * AK;
    if (!ONE && kt >= NW) issue_k(j0);
    // ---- S^T = K^T Q   (rows beyond T hold garbage: replaced below, never accumulated)
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
    if (F16) {
#pragma unroll
      for (int t = 0; t < D / 16; ++t) {
        f16x8 ka;
        if constexpr (KV16) {
          ka = __builtin_bit_cast(f16x8, kq[t]);
        } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) ka[e] = (_Float16)kreg[8 * t + e];
        }
        const f16x8 qf = *reinterpret_cast<const f16x8*>(Qh + l31 * QP + t * 16 + lh * 8);
        S = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka, qf, S, 0, 0, 0);
      }
    } else {
    // the Q operand streams from LDS a few K-steps ahead of the MFMAs (pinned: hoisting all D/2 reads costs registers)
    float qb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qb[s] = Qs[(2 * s + lh) * AQ + l31];
#pragma unroll
    for (int s = 0; s < D / 2; ++s) {
      const float qcur = qb[s & 3];
      if (s + 4 < D / 2) qb[s & 3] = Qs[(2 * (s + 4) + lh) * AQ + l31];
      S = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[s], qcur, S, 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    }
    __builtin_amdgcn_sched_barrier(0);             // keep the V loads BEHIND the S MFMAs (K registers are free again)
    issue_v(j0);                                   // in flight under the softmax
    __builtin_amdgcn_sched_barrier(0);
    // ---- relative logits, mask, band capture, online softmax (lane = query column, regs = key rows)
    const bool near_band = (j0 - i0 <= AQ - 1 + W) && (i0 - j0 <= AK - 1 + W);     // wave-uniform
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jr = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int j = j0 + jr;
      float sv = S[r];
      const float mj = __shfl(mkey, jr);                       // key mask of row jr (lane jr loaded key j0+jr)
      if (near_band) {
        const int rel = j - iq;
        const bool inband = (rel >= -W) && (rel <= W) && iok && j < T;
        if (inband) sv += Qe[(rel + W) * AQ + l31];
        if (!(mi != 0.f && mj != 0.f)) sv = -1e4f;              // masked_fill(mask == 0, -1e4), attentions.py:297
        if (inband) Sb[(rel + W) * AQ + l31] = sv;
      } else {
        if (!(mi != 0.f && mj != 0.f)) sv = -1e4f;
      }
      if (j >= T) sv = -INFINITY;                               // key does not exist
      if (!iok) sv = (j < T) ? 0.f : -INFINITY;                 // dead query column: keep it finite
      S[r] = sv;
      tmax = fmaxf(tmax, sv);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __expf(m_run - m_new);                  // first tile: exp(-inf) = 0
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __expf(S[r] - m_new);
      psum += p;
      S[r] = p;
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (!ONE && kt >= NW) {
#pragma unroll
      for (int m = 0; m < DT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[m][r] *= alpha;
    }
    // ---- O^T += V P^T, K-steps in D-layout row order
    if (F16) {
      f16x8 pf[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[s2][e] = (_Float16)S[8 * s2 + e];
#pragma unroll
      for (int m = 0; m < DT; ++m) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int ja = j0 + 16 * s2 + 4 * lh, jb2 = ja + 8;
          if constexpr (KV16) {
            f16x8 vf = __builtin_bit_cast(f16x8, u32x4{vq[m][s2][0].x, vq[m][s2][0].y, vq[m][s2][1].x, vq[m][s2][1].y});
            if (jb2 + 3 >= T) {                               // tile tail: keys that do not exist contribute exactly 0 (their slots hold garbage)
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if ((e < 4 ? ja + e : jb2 + e - 4) >= T) vf[e] = (_Float16)0.f;
            }
            O[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s2], O[m], 0, 0, 0);
            continue;
          }
          f32x4 va = vreg[m][2 * s2], vb = vreg[m][2 * s2 + 1];
          if (jb2 + 3 >= T) {                                 // tile tail: keys that do not exist contribute exactly 0
            va.x = ja + 0 < T ? va.x : 0.f; va.y = ja + 1 < T ? va.y : 0.f; va.z = ja + 2 < T ? va.z : 0.f; va.w = ja + 3 < T ? va.w : 0.f;
            vb.x = jb2 + 0 < T ? vb.x : 0.f; vb.y = jb2 + 1 < T ? vb.y : 0.f; vb.z = jb2 + 2 < T ? vb.z : 0.f; vb.w = jb2 + 3 < T ? vb.w : 0.f;
          }
          f16x8 vf;
          vf[0] = (_Float16)va.x; vf[1] = (_Float16)va.y; vf[2] = (_Float16)va.z; vf[3] = (_Float16)va.w;
          vf[4] = (_Float16)vb.x; vf[5] = (_Float16)vb.y; vf[6] = (_Float16)vb.z; vf[7] = (_Float16)vb.w;
          O[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s2], O[m], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
    for (int m = 0; m < DT; ++m) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 v4 = vreg[m][g4];
        const int jb = j0 + 8 * g4 + 4 * lh;
        if (jb + 3 >= T) {                                    // tile tail: keys that do not exist contribute exactly 0
          v4.x = jb + 0 < T ? v4.x : 0.f; v4.y = jb + 1 < T ? v4.y : 0.f;
          v4.z = jb + 2 < T ? v4.z : 0.f; v4.w = jb + 3 < T ? v4.w : 0.f;
        }
        O[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(v4.x, S[4 * g4 + 0], O[m], 0, 0, 0);
        O[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(v4.y, S[4 * g4 + 1], O[m], 0, 0, 0);
        O[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(v4.z, S[4 * g4 + 2], O[m], 0, 0, 0);
        O[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(v4.w, S[4 * g4 + 3], O[m], 0, 0, 0);
      }
    }
    }
  }

  if (A.dbg) ts2 = __builtin_amdgcn_s_memtime();
  // fused conv_o: this wave's first 32-row tile of W_o (12 float4 per lane at D = 96) goes in flight NOW — the K / V registers are
  // dead — and lands under the merge below; loaded where it is used it was an exposed L2 round trip after the last barrier
  // (tools/timeline.py: normalise + conv_o + store 16-21k cycles of a 46-59k-cycle workgroup)
  f32x4 wr0[D / 8];
  const bool pre_w = A.wo != nullptr && wid * 32 < A.Co;
  if (pre_w) {
    const f32x4* wp = reinterpret_cast<const f32x4*>(A.wo) + ((int64_t)wid * A.wo_groups + h * (D / 8)) * 64 + lh * 32 + l31;
#pragma unroll
    for (int g = 0; g < D / 8; ++g) wr0[g] = wp[g * 64];
  }
  // ---- merge the waves' partials: (max, sum) first, then the rescaled O tiles through ANS slots in fixed order
  if (lh == 0) { Mw[wid * AQ + l31] = m_run; Lw[wid * AQ + l31] = l_run; }
  __syncthreads();
  float m_tot = -INFINITY;
#pragma unroll
  for (int w = 0; w < NW; ++w) m_tot = fmaxf(m_tot, Mw[w * AQ + l31]);
  float l_tot = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const float mw = Mw[w * AQ + l31];
    l_tot += (mw == -INFINITY) ? 0.f : Lw[w * AQ + l31] * __expf(mw - m_tot);
  }
  const float il = 1.0f / l_tot;                    // per lane: query l31 (== tid & 31 below since AQ == 32)
  const float fac = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_tot);
  // band logits -> band probabilities p[i][i+r] (each (r, i) once)
  // (element e = tid + n NT belongs to query e & 31 == tid & 31 == l31 — NT is a multiple of 64 — so the query's totals are the m_tot / l_tot this
  // lane has just formed: re-deriving them per element was 16 LDS reads and 8 exponentials each)
  for (int e = tid; e < NR * AQ; e += NT) {
    const int r = e >> 5;
    const int j = iq + r - W;
    // key split: a band key outside [kt0, kt1) belongs to another workgroup (its Sb slot was never written here)
    const bool mine = j >= kt0 * AK && j < kt1 * AK;
    Sb[e] = (j >= 0 && j < T && iok && mine) ? __expf(Sb[e] - m_tot) / l_tot : 0.f;
  }
  if (KS > 1 && tid < AQ && i0 + tid < T) {
    // (m_r, l_r) of this key range for the merge in the LayerNorm: [b][h][kr][2][T]
    float* ml = A.ml_out + ((((int64_t)b * A.H + h) * KS + kr) * 2) * T + i0 + tid;
    ml[0] = m_tot;                                 // tid < 32: lane l31 == tid, so m_tot / l_tot are query tid's
    ml[T] = l_tot;
  }
  if (A.dbg) ts3 = __builtin_amdgcn_s_memtime();
  constexpr int ROUNDS = (NW + ANS - 1) / ANS;
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) {
    if (wid / ANS == rd) {
      float* slot = Os + (wid % ANS) * (D * AQ);
#pragma unroll
      for (int m = 0; m < DT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const float val = O[m][r] * fac;
          if (rd == 0) slot[c * AQ + l31] = val; else slot[c * AQ + l31] += val;
        }
    }
    __syncthreads();
  }
  constexpr int NSLOT = NW < ANS ? NW : ANS;
  if (A.dbg) ts4 = __builtin_amdgcn_s_memtime();

  // ---- normalise, add relative-value term, store (coalesced over queries)
  float* op = A.out + (int64_t)b * HD * T + (int64_t)(h * D) * T;
  const int i = tid & 31, ig = i0 + i;
  if (A.wo) {
    // Fused output projection (conv_o, attentions.py:269): the head's normalised [D][32 queries] tile goes to LDS (the query tile's
    // region: every wave is past the main loop) and the workgroup multiplies it by ITS K-slice of W_o — columns [hD, hD + D) of
    // the fragment-ordered 1x1 weight — on the fp32 matrix core: one 32-row tile per wave, D/2 MFMAs.  Head h writes partial slab
    // h of the [B][C_out][T] output (head 0 adds the bias and the residual); the LayerNorm that follows sums the H slabs exactly as
    // it sums split-K slabs.  Removes the conv_o launch (and its HBM round trip of the attention output) from every layer.
    float* Of = Qs;
    // The band term Σ_r p[i][i+r] * Ev[r][c]: this thread's NR probabilities are the same for every channel it handles — read ONCE
    // into registers — and the channel loop is fully unrolled with a compile-time band bound (9 taps for the model's window 4, 17 for
    // the kernel's maximum), unused taps contributing an exact 0.  As `for (r < NR)` with a runtime NR inside a runtime channel loop
    // it was 6 x 9 dependent LDS round trips (tools/timeline.py: normalise + conv_o + store 14k of a 36k-cycle workgroup).
    auto normalise = [&](auto nrm_c) __attribute__((always_inline)) {
      constexpr int NRM = decltype(nrm_c)::value;
      float sbv[NRM];
#pragma unroll
      for (int r = 0; r < NRM; ++r) {
        const float pv = Sb[(r < NR ? r : 0) * AQ + i];
        sbv[r] = r < NR ? pv : 0.f;
      }
      const int c0 = tid >> 5;
#pragma unroll 1
      for (int it = 0; it < (D + 2 * NW - 1) / (2 * NW); ++it) {   // rolled: unrolled it keeps 6 x 21 loads live (222 registers, half the occupancy)
        const int c = c0 + it * 2 * NW;
        const int cc = c < D ? c : D - 1;
        float o = 0.f;
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) o += Os[sl * (D * AQ) + cc * AQ + i];
        o *= il;
#pragma unroll
        for (int r = 0; r < NRM; ++r) o += sbv[r] * Ev[(r < NR ? r : 0) * D + cc];
        if (c < D) Of[c * AQ + i] = ig < T ? o : 0.f;
      }
    };
    if (NR <= 9) normalise(std::integral_constant<int, 9>{});
    else normalise(std::integral_constant<int, 2 * AMAXW + 1>{});
    const int Co = A.Co, G = A.wo_groups;
    const float* const wo = A.wo;
    const float* const bo = h == 0 ? A.bo : nullptr;
    const float* const resp = h == 0 && A.res ? A.res + (int64_t)b * Co * T : nullptr;
    float* const ob = A.o_out + (int64_t)(h * KS + kr) * A.o_slab_stride + (int64_t)b * Co * T;
    __syncthreads();
    for (int mt = wid; mt * 32 < Co; mt += NW) {
      f32x4 wr[D / 8];
      if (mt == wid) {                               // prefetched above
#pragma unroll
        for (int g = 0; g < D / 8; ++g) wr[g] = wr0[g];
      } else {
        const f32x4* wp = reinterpret_cast<const f32x4*>(wo) + ((int64_t)mt * G + h * (D / 8)) * 64 + lh * 32 + l31;
#pragma unroll
        for (int g = 0; g < D / 8; ++g) wr[g] = wp[g * 64];
      }
      // epilogue operands in flight under the MFMAs
      float bs[16], rv[16];
      const int row0 = mt * 32 + 4 * lh;
      const int igc = iok ? iq : T - 1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = row0 + (r & 3) + 8 * (r >> 2);
        row = row < Co ? row : Co - 1;
        bs[r] = bo ? bo[row] : 0.f;
        rv[r] = resp ? resp[(int64_t)row * T + igc] : 0.f;
      }
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int g = 0; g < D / 8; ++g) {
        const float* ob4 = Of + (8 * g + lh) * AQ + l31;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[g].x, ob4[0 * AQ], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[g].y, ob4[2 * AQ], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[g].z, ob4[4 * AQ], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[g].w, ob4[6 * AQ], acc, 0, 0, 0);
      }
      if (iok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + (r & 3) + 8 * (r >> 2);
          if (row < Co) ob[(int64_t)row * T + iq] = (acc[r] + bs[r]) + rv[r];
        }
      }
    }
  } else if (ig < T) {
    // the plain output [B][H D][T] (the fp16 flow at batch >= 8, BERT): the band term as in `normalise` above — the thread's NR probabilities
    // once into registers, compile-time band bound (as a runtime `for (r < NR)` inside the channel loop it was 6 x 9 dependent LDS round
    // trips: 11.4k of a 31k-tick workgroup at B = 32, profiles/r05_timeline_c3_f16_convs.txt)
    auto finish = [&](auto nrm_c) __attribute__((always_inline)) {
      constexpr int NRM = decltype(nrm_c)::value;
      float sbv[NRM];
#pragma unroll
      for (int r = 0; r < NRM; ++r) {
        const float pv = Sb[(r < NR ? r : 0) * AQ + i];
        sbv[r] = r < NR ? pv : 0.f;
      }
#pragma unroll 2
      for (int c = tid >> 5; c < D; c += 2 * NW) {
        float o = 0.f;
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) o += Os[sl * (D * AQ) + c * AQ + i];
        o *= il;
#pragma unroll
        for (int r = 0; r < NRM; ++r) o += sbv[r] * Ev[(r < NR ? r : 0) * D + c];
        op[(int64_t)c * T + ig] = o;
      }
    };
    if (NR <= 1) finish(std::integral_constant<int, 1>{});
    else if (NR <= 9) finish(std::integral_constant<int, 9>{});
    else finish(std::integral_constant<int, 2 * AMAXW + 1>{});
  }
  if (A.dbg && tid == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long* d = A.dbg + 8ull * (((unsigned long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
    d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = __builtin_amdgcn_s_memtime();
    d[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    d[5] = ts3; d[6] = ts4; d[7] = 1;
  }
}

static size_t attn_lds_bytes(int D, int NW) {
  return sizeof(float) * (size_t)(ANS * D * AQ + D * AQ + (2 * AMAXW + 1) * D + 2 * (2 * AMAXW + 1) * AQ + 2 * NW * AQ);
}

template <int DT, int NW, bool F16, bool ONE, bool KV16 = false>
static int launch_attn_variant2(hipStream_t stream, const AttnArgs& a, dim3 grid) {
  const size_t lds = attn_lds_bytes(32 * DT, NW);
  auto kern = attention_kernel<DT, NW, F16, ONE, KV16>;
  ensure_dyn_lds((const void*)kern, lds);
  AttnArgs at = a;
  if (a.xcd_b) {
    at.xcd_gx = (int)grid.x; at.xcd_per = (int)(grid.x * grid.y);
    grid = dim3(xcd_grid(a.B, at.xcd_per), 1, 1);
  }
  at.dbg = timeline_slice(grid.x, grid.y, grid.z, 77000 + 10 * DT + NW, 0, a.D, a.T);    // tile id 77xxx: attention
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, stream, at);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int DT, int NW, bool ONE>
static int launch_attn_variant(hipStream_t stream, const AttnArgs& a, dim3 grid) {
  if (a.f16 && a.kh && a.vh) return launch_attn_variant2<DT, NW, true, ONE, true>(stream, a, grid);
  return a.f16 ? launch_attn_variant2<DT, NW, true, ONE>(stream, a, grid) : launch_attn_variant2<DT, NW, false, ONE>(stream, a, grid);
}

template <int DT>
static int launch_attn_d(hipStream_t stream, const AttnArgs& a, dim3 grid, int ntiles) {
  // ntiles = key tiles per workgroup (of its key range when the keys are split)
  // Up to 8 key tiles: one per wave, no loop (T = 128: 19.4 -> 16.9 us).  More: 8 waves round-robin.  Measured at T_y = 384
  // (12 tiles) and NOT kept: 12 waves with one tile each (53 spilled registers: 27 -> 41 us) and 6 waves with two tiles each
  // (balanced, but 27 -> 31 us: the merge wait that tools/timeline.py shows is not the 1-vs-2-tile imbalance).
  // fused conv_o: its C_out / 32 row tiles are dealt to the waves — with more than 4 of them (hidden 192: 6) eight waves finish them
  // in one round even if only 4 have a key tile
  if (ntiles <= 4 && !(a.wo && a.Co > 128)) return launch_attn_variant<DT, 4, true>(stream, a, grid);
  if (ntiles <= 8) return launch_attn_variant<DT, 8, true>(stream, a, grid);
  // head dim 128 with more than 8 key tiles: four waves walking the tiles (one wave per SIMD: 512 registers each) — the eight-wave loop form
  // spilled 29 registers there (K tile 64 + V 64 + O 64 + S 16 per wave against 256), and a scratch segment is not allowed on any path
  if constexpr (DT == 4) return launch_attn_variant<DT, 4, false>(stream, a, grid);
  else return launch_attn_variant<DT, 8, false>(stream, a, grid);
}

int launch_attention(hipStream_t stream, const AttnArgs& a) {
  if (a.W > AMAXW || a.W < 0 || a.T < 1 || a.B < 1 || a.H < 1 || a.ld % 32 || a.ld < a.T) return -1;
  if ((a.kh != nullptr) != (a.vh != nullptr) || (a.kh && !a.f16)) return -1;                  // fp16 K / V: both, and only for the fp16 products
  if (a.wo && (a.f16 || !a.o_out || a.Co < 1 || a.wo_groups * 8 < a.H * a.D)) return -1;   // fused conv_o: fp32 form only
  const int ks = a.ksplit > 1 ? a.ksplit : 1;
  const int ntiles_all = (a.T + AK - 1) / AK;
  if (ks > 1 && (!a.wo || !a.ml_out || a.bo || a.res || ks > ntiles_all)) return -1;        // key split: partial slabs merged by the LayerNorm
  dim3 grid(((a.T + AQ - 1) / AQ) * ks, a.H, a.B);
  const int ntiles = (ntiles_all + ks - 1) / ks;                                            // the largest key range
  switch (a.D) {
    case 32: return launch_attn_d<1>(stream, a, grid, ntiles);
    case 64: return launch_attn_d<2>(stream, a, grid, ntiles);
    case 96: return launch_attn_d<3>(stream, a, grid, ntiles);
    case 128: return launch_attn_d<4>(stream, a, grid, ntiles);
    default: return -2;   // head dim must be a multiple of 32 up to 128
  }
}

// Key ranges per (head, query tile).  Splitting pays only when a wave would otherwise walk several key tiles in turn (more tiles
// than the 8 waves of a workgroup) and the launch is far from filling the chip; the target is one key tile per SIMD (4 per
// workgroup).  H * ks partial slabs must be a count the slab-summing LayerNorm has (2, 4 or 8) and fit `max_slabs`.
int attention_pick_ksplit(int B, int H, int T, int max_slabs) {
  const int ntiles = (T + AK - 1) / AK, qtiles = (T + AQ - 1) / AQ;
  if (ntiles <= 8 || H < 1 || (H & (H - 1))) return 1;
  int ks = 1;
  while (ks * 4 < ntiles && H * ks * 2 <= max_slabs && H * ks * 2 <= 8 && (long)B * H * qtiles * ks * 2 <= 512) ks *= 2;
  return ks;
}

double attention_flops(const AttnArgs& a) {
  // QK^T + PV: 4*T*T*D per head (the 2W+1 band dot products per query ride on the q/k/v projection)
  return (double)a.B * a.H * (4.0 * a.T * (double)a.T * a.D);
}

}  // namespace bv2

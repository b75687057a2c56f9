// deberta_attn.hip — DeBERTa-v2's DisentangledSelfAttention (transformers models/deberta_v2/modeling_deberta_v2.py; reference call
// sites text/japanese_bert.py:34-43, text/english_bert_mock.py:30-41), flash-style on the fp32 matrix core:
//
//     score[i][j] = ( Q_i.K_j  +  Q_i.PK[t(i-j)]  +  K_j.PQ[t(i-j)] ) / sqrt(3 d),      softmax over j,   out_i = sum_j p[i][j] V_j
//
// PK / PQ = key_proj / query_proj(LayerNorm(rel_embeddings)) are functions of the WEIGHTS only (share_att_key): the host packs them
// once per layer, transposed per head to [d][2 span] so that a run of relative indices is a contiguous row read; t(r) =
// clamp(bucket(r) + span) is the log-bucket table of make_log_bucket_position, also packed (both relative terms use the same index:
// the bucket function is odd, so HF's p2c index clamp(-bucket(j-i) + span) is t(i-j)).  1/sqrt(3d) is folded into the query rows of
// the fused q/k/v projection and into PQ.
//
// Mapping = attention.hip's: one workgroup = 32 queries of one (batch, head), 4 waves, each wave takes 32-key tiles round-robin with
// a running (max, sum) and they merge through LDS.  Per (query tile, key tile) pair the relative indices span at most 63 consecutive
// table rows (t is monotone with slope <= 1), so the two relative terms are two small GEMMs on the matrix core —
//     Tc^T [64 rows x 32 queries] = PK[lo .. lo+63] . Q_tile         (A = PK^T rows from global / L2, B = the query tile in LDS)
//     Tp^T [64 rows x 32 keys]    = PQ[lo .. lo+63] . K_tile         (B = the K registers: the S^T A-fragment IS the B-fragment)
// written to a per-wave LDS table and gathered per score element.  S^T / softmax / PV exactly as in attention.hip.
#include <hip/hip_runtime.h>
#include <math.h>
#include "../bv2_kernels.h"

namespace bv2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int DQ = 32, DK = 32, DNW = 4, DNS = 4, DTP = 65;     // queries / keys per tile, waves, merge slots, table pitch
__device__ __forceinline__ float dld(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ f32x4 dld4(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byte_off);
}
}  // namespace

template <int DT>   // head dim D = 32 * DT
__global__ void __launch_bounds__(64 * DNW) deberta_attn_kernel(const DebertaAttnArgs A) {
  constexpr int D = 32 * DT, NT = 64 * DNW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * DQ;
  const int T = A.T, ld = A.ld, HD = A.H * D, P = A.P, span2 = 2 * A.span;

  float* Os = smem;                                 // [DNS][D][DQ] partial outputs
  float* Qs = Os + DNS * D * DQ;                    // [D][DQ] query tile (pre-scaled by the projection)
  float* Mw = Qs + D * DQ;                          // [DNW][DQ]
  float* Lw = Mw + DNW * DQ;                        // [DNW][DQ]
  float* Tb = Lw + DNW * DQ;                        // [2P - 1] relative index table t(r), r = -(P-1) .. P-1
  float* Tw = Tb + ((2 * P - 1 + 31) & ~31) + (size_t)wid * (2 * DQ * DTP);   // this wave's [2][32][DTP]: Tc (by query), Tp (by key)

  const float* base = A.qkv + (int64_t)b * 3 * HD * ld;
  const float* qp = base + (int64_t)(h * D) * ld;
  const float* kp = qp + (int64_t)HD * ld;
  const float* vp = kp + (int64_t)HD * ld;
  const float* mp = A.mask + (int64_t)b * T;
  const float* pk = A.pk + (int64_t)h * D * span2;  // [D][2 span]
  const float* pq = A.pq + (int64_t)h * D * span2;

  const int iq = i0 + l31;
  const bool iok = iq < T;
  const int ntiles = (T + DK - 1) / DK;

  float kreg[D / 2];
  f32x4 vreg[DT][4];
  float mkey = 0.f;
  const unsigned koff = 4u * (unsigned)(lh * ld + l31);
  const unsigned voff = 4u * (unsigned)(l31 * ld + 4 * lh);
  auto issue_k = [&](int j0) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < D / 2; ++s) kreg[s] = dld(kp, koff + 4u * (unsigned)(j0 + 2 * s * ld));
    const int jm = j0 + l31;
    mkey = mp[jm < T ? jm : T - 1];
  };
  auto issue_v = [&](int j0) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < DT; ++m)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) vreg[m][g4] = dld4(vp, voff + 4u * (unsigned)((m * 32) * ld + j0 + 8 * g4));
  };
  if (wid < ntiles) issue_k(wid * DK);
  const float mi = iok ? mp[iq] : 0.f;
  {
    constexpr int QPT = (D * DQ + NT - 1) / NT;
    float qv[QPT];
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
      int e = tid + q * NT;
      e = e < D * DQ ? e : D * DQ - 1;
      const int col = i0 + (e & 31);
      qv[q] = qp[(e >> 5) * ld + (col < ld ? col : ld - 1)];
    }
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
      const int e = tid + q * NT;
      if (e < D * DQ) Qs[e] = (i0 + (e & 31) < T) ? qv[q] : 0.f;
    }
  }
  for (int e = tid; e < 2 * P - 1; e += NT) Tb[e] = A.tab[e];
  __syncthreads();

  f32x16 O[DT];
#pragma unroll
  for (int m = 0; m < DT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[m][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  float* Tc = Tw;                                   // [32 queries][DTP]
  float* Tp = Tw + DQ * DTP;                        // [32 keys][DTP]

#pragma unroll 1
  for (int kt = wid; kt < ntiles; kt += DNW) {
    const int j0 = kt * DK;
    if (kt >= DNW) issue_k(j0);
    // ---- S^T = K^T Q
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
    for (int s = 0; s < D / 2; ++s) S = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[s], Qs[(2 * s + lh) * DQ + l31], S, 0, 0, 0);
    // ---- relative tables of this tile pair: rows lo .. lo + 63 of PK / PQ (t is monotone with slope <= 1: the pair needs <= 63 rows)
    int r_lo = i0 - (j0 + DK - 1) + P - 1, r_hi = i0 + DQ - 1 - j0 + P - 1;
    r_lo = r_lo < 0 ? 0 : r_lo;
    r_hi = r_hi > 2 * P - 2 ? 2 * P - 2 : r_hi;
    const int lo = (int)Tb[r_lo];
    const int nblk = ((int)Tb[r_hi] - lo) / 32 + 1;  // 1 or 2 blocks of 32 table rows (wave-uniform: r_lo / r_hi are)
    for (int blk = 0; blk < nblk; ++blk) {
      int row = lo + 32 * blk + l31;
      row = row < span2 ? row : span2 - 1;
      f32x16 tc, tp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { tc[r] = 0.f; tp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < D / 2; ++s) {
        const float ak = pk[(2 * s + lh) * span2 + row];
        const float aq = pq[(2 * s + lh) * span2 + row];
        tc = __builtin_amdgcn_mfma_f32_32x32x2f32(ak, Qs[(2 * s + lh) * DQ + l31], tc, 0, 0, 0);   // [table row][query]
        tp = __builtin_amdgcn_mfma_f32_32x32x2f32(aq, kreg[s], tp, 0, 0, 0);                       // [table row][key]
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tr = 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * lh;   // table row of this register; the lane's column is l31
        Tc[l31 * DTP + tr] = tc[r];
        Tp[l31 * DTP + tr] = tp[r];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    issue_v(j0);                                   // in flight under the gather + softmax
    __builtin_amdgcn_sched_barrier(0);
    // ---- add the two relative terms, mask, online softmax (lane = query column, regs = key rows)
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jr = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int j = j0 + jr;
      int ri = iq - j + P - 1;
      ri = ri < 0 ? 0 : (ri > 2 * P - 2 ? 2 * P - 2 : ri);
      int t = (int)Tb[ri] - lo;
      t = t < 0 ? 0 : (t > 63 ? 63 : t);             // only reachable for non-existent queries / keys (dropped below)
      float sv = S[r] + (Tc[l31 * DTP + t] + Tp[jr * DTP + t]);
      const float mj = __shfl(mkey, jr);
      if (!(mi != 0.f && mj != 0.f)) sv = -1e4f;     // masked_fill(~mask, finfo.min): probability exactly 0 next to any valid key
      if (j >= T) sv = -INFINITY;
      if (!iok) sv = (j < T) ? 0.f : -INFINITY;
      S[r] = sv;
      tmax = fmaxf(tmax, sv);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __expf(m_run - m_new);
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
    if (kt >= DNW) {
#pragma unroll
      for (int m = 0; m < DT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[m][r] *= alpha;
    }
    // ---- O^T += V P^T
#pragma unroll
    for (int m = 0; m < DT; ++m) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 v4 = vreg[m][g4];
        const int jb = j0 + 8 * g4 + 4 * lh;
        if (jb + 3 >= T) {
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

  // ---- merge the waves' partials
  if (lh == 0) { Mw[wid * DQ + l31] = m_run; Lw[wid * DQ + l31] = l_run; }
  __syncthreads();
  float m_tot = -INFINITY;
#pragma unroll
  for (int w = 0; w < DNW; ++w) m_tot = fmaxf(m_tot, Mw[w * DQ + l31]);
  float l_tot = 0.f;
#pragma unroll
  for (int w = 0; w < DNW; ++w) {
    const float mw = Mw[w * DQ + l31];
    l_tot += (mw == -INFINITY) ? 0.f : Lw[w * DQ + l31] * __expf(mw - m_tot);
  }
  const float il = 1.0f / l_tot;
  const float fac = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_tot);
  {
    float* slot = Os + wid * (D * DQ);              // DNW == DNS: one slot per wave, one round
#pragma unroll
    for (int m = 0; m < DT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) slot[(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * DQ + l31] = O[m][r] * fac;
  }
  __syncthreads();
  float* op = A.out + (int64_t)b * HD * T + (int64_t)(h * D) * T;
  const int i = tid & 31, ig = i0 + i;
  if (ig < T) {
    for (int c = tid >> 5; c < D; c += 2 * DNW) {
      float o = 0.f;
#pragma unroll
      for (int sl = 0; sl < DNS; ++sl) o += Os[sl * (D * DQ) + c * DQ + i];
      op[(int64_t)c * T + ig] = o * il;             // il is per lane = query tid & 31 (DQ == 32)
    }
  }
}

int launch_deberta_attn(hipStream_t stream, const DebertaAttnArgs& a) {
  if (a.T < 1 || a.B < 1 || a.H < 1 || a.ld % 32 || a.ld < a.T || a.P < a.T || a.span < 1 || !a.qkv || !a.pk || !a.pq || !a.tab || !a.mask)
    return -1;
  static_assert(DNW == DNS, "one merge slot per wave");
  const size_t lds = sizeof(float) * ((size_t)DNS * a.D * DQ + (size_t)a.D * DQ + 2 * DNW * DQ + ((2 * a.P - 1 + 31) & ~31) +
                                      (size_t)DNW * 2 * DQ * DTP);
  if (lds > 160 * 1024) return -2;
  dim3 grid((a.T + DQ - 1) / DQ, a.H, a.B);
  auto go = [&](auto kern) {
    ensure_dyn_lds((const void*)kern, lds);
    hipLaunchKernelGGL(kern, grid, dim3(64 * DNW), lds, stream, a);
  };
  switch (a.D) {
    case 32: go(deberta_attn_kernel<1>); break;
    case 64: go(deberta_attn_kernel<2>); break;
    case 96: go(deberta_attn_kernel<3>); break;
    case 128: go(deberta_attn_kernel<4>); break;
    default: return -2;
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace bv2

// conv_mfma.hip — conv1d (any k / dilation / padding, incl. 1x1 and the polyphase branches of ConvTranspose1d) as an
// implicit GEMM on the gfx950 fp32 matrix core (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains at the 157 TF vector rate).
//
//   GEMM view:  M = C_out (rows, from the packed weight),  N = time (columns, contiguous in HBM),  K = (tap j, C_in).
//   A[m][kk] = Wp[j][ci][co]  -> LDS tile Ws[CK][BM]  (co fastest: lane l reads Ws[kk = l>>5][m = l&31], conflict free)
//   B[kk][n] = act(x)[ci][t0 + n - pad_left + j*dil] -> LDS tile Xs[CK][BN + (k-1)*dil] staged ONCE per C_in chunk and
//              re-used by all k taps (the pre-activation / input mask / 3-way mean is applied while staging, once per
//              element instead of once per use).  Lane l reads Xs[kk = l>>5][n = l&31 (+ tap shift)], conflict free.
//   D: lane holds column n = l&31, rows (r&3)+8(r>>2)+4(l>>5): each register stores as two 128-byte row segments.
//
// Replaces (reference): every Conv1d of modules.ResBlock1 (modules.py:296-309), Generator.conv_pre / ups
// (models.py:539-545), attentions.FFN (attentions.py:438-446), q/k/v/o and all 1x1 projections, DurationPredictor
// convs (models.py:285-299), WN in/res_skip layers (modules.py:192-210).
#include <hip/hip_runtime.h>
#include "../bv2_kernels.h"

namespace bv2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WM, int WN, int MI, int NI, int CK>
__global__ void __launch_bounds__(256) conv1d_mfma_kernel(const ConvLaunch L, const int mtiles) {
  constexpr int BM = WM * MI * 32;
  constexpr int BN = WN * NI * 32;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int W4 = CK * BM / 4;                 // float4 per weight tile
  constexpr int NW4 = (W4 + 255) / 256;           // float4 per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const ConvProb& P = L.p[blockIdx.z];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const int b = blockIdx.y / mtiles;
  const int m0 = (blockIdx.y - b * mtiles) * BM;
  const int t0 = blockIdx.x * BN;
  if (m0 >= P.cout_pad) return;                   // problems in one launch may have different C_out

  const int k = P.k, dil = P.dil;
  const int XW = BN + (k - 1) * dil;
  float* Ws = smem;                               // [2][CK][BM]
  float* Xs = smem + 2 * CK * BM;                 // [CK][XW]

  const int nchunks = P.cin_pad / CK;
  const int nsteps = nchunks * k;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  float4 wreg[NW4];

  auto load_w = [&](int step) {
    const int c = step / k, j = step - c * k;
    const float* src = P.w + ((int64_t)(j * P.cin_pad + c * CK)) * P.w_ld + m0;
#pragma unroll
    for (int q = 0; q < NW4; ++q) {
      const int idx = tid + q * 256;
      if (idx < W4) {
        const int row = idx / (BM / 4), c4 = idx % (BM / 4);
        wreg[q] = *reinterpret_cast<const float4*>(src + (int64_t)row * P.w_ld + c4 * 4);
      }
    }
  };
  auto store_w = [&](int buf) {
    float* dst = Ws + buf * CK * BM;
#pragma unroll
    for (int q = 0; q < NW4; ++q) {
      const int idx = tid + q * 256;
      if (idx < W4) *reinterpret_cast<float4*>(dst + idx * 4) = wreg[q];
    }
  };
  auto stage_x = [&](int c) {
    const int tbase = t0 - P.pad_left;
    for (int ci = wid; ci < CK; ci += 4) {
      const int cg = c * CK + ci;
      const bool cok = cg < P.cin;
      const int64_t roff = (int64_t)b * P.x_bstride + (int64_t)cg * P.x_rstride;
      for (int i = lane; i < XW; i += 64) {
        const int t = tbase + i;
        float v = 0.f;
        if (cok && t >= 0 && t < P.Lin) {
          v = P.x[0][roff + t];
          if (P.nsrc > 1) v += P.x[1][roff + t];
          if (P.nsrc > 2) v += P.x[2][roff + t];
          v *= P.in_scale;
          if (P.pre_act == PRE_LRELU) v = v > 0.f ? v : v * P.slope;
          if (P.in_mask) v *= P.in_mask[(int64_t)b * P.in_mask_bstride + t];
        }
        Xs[ci * XW + i] = v;
      }
    }
  };

  // prologue
  load_w(0);
  stage_x(0);
  store_w(0);
  __syncthreads();

  for (int step = 0; step < nsteps; ++step) {
    const int buf = step & 1;
    const int c = step / k, j = step - c * k;
    const bool more = (step + 1) < nsteps;
    if (more) load_w(step + 1);                   // global loads stay in flight under the MFMAs below

    const float* wsb = Ws + buf * CK * BM + lh * BM + wm * (MI * 32) + l31;
    const float* xsb = Xs + lh * XW + wn * (NI * 32) + l31 + j * dil;
#pragma unroll
    for (int s = 0; s < CK / 2; ++s) {
      float a[MI], bb[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = wsb[(2 * s) * BM + mi * 32];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bb[ni] = xsb[(2 * s) * XW + ni * 32];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], bb[ni], acc[mi][ni], 0, 0, 0);
    }

    if (more) {
      if (j == k - 1) {                           // next step starts a new C_in chunk: everyone is done reading Xs
        __syncthreads();
        stage_x(c + 1);
      }
      store_w(buf ^ 1);
    }
    __syncthreads();
  }

  // epilogue
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = t0 + wn * (NI * 32) + ni * 32 + l31;
      if (col >= L.L) continue;
      const float om = P.out_mask ? P.out_mask[(int64_t)b * P.out_mask_bstride + col] : 1.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (MI * 32) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= P.cout) continue;
        float v = acc[mi][ni][r];
        if (P.bias) v += P.bias[row];
        if (P.bias2) v += P.bias2[(int64_t)b * P.bias2_bstride + row];
        if (P.act == ACT_RELU) v = fmaxf(v, 0.f);
        if (P.mask_pre) v *= om;
        const int64_t oidx = (int64_t)row * P.out_rstride + (int64_t)col * P.out_tstride + P.out_toff;
        if (P.res_mode == RES_ADD) v += P.res[(int64_t)b * P.res_bstride + oidx];
        else if (P.res_mode == RES_RSUB) v = P.res[(int64_t)b * P.res_bstride + oidx] - v;
        if (P.mask_post) v *= om;
        P.out[(int64_t)b * P.out_bstride + oidx] = v;
      }
    }
  }
}

struct TileCfg { int id, bm, bn; const char* name; };
static const TileCfg kTiles[] = {
    // order = preference of the auto picker (largest first)
    {TILE_128x128, 128, 128, "conv1d_mfma<128x128>"}, {TILE_64x128, 64, 128, "conv1d_mfma<64x128>"},
    {TILE_32x256, 32, 256, "conv1d_mfma<32x256>"},    {TILE_64x64, 64, 64, "conv1d_mfma<64x64>"},
    {TILE_32x128, 32, 128, "conv1d_mfma<32x128>"},
};

template <int WM, int WN, int MI, int NI>
static int launch_variant(hipStream_t stream, const ConvLaunch& L, int ck, int max_cout_pad, int max_xw_extra) {
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
  const int mtiles = (max_cout_pad + BM - 1) / BM;
  dim3 grid((L.L + BN - 1) / BN, mtiles * L.B, L.nprob);
  const size_t lds = sizeof(float) * (size_t)(2 * ck * BM + ck * (BN + max_xw_extra));
  if (lds > 160 * 1024) return -2;
  if (ck == 32) {
    auto kern = conv1d_mfma_kernel<WM, WN, MI, NI, 32>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, L, mtiles);
  } else {
    auto kern = conv1d_mfma_kernel<WM, WN, MI, NI, 16>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, L, mtiles);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_conv1d(hipStream_t stream, const ConvLaunch& L, int tile, const char** variant_name) {
  if (L.nprob < 1 || L.nprob > BV2_MAX_PROBS || L.B < 1 || L.L < 1) return -1;
  int max_cout_pad = 0, max_extra = 0, ck = 32;
  for (int i = 0; i < L.nprob; ++i) {
    const ConvProb& p = L.p[i];
    if (p.cout_pad % 32 || p.cin_pad % 16 || p.k < 1 || p.dil < 1 || p.w_ld % 128 || p.w_ld < p.cout_pad) return -1;
    if (p.cout_pad > max_cout_pad) max_cout_pad = p.cout_pad;
    if ((p.k - 1) * p.dil > max_extra) max_extra = (p.k - 1) * p.dil;
    if (p.cin_pad % 32) ck = 16;
  }
  if (tile == TILE_AUTO) {
    // largest tile that still yields >= ~1 workgroup per CU (256 CUs); small problems fall to the smallest tiles
    const long target = 256;
    tile = TILE_32x128;
    for (const TileCfg& t : kTiles) {
      if (t.bm > max_cout_pad && t.bm != 32) continue;
      if (max_cout_pad % t.bm && t.bm != 32) continue;
      const long blocks = (long)((L.L + t.bn - 1) / t.bn) * ((max_cout_pad + t.bm - 1) / t.bm) * L.B * L.nprob;
      if (blocks >= target) { tile = t.id; break; }
    }
    if (tile == TILE_32x128 && max_cout_pad % 64 == 0 && L.L <= 64) tile = TILE_64x64;
  }
  for (const TileCfg& t : kTiles)
    if (t.id == tile && variant_name) *variant_name = t.name;
  switch (tile) {
    case TILE_128x128: return launch_variant<2, 2, 2, 2>(stream, L, ck, max_cout_pad, max_extra);
    case TILE_64x128:  return launch_variant<2, 2, 1, 2>(stream, L, ck, max_cout_pad, max_extra);
    case TILE_64x64:   return launch_variant<2, 2, 1, 1>(stream, L, ck, max_cout_pad, max_extra);
    case TILE_32x128:  return launch_variant<1, 4, 1, 1>(stream, L, ck, max_cout_pad, max_extra);
    case TILE_32x256:  return launch_variant<1, 4, 1, 2>(stream, L, ck, max_cout_pad, max_extra);
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

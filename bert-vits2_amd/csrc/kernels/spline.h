// spline.h — inverse piecewise rational-quadratic spline with linear tails (K = 10 bins), one element per call; always fp32.
// reference transforms.py:49-96 (tails) and :99-187 (inverse branch :160-173); op order kept (softmax, min-width affine,
// sequential cumsum, knots forced to +-tail, widths as knot differences, searchsorted with +1e-6 on the last knot).
// Shared by spline_kernel (misc.hip) and the ConvFlow epilogue of dds_layer_kernel (dds_fused.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace bv2 {

constexpr int SPK = 10;
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

__device__ __forceinline__ void spline_knots(const float* u, float lo, float hi, float minw, float wscale, float* cum /*K+1*/,
                                             float* wd /*K*/) {
  float mx = u[0];
#pragma unroll
  for (int i = 1; i < SPK; ++i) mx = fmaxf(mx, u[i]);
  float e[SPK], sum = 0.f;
#pragma unroll
  for (int i = 0; i < SPK; ++i) { e[i] = expf(u[i] - mx); sum += e[i]; }
  float c = 0.f;
  cum[0] = lo;
#pragma unroll
  for (int i = 0; i < SPK; ++i) {
    const float w = minw + wscale * (e[i] / sum);
    c += w;
    cum[i + 1] = (hi - lo) * c + lo;
  }
  cum[SPK] = hi;
#pragma unroll
  for (int i = 0; i < SPK; ++i) wd[i] = cum[i + 1] - cum[i];
}

// y -> x for one element: uw / uh = unnormalised widths / heights ALREADY divided by sqrt(filter_channels), ud[0] = ud[SPK] = cst
// (the padded derivative constant log(exp(1 - 1e-3) - 1), evaluated in double on the host), ud[1..SPK-1] the 9 raw derivatives
__device__ __forceinline__ float rq_spline_inverse_one(float y, const float* uw, const float* uh, const float* ud, float tail,
                                                       float wscale) {
  if (!(y >= -tail && y <= tail)) return y;
  float cw[SPK + 1], w[SPK], ch[SPK + 1], hh[SPK];
  spline_knots(uw, -tail, tail, 1e-3f, wscale, cw, w);
  spline_knots(uh, -tail, tail, 1e-3f, wscale, ch, hh);
  int bin = -1;
#pragma unroll
  for (int i = 0; i <= SPK; ++i) {
    const float kn = (i == SPK) ? ch[i] + 1e-6f : ch[i];
    bin += (y >= kn) ? 1 : 0;
  }
  bin = bin < 0 ? 0 : (bin > SPK - 1 ? SPK - 1 : bin);
  float icw = 0, iw = 0, ich = 0, ih = 0, d0 = 0, d1 = 0;
#pragma unroll
  for (int i = 0; i < SPK; ++i)
    if (i == bin) { icw = cw[i]; iw = w[i]; ich = ch[i]; ih = hh[i]; d0 = 1e-3f + softplus_f(ud[i]); d1 = 1e-3f + softplus_f(ud[i + 1]); }
  const float idl = ih / iw;
  const float tt = y - ich;
  const float s = d0 + d1 - 2.f * idl;
  const float a = tt * s + ih * (idl - d0);
  const float bq = ih * d0 - tt * s;
  const float cq = -idl * tt;
  const float disc = bq * bq - 4.f * a * cq;
  const float root = (2.f * cq) / (-bq - sqrtf(disc));
  return root * iw + icw;
}

}  // namespace bv2

/*
 * bv2_testing.h — kernel-level entry points of libbv2.so used ONLY by tests/ (unit parity of single kernels against
 * torch fp32 references).  Not part of the drop-in boundary; synchronous where noted.
 */
#ifndef BV2_TESTING_H
#define BV2_TESTING_H
#include <stdint.h>
#include "bv2.h"
#ifdef __cplusplus
extern "C" {
#endif

/* floats of device scratch bv2_test_conv1d needs for the packed weight + bias */
int64_t bv2_test_conv_pack_floats(int cin, int cout, int k);

/* One conv1d problem through the MFMA implicit-GEMM kernels with a forced variant (0 = auto, 1..5 LDS-tiled kernel tile
 * shapes, 6 = split-K kernel; see bv2_kernels.h TILE_*).  With tile 6 and ksplit > 1 the output is `ksplit` partial
 * slabs, `slab_stride` floats apart, whose SUM is the result (bias / residual ride on slab 0).  w_host [cout][cin][k] / bias_host [cout] are HOST pointers (packed + uploaded
 * synchronously into wpack_dev); every other pointer is DEVICE.  x [B][cin][L], out/res [B][cout][L*out_tstride],
 * masks [B][L].  pad_left < 0 means "same" padding ((k-1)/2*dil). */
int bv2_test_conv1d(void* stream, const float* x, const float* w_host, const float* bias_host, float* out, float* wpack_dev,
                    int B, int cin, int cout, int k, int dil, int pad_left, int L, int tile, float lrelu_slope, int relu,
                    const float* res, int res_mode, const float* in_mask, const float* out_mask, int mask_pre,
                    int mask_post, const float* bias2, int nsrc, const float* x1, const float* x2, float in_scale, int ksplit,
                    int64_t slab_stride);

/* tile 8 = the split-bf16 form of the LDS-tiled kernel with its own tile choice (kernels/conv_x6.hip; cin % 32 == 0, nsrc == 1),
 * 9 / 10 / 11 / 12 = its 128x64 / 128x64-with-loader-waves / 64x128 / 32x256 workgroup tiles forced. */

/* tile 14 = the two-plane fp16 form of the same kernel ("x3", bv2_kernels.h): tile 8's choice with the scaled fp16 planes, max |x|
 * reduced into a slot by a launch in front.  For cin % 32 == 0 and ksplit <= 1 EVERY tile also publishes max |out| (ConvProb::omax)
 * as fp32 bits in a slot (eight words, 32 floats apart: one per XCD — the value is their max) at float offset
 * bv2_test_x3_omax_off(cin, cout, k) of wpack_dev. */
int64_t bv2_test_x3_omax_off(int cin, int cout, int k);

/* v = h[0] + h[1] + h[2] exactly as bf16 bit patterns: the host-side split the packer applies to the x6 weight planes (host only) */
void bv2_test_x6_split(float v, uint16_t* h3);
/* where the packed blob holds x6 weight planes: float offset / float count of every region ([unit][plane 3][64 lanes][8] uint16); returns
 * the number of regions (host only; layout is known after bv2_create) */
int bv2_test_x6_regions(const bv2_handle* h, int64_t* off_floats, int64_t* n_floats, int max_regions);
/* ... and its scaled fp16 planes of the x3 form, in the same conv order: a region = 64 floats (the first = 1 / S_w) followed by
 * [unit][plane 2][64 lanes][8] fp16 — unit / lane / element order as in the x6 region of the same conv */
int bv2_test_x3_regions(const bv2_handle* h, int64_t* off_floats, int64_t* n_floats, int max_regions);

/* fused ResBlock1 pair (kernels/resblock_fused.hip): out = x + conv2(lrelu(conv1(lrelu(x), k, dil) + b1), k, 1) + b2 on
 * [B][C][L]; w*_host [C][C][k], b*_host [C] are HOST pointers; wpack_dev needs 2 * bv2_test_conv_pack_floats(C, C, k) floats */
int bv2_test_resblock_fused(void* stream, const float* x, float* out, const float* w1_host, const float* b1_host,
                            const float* w2_host, const float* b2_host, float* wpack_dev, int B, int C, int k, int dil, int L,
                            float slope);

/* channels-last bf16 conv (kernels/gen_bf16.hip): x* / out / res are DEVICE bf16 [B][L][C] tensors, w_host [cout][cin][k]
 * and bias_host [cout] HOST fp32 (rounded to bf16 / kept fp32 by the packer); wpack_dev needs
 * bv2_test_conv_cl_pack_bytes(cin, cout, k) bytes; the nsrc sources are averaged; pad_left < 0 means "same" padding */
int64_t bv2_test_conv_cl_pack_bytes(int cin, int cout, int k);
int bv2_test_conv_cl_bf16(void* stream, const void* x0, const void* x1, const void* x2, int nsrc, const float* w_host,
                          const float* bias_host, void* wpack_dev, void* out, const void* res, const float* bias2, int B, int cin,
                          int cout, int k, int dil, int pad_left, int L, int pre_lrelu, float slope);

/* a WHOLE ResBlock1 (nd (dilated conv, conv) pairs with their residuals, reference modules.py:296-309) of a narrow Generator stage in one
 * launch, bf16 channels-last: x / out DEVICE bf16 [B][L][C] (out != x), w_host [nd][2][C][C][k] and bias_host [nd][2][C] HOST fp32 (index
 * [d][0] = convs1[d] with dilation dil[d], [d][1] = convs2[d]); variant 0 = kernels/resblock_cl_bf16.hip (v_mfma_f32_32x32x16_bf16, C = 16 /
 * 32), 1 = kernels/resblock_c16_bf16.hip (v_mfma_f32_16x16x32_bf16, C = 16); lens: optional DEVICE int64 [B] valid rows per item;
 * wpack_dev needs bv2_test_resblock_cl_pack_bytes(C, k, nd) bytes.  Returns -2 for an unsupported shape. */
int64_t bv2_test_resblock_cl_pack_bytes(int C, int k, int nd);
int bv2_test_resblock_cl(void* stream, const void* x, void* out, const float* w_host, const float* bias_host, void* wpack_dev, int B,
                         int C, int k, const int* dil, int nd, int L, float slope, int variant, const int64_t* lens);

/* ONE (dilated conv, conv) pair of ResBlock1 with its residual in one launch, bf16 channels-last (kernels/respair_cl_bf16.hip): x / out
 * DEVICE bf16 [B][L][C] (out != x), C = 32 / 64 / 128 / 256; w_host [2][C][C][k] (index 0 = the dilated conv), bias_host [2][C] HOST fp32;
 * form 1 = 64-channel x 128-row wave tiles on the swizzled tile (C >= 64), 0 = 32-channel waves on the padded tile; lens optional DEVICE
 * int64 [B]; wpack_dev needs bv2_test_respair_cl_pack_bytes(C, k) bytes */
int64_t bv2_test_respair_cl_pack_bytes(int C, int k);
int bv2_test_respair_cl(void* stream, const void* x, void* out, const float* w_host, const float* bias_host, void* wpack_dev, int B,
                        int C, int k, int dil, int L, float slope, int form, const int64_t* lens);
/* the same pair in fp32 with both convs on the bf16 matrix core from exact three-way splits (kernels/respair_x6.hip): x / out DEVICE fp32
 * [B][C][L], C = 16 / 32 / 64 / 128; the packer splits w_host into its three bf16 planes (x6_split) */
int64_t bv2_test_respair_x6_pack_bytes(int C, int k);
int bv2_test_respair_x6(void* stream, const float* x, float* out, const float* w_host, const float* bias_host, void* wpack_dev, int B,
                        int C, int k, int dil, int L, float slope, const int64_t* lens);
/* the same pair on the two-plane fp16 form ("x3": scaled fp16 halves, three products, per-workgroup activation scales); same arguments */
int bv2_test_respair_x3(void* stream, const float* x, float* out, const float* w_host, const float* bias_host, void* wpack_dev, int B,
                        int C, int k, int dil, int L, float slope, const int64_t* lens);
/* the flow's coupling boundary in one launch (kernels/flow_boundary.hip): h = LayerNorm_C(sum of nslab slabs a [B][C][T]) * mask;
 * x1 [C/2 rows][T] (inside z, z_bstride floats per item) = (x1 - post(h) - post_b) * mask in place; pre_out [B][C][T] = (pre(x1) + pre_b) *
 * mask unless pre_w_host is NULL.  post_w_host [C/2][C], pre_w_host [C][C/2] HOST fp32; every other pointer DEVICE; wpack_dev needs
 * bv2_test_flow_boundary_pack_floats(C) floats.  C = 192 only (-2 otherwise). */
int64_t bv2_test_flow_boundary_pack_floats(int C);
int bv2_test_flow_boundary(void* stream, const float* a, int nslab, int64_t slab_stride, const float* gamma, const float* beta,
                           const float* mask, float* x1, int64_t z_bstride, const float* post_w_host, const float* post_b_host,
                           const float* pre_w_host, const float* pre_b_host, float* pre_out, float* wpack_dev, int B, int C, int T);

/* fp16 Encoder conv (kernels/enc_f16.hip).  in_ct: x is DEVICE fp32 [B][cin][L] (in_mask [B][L] optional) else fp16 [B][L][cin];
 * out_ct: out is DEVICE fp32 [B][cout][out_rstride] (res like out, res_mode 0/1/2 = none/add/rsub) else fp16 [B][L][cout];
 * w_host [cout][cin][k], bias_host [cout] HOST fp32; wpack_dev needs bv2_test_conv_cl_pack_bytes(cin, cout, k) bytes;
 * "same" padding; act 1 = ReLU; out_mask [B][L] applied before (mask_pre) and/or after (mask_post) the residual op */
int bv2_test_conv_f16(void* stream, const void* x, int in_ct, const float* in_mask, const float* w_host, const float* bias_host,
                      void* wpack_dev, void* out, int out_ct, const float* res, int res_mode, const float* out_mask, int mask_pre,
                      int mask_post, int act, int B, int cin, int cout, int k, int dil, int L, int out_rstride);

/* Decode one Generator conv of a packed HOST blob back to dense form (checks the bf16 packer on a CPU-only box):
 * kind 0 = dec.conv_pre, 1 = dec.ups[i] in its channels-last single-conv form (C_out' = u*C_out), 2 = resblock conv
 * rb[i][j][d][e], 3 = fp16 stream of a transformer-flow Encoder conv (coupling i in application order, layer j, d = 0 fused
 * q/k/v(+relative-key rows) / 1 conv_o / 2 FFN conv_1 / 3 FFN conv_2), 4 = rb[i][j][d][e] read back from the tap-major
 * whole-ResBlock stream (must equal kind 2), 5 = rb[i][j][d][e] (C = 16 stage) read back from the tap-pair stream of
 * kernels/resblock_c16_bf16.hip, bias from its own bias block (must equal kind 2).  dims = {cin, cout, k, pad_left}; w_out [cout][cin][k] (bf16 values widened to fp32) and bias_out [cout]
 * may be NULL to query dims only.  Returns 0, or a negative status. */
int bv2_test_dump_cl_conv(bv2_handle* h, const void* host_blob, int kind, int i, int j, int d, int e, int32_t* dims,
                          float* w_out, float* bias_out);

/* windowed relative-position attention; qkv [B][3*H*D + H*(2W+1)][ld] (q rows pre-divided by sqrt(D); the last H*(2W+1)
 * rows are the relative-key logits q_i·Ek[r]/sqrt(D)), ld % 32 == 0, mask [B][T], erv [2W+1][D], out [B][H*D][T] (all DEVICE) */
int bv2_test_attention(void* stream, const float* qkv, int ld, const float* mask, const float* erv, float* out,
                       int B, int H, int D, int T, int W);
/* the same with QK^T / PV on the fp16 matrix core (bv2_set_flow_dtype(BV2_F16)) */
int bv2_test_attention_f16(void* stream, const float* qkv, int ld, const float* mask, const float* erv, float* out,
                       int B, int H, int D, int T, int W);

/* channel LayerNorm family (see bv2_kernels.h LnArgs); all DEVICE pointers, nullable where optional */
int bv2_test_layernorm(void* stream, const float* a, const float* add, int mode, const float* dww, const float* dwb, int dil,
                       const float* in_mask, const float* gamma, const float* beta, int post_gelu, const float* res,
                       const float* vec, const float* mask, float* out, int B, int C, int T, int nslab, int64_t slab_stride);

/* one fused DDSConv layer (kernels/dds_fused.hip) on x [B][C][T] -> out [B][C][T] (out != x): depthwise k=3 conv (dww_host [C][3],
 * dwb_host [C], dilation dil) + LN1 (g1/b1) + GELU + 1x1 conv (w_host [C][C], bias_host [C]) + LN2 (g2/b2) + GELU + residual,
 * times mask when last_mask.  Optional ConvFlow.pre input transform: pre_w/pre_b HOST [C] with z [B][2][T], z_src and g [B][C][T]
 * (x is ignored then).  Optional post projection post_w_host [post_cout][C] / post_b_host [post_cout]: with post_out
 * [B][post_cout][T] it is written as (W y + b) * mask; with post_out NULL and zio [B][2][T] the 29 rows are the spline
 * parameters applied to zio (z_src / z_dst).  Host arrays are packed into wpack_dev (>= bv2_test_dds_pack_floats(C) floats). */
int64_t bv2_test_dds_pack_floats(int C);
int bv2_test_dds_layer(void* stream, const float* x, const float* pre_w_host, const float* pre_b_host, const float* z, int z_src,
                       const float* g, const float* mask, const float* dww_host, const float* dwb_host, const float* g1_host,
                       const float* b1_host, const float* g2_host, const float* b2_host, const float* w_host,
                       const float* bias_host, float* out, int dil, int last_mask, const float* post_w_host,
                       const float* post_b_host, int post_cout, float* post_out, float* zio, int z_dst, float* wpack_dev,
                       int B, int C, int T);

/* tuning experiments (tools/kbench.py): force the split-K wave count / C_in chunk / tile-count target of the fp32 conv; 0 = default */
void bv2_test_set_tuning(int splitk_waves, int force_ck, long tile_target);
/* conv_x6.hip: forced tile id (9 / 11, see bv2_kernels.h TILE_X6_*) per C_out class (multiple of 256 / of 128 / other); the last argument is unused */
void bv2_test_set_x6_tuning(int t256, int t128, int t64, int ck);
/* workgroups per CU hipOccupancyMaxActiveBlocksPerMultiprocessor grants the conv_x6 variants {128x64, 128x64 + loader waves, 64x128, 32x256} */
void bv2_test_x6_occupancy(int* out4);
/* bf16 / fp16 conv variants: cl_spec "<nt>:<id>[,...]" forces bf16 variant <id> for launches with <nt> 32-channel tiles ("" = the
 * shipped choice), cl_generic / hc_generic = 1 force the generic GEMM loop instead of the C_in-specialised tap-major one */
void bv2_test_set_variants(const char* cl_spec, int cl_generic, int hc_generic);

/* tools/timeline.py: while dev_buf (capacity_u64 zeroed uint64 on the DEVICE) is set, every fp32 conv launch records 8 uint64 per
 * workgroup {s_memtime at start, after the prologue, after the main loop, at the end, HW_ID, XCC_ID, k or units, valid} into its own
 * slice; bv2_test_conv_timeline_report fills 8 int64 per launch {offset, grid x, y, z, tile (BM*1000+BN), k0|k1<<8|k2<<16, cin, L}. */
void bv2_test_conv_timeline(void* dev_buf, long long capacity_u64);
int bv2_test_conv_timeline_report(long long* meta, int max_launches);

/* inverse RQ spline on channel `dst` of z [B][2][T] with params [B][prow][T] (DEVICE) */
int bv2_test_spline(void* stream, float* z, int src, int dst, const float* params, int prow, const float* mask,
                    float sqrt_fc, float tail, int B, int T);

#ifdef __cplusplus
}
#endif
#endif

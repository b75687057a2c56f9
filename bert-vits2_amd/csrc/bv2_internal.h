// bv2_internal.h — host-side model description (packed-weight layout), handle, and executor interfaces of libbv2.so.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/bv2.h"
#include "bv2_kernels.h"

namespace bv2 {

constexpr int kAttnWindow = 4;       // reference attentions.py:46
constexpr int kCondLayer = 2;        // reference attentions.py:69-71
constexpr int kSdpLayers = 3;        // DDSConv n_layers, reference models.py:171-173
constexpr int kSdpKernel = 3;
constexpr int kSdpBins = 10;
constexpr int kSdpFlowsUsed = 3;     // ConvFlows applied at inference (the 4th is the dropped "useless vflow")
constexpr int kDpFilter = 256;       // reference models.py:929-931
constexpr int kDpKernel = 3;
constexpr int kFlowKernel = 5;       // reference models.py:903-924
constexpr int kMaxLayers = 16;
constexpr int kMaxFlows = 8;
constexpr uint32_t kBlobMagic = 0x32765642u;  // "BVv2"
constexpr int kBlobHeaderFloats = 64;

// ---- packed-weight descriptors: offsets are in floats from the start of the blob ----
struct ConvW {
  int cin = 0, cin_pad = 0, cout = 0, cout_pad = 0, w_ld = 0, k = 1;
  int64_t w_off = -1, b_off = -1;    // b_off < 0: no bias
  int64_t wb_off = -1;               // Generator convs only: bf16 fragment stream (cl_w_index), offset in floats
  int64_t wh_off = -1;               // flow Encoder convs only: fp16 fragment stream (cl_w_index), offset in floats
  int64_t wx_off = -1;               // wide Generator ResBlock convs only: three bf16 planes (x6_w_index), offset in floats
  int64_t wy_off = -1;               // fp32 Generator ResBlock convs with x6 planes: 1 / S_w, then (X3_HDR_FLOATS in) two fp16 planes (x3_w_index)
};
struct VecW { int64_t off = -1; int64_t n = 0; };
struct GemvW { int cout = 0, cin = 0; int64_t w_off = -1, b_off = -1; };

struct EncLayerW {
  ConvW qkv, o, ffn1, ffn2;
  VecW erv, g1, b1, g2, b2;
};
struct EncoderW {
  int n_layers = 0, ksize = 1, hidden = 0, filter = 0, heads = 0;
  GemvW spk;
  EncLayerW layer[kMaxLayers];
};
struct DDSLayerW { VecW dww, dwb; ConvW c1x1; VecW g1, b1, g2, b2; int dil = 1; };
struct DDSW { DDSLayerW l[kSdpLayers]; };
struct ConvFlowW { VecW pre_w, pre_b; DDSW convs; ConvW proj; };

struct CouplingW {
  bool flipped = false;              // channel Flip folded into pre/post weight permutations
  ConvW pre, post;
  EncoderW enc;                      // transformer flow
  GemvW wn_cond;                     // residual (WN) flow
  ConvW wn_in[kMaxLayers];             // rows in gate order (ACT_GATE): per 32-row tile 16 tanh rows then their 16 sigmoid rows
  ConvW wn_res[kMaxLayers], wn_skip[kMaxLayers];   // res_skip_layers split into its x-update rows and its output rows
  int wn_layers = 0;
};

struct UpW {
  int u = 1, k = 1, cin = 0, cout = 0, ntaps = 1;
  ConvW phase[BV2_MAX_UPS];          // one conv problem per output phase (polyphase ConvTranspose1d)
  int pad_left[BV2_MAX_UPS];
  // channels-last form (bf16 path): ONE conv cin -> u*cout over the union of the phases' tap windows (zero weights where
  // a phase does not use a tap); row t of its output is rows t*u .. t*u+u-1 of the upsampled tensor
  ConvW cl;
  int cl_pad_left = 0;
};

struct Model {
  bv2_config cfg;
  bool flow_flip_first = false;      // odd number of couplings: the reverse pass starts with a real channel Flip of z (bv2_model.cpp)
  // enc_p
  VecW emb, tone_emb, lang_emb;
  ConvW bert[3];
  EncoderW enc;
  ConvW proj_m, proj_logs;
  // durations
  ConvW sdp_pre, sdp_proj;
  GemvW sdp_cond;
  DDSW sdp_convs;
  ConvFlowW cf[kSdpFlowsUsed];       // application order: flows.7, flows.5, flows.3
  VecW ea_m, ea_logs;
  GemvW dp_cond;
  ConvW dp_c1, dp_c2, dp_proj;
  VecW dp_g1, dp_b1, dp_g2, dp_b2;
  VecW emb_g;
  // flow: application order (reverse pass): a = 0 is reference flows.{2(n-1)}
  int n_coupling = 0;
  CouplingW coupling[kMaxFlows];
  // dec
  ConvW conv_pre;
  GemvW dec_cond;
  int n_ups = 0, n_rbk = 0, n_rbd = 0;
  int rb_type = 1;                   // 1 = ResBlock1 ((dilated conv, conv) pairs), 2 = ResBlock2 (one conv per dilation: rb[..][d][0] only)
  UpW ups[BV2_MAX_UPS];
  ConvW rb[BV2_MAX_UPS][BV2_MAX_RESBLOCK_KERNELS][BV2_MAX_RESBLOCK_DILATIONS][2];
  // whole-ResBlock bf16 streams of the narrow stages (kernels/resblock_cl_bf16.hip); -1 where the stage is too wide
  int64_t rbcl_w_off[BV2_MAX_UPS][BV2_MAX_RESBLOCK_KERNELS];
  int64_t rbcl_b_off[BV2_MAX_UPS][BV2_MAX_RESBLOCK_KERNELS];
  // the C = 16 stage once more in the tap-pair layout of kernels/resblock_c16_bf16.hip (rb16_w_index); -1 elsewhere
  int64_t rb16_w_off[BV2_MAX_UPS][BV2_MAX_RESBLOCK_KERNELS];
  int64_t rb16_b_off[BV2_MAX_UPS][BV2_MAX_RESBLOCK_KERNELS];
  VecW conv_post;
  int post_c = 0, post_k = 7;
  int total_up = 1;
  int64_t total_floats = 0;
  uint32_t cfg_hash = 0;
};

struct HostTensor { std::vector<float> data; std::vector<int64_t> shape; };

struct ProfileRec { hipEvent_t e0, e1; int fam; double flops, bytes; };

struct Tap { float* dst; int64_t cap; };

}  // namespace bv2

struct bv2_handle {
  bv2::Model model;
  std::map<std::string, bv2::HostTensor> tensors;
  const float* blob = nullptr;       // device
  std::string err;
  std::map<std::string, bv2::Tap> taps;
  int gen_dtype = BV2_F32;           // Generator arithmetic: BV2_F32 (conv_mfma.hip) or BV2_BF16 (gen_bf16.hip)
  int flow_dtype = BV2_F32;          // transformer-flow Encoder convs: BV2_F32 (conv_mfma.hip) or BV2_F16 (enc_f16.hip)
  // bv2_set_option switches (tests compare the fused kernels with the layer-wise ones)
  bool no_conv_x6 = false;           // "conv_x6" = 0: wide Generator convs on the fp32 matrix core (conv_mfma.hip) instead of the bf16x6 form
  bool no_conv_x3 = false;           // "conv_x3" = 0: the wide fp32 Generator convs on the three-plane bf16 form (six products) instead of the two-plane fp16 form
  bool x6_narrow = true;             // "conv_x6_c32" = 0: the C = 32 stage on the fused fp32 pair kernel instead of layer-wise on conv_x6.hip
  bool no_fused_resblock = false;    // "fused_resblock" = 0: narrow Generator stages layer by layer
  int prefetch = 0;                  // "prefetch": bit 0 LayerNorm launches, bit 1 split-K launches carry the next launch's weight stream (batch 1); measured: nothing at config 2 (profiles/r05_ab_prefetch_c2.txt), off
  bool no_xcd_affine = false;        // "xcd_affine" = 0: plain grids for the fp16 Encoder stacks (default: batch item b on XCD b % 8 when B >= 8 && (B % 8 == 0 || B >= 32), Ctx::xcd_affine)
  bool no_f16_fused_ln = false;      // "f16_fused_ln" = 0: LayerNorm-1 / the plain LayerNorm-2s of the fp16 Encoder stacks as launches of their own instead of in conv_o's / conv_2's epilogue
  int f16_wn = 0, f16_ni = 0;        // "f16_wn" / "f16_ni": tile tuning of the fp16 Encoder convs (HcLaunch wn_pref / ni_pref); 0 = the launcher's choice
  bool no_f16_kv = false;            // "f16_kv" = 0: the fp16 Encoder stacks' q/k/v projection writes K and V as fp32 rows and the attention kernel rounds them in registers
  bool no_f16_ksplit = false;        // "f16_ksplit" = 0: the fp16 FFN conv_2 (768 -> 192 rows, 64-column tiles) as 6 waves over three staged chunks instead of 12 waves on K halves of one tile
  bool no_conv_post_rows = false;    // "conv_post_rows" = 0: the bf16 path's conv_post + tanh on the any-width kernel also at C = 16 (default: the row-wise kernel, gen_bf16.hip)
  bool no_ups_phase_taps = false;    // "ups_phase_taps" = 0: the bf16 ConvTranspose1d launches multiply through the zero taps of the union window (A/B and bit-identity tests)
  bool no_stage_sum = true;          // "stage_sum" = 1: the launch that finishes a bf16 Generator stage writes ONE tensor, the branch mean (default 0: the n branch tensors are handed over and
  // the next launch forms the mean).  Measured round 6 (profiles/r06_ab_stage_sum.txt, r06_fam_stage_sum.txt, B = 32): the consumers gain 228 us per step (four
  // ConvTranspose1d launches read 1/3 of the bytes), the producers lose 346 us — a workgroup that runs three branches back to back lives 3x as long (tail rounds of
  // a 416-workgroup grid, no k = 11 / k = 3 tiles side by side on a CU any more), every branch pays the widest branch's halo, the running sum is re-read through L2:
  // 14.65 vs 14.66-14.72 ms at config 3, 15.7-15.9 vs 15.6 at config 5.  Bit-identical either way (tests/test_stage_sum_gpu.py).
  bool no_resblock_c16 = false;      // "resblock_c16" = 0: the C = 16 bf16 stage on the 32x32x16 whole-ResBlock kernel (resblock_cl_bf16.hip) instead of resblock_c16_bf16.hip
  bool no_respair_c32 = false;       // "respair_c32" = 0: the C = 32 bf16 stage as whole-ResBlock launches (resblock_cl_bf16.hip) instead of pair by pair
  int respair_form = 1;              // "respair_form": 1 = 64 x 128 wave tiles (respair2_cl_bf16_kernel), 0 = 32-channel waves
  bool respair_problem_major = false; // "respair_mix" = 0: the pair kernel's branches dispatched one after the other (A/B only)
  bool x6_pair_c128 = false;         // "x6_pair_c128" = 1: the pair kernel also on the C = 128 stage (one 8-wave workgroup per CU; measured, see DESIGN)
  bool no_x6_pair_c16 = false;       // "x6_pair_c16" = 0: the C = 16 stage on the fused fp32-MFMA pair kernel (resblock_fused.hip)
  bool no_x6_pair_c64 = false;       // "x6_pair_c64" = 0: the pair kernel on the C = 32 stage only
  bool no_x6_pair = false;           // "x6_pair" = 0: the C = 32 fp32 stage as two conv_x6 launches per ResBlock pair instead of one respair_x6 launch
  bool no_fused_boundary = false;    // "fused_boundary" = 0: LayerNorm-2, post and the next pre of the transformer flow as three launches
  bool no_fused_respair = false;     // "fused_respair" = 0: the wide bf16 Generator stages one conv per launch (gen_bf16.hip) instead of one pair per launch
  bool no_fused_attn_o = false;      // "fused_attn_o" = 0: conv_o as its own launch after the attention kernel
  int attn_ksplit = -1;              // "attn_ksplit": key ranges per (head, query tile) of the fused attention; -1 = picked per shape, 0 / 1 = off
  bool no_fused_dds = false;         // "fused_dds" = 0: DDSConv layers as 3 launches each
  bool no_overlap_dp = true;         // "overlap_dp" = 1: the (independent) DurationPredictor on an internal side stream, forked from and
  // joined back into the caller's stream with events (created on first use; capturable).  OFF by default: measured on MI355X at
  // batch 1 the fork/join costs more than the 5 short launches it hides (4.70 -> 4.82 ms per step eager, 4.75 -> 4.78 replayed).
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // profiling
  bool prof_on = false;
  int prof_mode = 1;                 // 1: every MFMA kernel launch, 2: Generator (dec.*) launches only, 3: per site, 4: dec.ups per site
  std::vector<bv2::ProfileRec> prof_pool;
  size_t prof_used = 0;
  std::vector<std::string> prof_names;
};

namespace bv2 {

// bv2_model.cpp
int build_layout(Model& m, std::string& err);                       // fills every descriptor + total_floats
int pack_blob(const Model& m, const std::map<std::string, HostTensor>& t, float* blob, std::string& err);
bool key_in_schema(const Model& m, const std::string& key);

// bv2_exec.cpp
int64_t workspace_bytes(const Model& m, int B, int T, int Ty);
int run_encode(bv2_handle* h, hipStream_t s, const bv2_encode_in& in, const bv2_encode_out& out, void* ws, int64_t wsb);
int run_decode(bv2_handle* h, hipStream_t s, const bv2_decode_in& in, const bv2_decode_out& out, void* ws, int64_t wsb);
int run_flow(bv2_handle* h, hipStream_t s, int B, int Ty, const float* z_p, const int64_t* y_lengths, const float* y_mask,
             const float* g, float* z, void* ws, int64_t wsb);
int run_stage_emb_g(bv2_handle* h, hipStream_t s, int B, const int64_t* sid, float* g);
int run_stage_enc_p(bv2_handle* h, hipStream_t s, int B, int T, const int64_t* x, const int64_t* tone, const int64_t* lang,
                    const float* b0, const float* b1, const float* b2, const float* g, const int64_t* x_lengths, float* xout,
                    float* m_p, float* logs_p, float* x_mask, void* ws, int64_t wsb);
int run_stage_sdp(bv2_handle* h, hipStream_t s, int B, int T, const float* x, const float* x_mask, const float* zin,
                  const float* g, float* logw, void* ws, int64_t wsb);
int run_stage_dp(bv2_handle* h, hipStream_t s, int B, int T, const float* x, const float* x_mask, const float* g, float* logw,
                 void* ws, int64_t wsb);
int run_generator(bv2_handle* h, hipStream_t s, int B, int Ty, int L, const float* z, const int64_t* y_lengths,
                  const float* g, float* o, void* ws, int64_t wsb);

}  // namespace bv2

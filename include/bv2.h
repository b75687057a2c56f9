/*
 * bv2.h — C ABI of libbv2.so, the MI355X-native (gfx950) SynthesizerTrn.infer() hot path of Bert-VITS2 v2.3.
 *
 * The reference (fishaudio/Bert-VITS2) is 100 % Python and has NO FFI/operator interface; the only seam is the
 * Python class models.SynthesizerTrn (reference models.py:811-1074).  This header is therefore the boundary a
 * maintainer would bind from the reference's Python (ctypes stub shown in INTEGRATION.md); each entry point names
 * the reference code it replaces.  The stage split mirrors the reference's own ONNX export, which cuts infer() into
 * emb / enc_p / sdp / dp / flow / dec and makes both noise draws explicit inputs
 * (reference onnx_modules/V230/models_onnx.py:896-1063, onnx_modules/V220_OnnxInference/__init__.py:88-117).
 *
 * Conventions
 *  - plain C, no torch types: raw pointers + sizes.  All activations are fp32, layout [B, C, T], T contiguous
 *    (the reference's layout).  Integer inputs are int64 like the reference's LongTensors.
 *  - every DEVICE buffer (inputs, outputs, workspace, packed weights) is allocated and owned by the CALLER
 *    (PyTorch in the shim).  The library never hipMalloc's; it only launches kernels / async copies on the
 *    stream it is handed, so every entry point is hipGraph-capturable.
 *  - status: 0 = ok, negative = error; bv2_last_error(h) holds the message (no exceptions cross the ABI).
 *    The reference raises Python exceptions at the same places (shape asserts infer.py:124, ValueError
 *    transforms.py:113-114); the Python shim turns a non-zero status into RuntimeError.
 *  - one handle per GPU / per caller thread (thread-compatible, like the single-threaded reference).
 */
#ifndef BV2_H
#define BV2_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BV2_ABI_VERSION 3   /* 3: bv2_decode_in.nz_tstride, the six ONNX-seam stage calls, bv2_detach_weights,
                               pack-layout version in the blob header */
#define BV2_PACK_LAYOUT 16   /* bumped whenever bv2_model.cpp changes the order / format of anything inside the packed blob:
                               a blob cached on disk by an older packer is rejected by bv2_attach_weights */
#define BV2_MAX_UPS 8
#define BV2_MAX_RESBLOCK_KERNELS 4
#define BV2_MAX_RESBLOCK_DILATIONS 4

typedef struct bv2_handle bv2_handle;
typedef void* bv2_stream; /* hipStream_t */

/* Construction arguments of reference SynthesizerTrn.__init__ (models.py:816-842) that shape the hot path,
 * as infer.get_net_g derives them from configs/config.json (infer.py:95-101). */
typedef struct bv2_config {
  int32_t struct_bytes;            /* = sizeof(bv2_config) */
  int32_t n_vocab, n_tones, n_languages, bert_dim;      /* 112, 12, 3, 1024 (text/symbols.py:167-183) */
  int32_t inter_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size;
  int32_t gin_channels, n_speakers;
  int32_t use_transformer_flow, n_flow_layer, n_layers_trans_flow;
  int32_t n_upsamples;
  int32_t upsample_rates[BV2_MAX_UPS];
  int32_t upsample_kernel_sizes[BV2_MAX_UPS];
  int32_t upsample_initial_channel;
  int32_t n_resblock_kernels;
  int32_t resblock_kernel_sizes[BV2_MAX_RESBLOCK_KERNELS];
  int32_t n_resblock_dilations;
  int32_t resblock_dilation_sizes[BV2_MAX_RESBLOCK_KERNELS][BV2_MAX_RESBLOCK_DILATIONS];
  /* appended in round 5 (a caller built against the shorter struct passes its own struct_bytes and gets ResBlock1):
   * 0 / 1 = modules.ResBlock1 (convs1 / convs2 pairs, reference modules.py:239-315), 2 = modules.ResBlock2 (ONE conv per dilation,
   * n_resblock_dilations = 2: `resblock: "2"` of models.py:508, modules.py:318-363) */
  int32_t resblock_type;
} bv2_config;

enum { BV2_F32 = 0, BV2_F16 = 1, BV2_BF16 = 2 };

/* ---- lifetime ------------------------------------------------------------------------------------------- */
int bv2_abi_version(void);
/* replaces SynthesizerTrn.__init__ (models.py:816-935). */
int bv2_create(const bv2_config* cfg, bv2_handle** out);
void bv2_destroy(bv2_handle* h);
/* h may be NULL: returns the message of the last failed bv2_create on this thread. */
const char* bv2_last_error(const bv2_handle* h);

/* ---- weights: replaces utils.load_checkpoint (utils.py:65-120) + per-forward weight_norm ---- */
/* Hand over one tensor of the reference state_dict by its reference key.  host_ptr is HOST
 * memory, copied.  Keys outside the inference schema (enc_q.*, sdp.post_*) are accepted and ignored (returns 1).
 * Both weight-norm forms are accepted: <p>.weight_g + <p>.weight_v, or the folded <p>.weight that
 * Generator.remove_weight_norm (models.py:559-564) leaves behind. */
int bv2_load_tensor(bv2_handle* h, const char* ref_key, const void* host_ptr, const int64_t* shape, int ndim, int dtype);
/* Size in bytes of the packed weight blob for this config. */
int64_t bv2_packed_bytes(const bv2_handle* h);
/* Fold weight-norm (w = g*v/||v||, dim 0; C_in for ConvTranspose1d), fold the flows' channel Flips into weight
 * permutations, fuse q/k/v, split ConvTranspose into polyphase taps, pad to MFMA tile multiples, and write the
 * packed blob into host_blob (HOST memory, bv2_packed_bytes large).  Fails listing any missing tensor. */
int bv2_pack_weights(bv2_handle* h, void* host_blob, int64_t bytes);
/* Point the handle at a packed blob resident in DEVICE memory (caller-owned; e.g. uploaded by rank 0 and broadcast
 * to the other GPUs with RCCL).  Verifies the blob header against this handle's config. */
int bv2_attach_weights(bv2_handle* h, const void* dev_blob, int64_t bytes);
/* Forget the attached blob (the caller is about to free or move it): every compute entry point fails with -8 until the
 * next bv2_attach_weights.  Captured graphs that baked the old address in must be destroyed by the caller. */
int bv2_detach_weights(bv2_handle* h);

/* ---- precision (BASELINE config 3: "bf16 weights/activations, fp32 accumulate") ------------------------------- */
/* Arithmetic of the HiFi-GAN Generator (dec, reference models.py:538-557 — 90 % of the path's FLOPs): BV2_F32 (default;
 * v_mfma_f32_32x32x2_f32, exact fp32) or BV2_BF16 (v_mfma_f32_32x32x16_bf16: bf16 weights and channels-last bf16
 * activations, fp32 accumulation / bias / residual, fp32 conv_post + tanh).  The reference's counterpart is running dec
 * under torch.autocast(bfloat16).  Text encoder, duration predictors, length regulation and flow stay fp32 in both modes
 * (durations stay bit-stable).  The packed blob always carries both weight forms, so this is a per-call switch. */
int bv2_set_generator_dtype(bv2_handle* h, int dtype);
/* Arithmetic of the transformer flow's Encoder convolutions (BASELINE config 5 "fp16 flow + fp32 spline"): the fused
 * q/k/v projection, conv_o and the FFN conv_1/conv_2 of TransformerCouplingLayer.enc (reference modules.py:561-580,
 * attentions.py:103-120, 263-266, 438-446) — 95 % of the flow's FLOPs.  BV2_F32 (default) or BV2_F16
 * (v_mfma_f32_32x32x16_f16: fp16 weights and conv inputs, fp32 accumulation; the FFN hidden activation is stored fp16;
 * the attention core's two matrix products QK^T and PV take their operands rounded to fp16 in registers).
 * LayerNorm, logits / softmax / running statistics, the residual stream, pre/post and everything before the flow (text
 * encoder, durations, spline) stay fp32, so durations and the alignment path are unchanged.  The reference's counterpart is running
 * flow under torch.autocast(float16).  Only the transformer flow has an fp16 form (returns -2 otherwise). */
int bv2_set_flow_dtype(bv2_handle* h, int dtype);

/* ---- workspace ------------------------------------------------------------------------------------------- */
/* Bytes of caller-provided DEVICE scratch needed by any stage call with batch B, T symbols, Ty_max frames. */
int64_t bv2_workspace_bytes(const bv2_handle* h, int B, int T, int Ty_max);

/* ---- phase A: models.py:1045-1057 (emb_g, enc_p, sdp, dp, exp/ceil/sum) ------------------------------------- */
typedef struct bv2_encode_in {
  int32_t B, T;
  const int64_t* x;          /* [B,T] symbol ids */
  const int64_t* x_lengths;  /* [B] */
  const int64_t* sid;        /* [B] */
  const int64_t* tone;       /* [B,T] */
  const int64_t* language;   /* [B,T] */
  const float* bert;         /* [B,bert_dim,T] */
  const float* ja_bert;      /* [B,bert_dim,T] */
  const float* en_bert;      /* [B,bert_dim,T] */
  const float* noise_w;      /* [B,2,T]  N(0,1): the torch.randn of models.py:248-251, drawn by the caller */
  float noise_scale_w, sdp_ratio, length_scale;
  /* Optional WORD-level BERT features.  The reference runs the BERT model, copies hidden_states[-3] to the host,
   * repeats word i's row word2ph[i] times (text/chinese_bert.py:37, 48-58) and uploads [1024, T] again.  With
   * bert_index[f] != NULL feature f (0 = bert, 1 = ja_bert, 2 = en_bert) is handed over as the BERT model emitted it,
   * [B, bert_dim, bert_cols[f]] — one column per word, still on the device — and symbol t of utterance b takes column
   * bert_index[f][b*T + t] (the word2ph repeat as a gather inside the TextEncoder front: the repeated matrix never exists).
   * NULL: the feature is [B, bert_dim, T], one column per symbol, as in the reference. */
  const int32_t* bert_index[3];
  int32_t bert_cols[3];
} bv2_encode_in;

typedef struct bv2_encode_out {   /* all DEVICE, caller-allocated */
  float* g;          /* [B,gin]            emb_g(sid)                       models.py:1046 */
  float* x;          /* [B,hidden,T]       encoder output                   models.py:1049 */
  float* m_p;        /* [B,inter,T] */
  float* logs_p;     /* [B,inter,T] */
  float* x_mask;     /* [B,T] (1.0 / 0.0)                                  models.py:392-394 */
  float* logw_sdp;   /* [B,T]  sdp(...) before mixing (may be NULL) */
  float* logw_dp;    /* [B,T]  dp(...)  before mixing (may be NULL) */
  float* logw;       /* [B,T]                                              models.py:1052-1054 */
  float* w_ceil;     /* [B,T]  ceil(exp(logw)*mask*length_scale)           models.py:1055-1056 */
  int64_t* y_lengths;/* [B]    clamp_min(sum(w_ceil),1)                    models.py:1057 */
} bv2_encode_out;

int bv2_encode_durations(bv2_handle* h, bv2_stream stream, const bv2_encode_in* in, const bv2_encode_out* out,
                         void* workspace, int64_t workspace_bytes);

/* ---- phase B: models.py:1058-1073 (sequence_mask, generate_path, expand, z_p, flow reverse, dec) ------------ */
typedef struct bv2_decode_in {
  int32_t B, T;
  int32_t Ty;                /* max(y_lengths), read back by the caller (the reference's one host sync, commons.py:120-122) */
  int32_t max_len;           /* frames fed to dec: (z*y_mask)[:, :, :max_len]; <=0 means all (models.py:1073) */
  const float* m_p;          /* [B,inter,T]  from phase A */
  const float* logs_p;       /* [B,inter,T] */
  const float* x_mask;       /* [B,T] */
  const float* w_ceil;       /* [B,T]  (the caller may substitute durations, as train_ms.evaluate-style tooling does) */
  const int64_t* y_lengths;  /* [B] */
  const float* g;            /* [B,gin] */
  const float* noise_z;      /* N(0,1) for models.py:1071; element (b,c,j) at noise_z[b*nz_bstride + c*nz_cstride + j*nz_tstride].
                                The reference draws it with randn_like on a TRANSPOSED view (strides (C*Ty, 1, C)); passing the
                                strides lets the caller hand that tensor over as it is (the RNG contract). */
  int64_t nz_bstride, nz_cstride, nz_tstride;   /* nz_tstride <= 0 means 1 */
  float noise_scale;
  int32_t exact_lengths;     /* 0: the reference's batch semantics — dec is unmasked, so in a padded batch the activations
                                past an utterance's end bleed into its last ~40 ms (models.py:1073 masks only z).
                                1: every Generator conv treats positions >= y_lengths[b]*(samples per frame at its stage) as
                                zero padding, i.e. each utterance of a ragged batch gets exactly the audio it gets when run
                                alone (what the reference produces, since it only ever infers at batch 1); tiles past an
                                utterance's end are skipped.
                                2: Ty is a BUCKET >= max(y_lengths) (a hipGraph captured once per bucket instead of once per exact
                                T_y — T_y = max(y_lengths) is data-dependent, commons.py:119-123): every Generator conv treats positions
                                >= max_b(y_lengths) * (samples per frame at its stage) as zero padding for EVERY utterance, so the first
                                max(y_lengths) frames of every output are what mode 0 produces at Ty = max(y_lengths); the flow needs
                                nothing (frames past y_lengths[b] are masked in both).  Outputs past that frame are zeros / padding. */
} bv2_decode_in;

typedef struct bv2_decode_out {   /* all DEVICE, caller-allocated; any pointer except o may be NULL */
  float* o;          /* [B,1,S]  S = min(Ty,max_len) * prod(upsample_rates) */
  float* attn;       /* [B,1,Ty,T] one-hot monotone path                   models.py:1061-1062 */
  float* y_mask;     /* [B,1,Ty] */
  float* z;          /* [B,inter,Ty] */
  float* z_p;        /* [B,inter,Ty] */
  float* m_p;        /* [B,inter,Ty] expanded */
  float* logs_p;     /* [B,inter,Ty] expanded */
} bv2_decode_out;

int bv2_decode(bv2_handle* h, bv2_stream stream, const bv2_decode_in* in, const bv2_decode_out* out,
               void* workspace, int64_t workspace_bytes);

/* ---- single stages: the reference's own ONNX cut of infer() --------------------------------------------------------
 * onnx_modules/V230/models_onnx.py:896-1063 exports six graphs (emb_g, enc_p, sdp, dp, flow, dec) and
 * onnx_modules/V230_OnnxInference/__init__.py:44-126 runs them with the numpy glue in between.  The six calls below take
 * the tensors of those graphs under the same names (comments give "onnx name"), so a MoeVS-style consumer can swap the
 * runtime stage by stage; the parity tests tap them against the oracle.  All fp32 [B,C,T] / int64, DEVICE, caller-owned. */
/* emb_g.run({"sid"}) -> g [B,gin]   (models.py:1046; the consumer unsqueezes to [B,gin,1]) */
int bv2_stage_emb_g(bv2_handle* h, bv2_stream stream, int B, const int64_t* sid, float* g);
/* enc.run({"x","t","language","bert_0","bert_1","bert_2","g"}) -> xout, m_p, logs_p, x_mask   (TextEncoder, models.py:377-400).
 * x_lengths may be NULL: every utterance is T symbols long, as in the exported graph (which has no length input). */
int bv2_stage_enc_p(bv2_handle* h, bv2_stream stream, int B, int T, const int64_t* x, const int64_t* t, const int64_t* language,
                    const float* bert_0, const float* bert_1, const float* bert_2, const float* g, const int64_t* x_lengths,
                    float* xout, float* m_p, float* logs_p, float* x_mask, void* workspace, int64_t workspace_bytes);
/* sdp.run({"x","x_mask","zin","g"}) -> logw [B,1,T]: StochasticDurationPredictor reverse (models.py:197-204, 245-256) where
 * zin [B,2,T] is the ALREADY SCALED noise (randn * noise_scale_w), exactly as the exported graph takes it. */
int bv2_stage_sdp(bv2_handle* h, bv2_stream stream, int B, int T, const float* x, const float* x_mask, const float* zin,
                  const float* g, float* logw, void* workspace, int64_t workspace_bytes);
/* dp.run({"x","x_mask","g"}) -> logw [B,1,T]: DurationPredictor (models.py:285-299) */
int bv2_stage_dp(bv2_handle* h, bv2_stream stream, int B, int T, const float* x, const float* x_mask, const float* g,
                 float* logw, void* workspace, int64_t workspace_bytes);
/* flow.run({"z_p","y_mask","g"}) -> z: flow(z_p, y_mask, g, reverse=True), models.py:1072.  The frame mask comes either as
 * y_lengths [B] (int64) or as the graph's y_mask [B,1,Ty] (fp32 0/1) — exactly one of the two is non-NULL.  z_p is not
 * modified; z receives the result. */
int bv2_stage_flow(bv2_handle* h, bv2_stream stream, int B, int Ty, const float* z_p, const int64_t* y_lengths,
                   const float* y_mask, const float* g, float* z, void* workspace, int64_t workspace_bytes);
/* dec.run({"z_in","g"}) -> o: Generator.forward models.py:538-557 on z_in[:, :, :L] (z_in has row stride Ty); with y_lengths
 * non-NULL the input is (z*y_mask)[:, :, :L] as at models.py:1073, with NULL it is taken as it is (the exported graph).
 * o is [B,1,L*prod(rates)]. */
int bv2_stage_generator(bv2_handle* h, bv2_stream stream, int B, int Ty, int L, const float* z, const int64_t* y_lengths,
                        const float* g, float* o, void* workspace, int64_t workspace_bytes);

/* ---- one-shot convenience: whole infer() with an internal stream sync between the phases ---------------------- */
/* Outputs must be sized for Ty_cap frames; returns -3 (and sets *Ty_out) if the realised Ty exceeds Ty_cap.
 * Row strides of the [.,.,Ty] outputs are the realised Ty (*Ty_out), tensors are written densely. */
int bv2_infer(bv2_handle* h, bv2_stream stream, const bv2_encode_in* in, const bv2_encode_out* enc_out,
              const float* noise_z, int64_t nz_bstride, int64_t nz_cstride, int64_t nz_tstride, float noise_scale, int32_t max_len,
              int32_t Ty_cap, const bv2_decode_out* dec_out, int32_t* Ty_out, void* workspace, int64_t workspace_bytes);

/* ---- 16-bit PCM (serving glue; replaces the host-side gradio convert_to_16_bit_wav the reference's callers run after
 * .cpu(): webui.py:86,129, hiyoriUI.py:343) -------------------------------------------------------------------- */
/* pcm[b][i] = (int16) trunc( wave[b][i] / max_j |wave[b][j]| * 32767 ) over the valid samples i, j < y_lengths[b]*hop
 * (capped at S); samples past the utterance are 0; an all-zero utterance stays 0.  wave [B][wave_bstride >= S] fp32,
 * pcm [B][pcm_bstride >= S] int16, peak_scratch [B] uint32 — all DEVICE.  Asynchronous on `stream`, graph-capturable. */
int bv2_pcm16(bv2_stream stream, const float* wave, int64_t wave_bstride, const int64_t* y_lengths, int32_t hop, int32_t B,
              int64_t S, int16_t* pcm, int64_t pcm_bstride, uint32_t* peak_scratch);

/* ---- hipGraph capture (BASELINE config 3: "hipGraph-captured decode") ----------------------------------------- */
/* Both phases are fixed launch sequences on the caller's stream with no allocation, host sync or device->host copy,
 * so they can be recorded once and replayed: bv2_graph_capture_* puts `stream` (which must NOT be the legacy default
 * stream) into capture mode, records the phase exactly as bv2_encode_durations / bv2_decode would launch it, and
 * instantiates an executable graph.  The graph bakes in every pointer and scalar of in/out/workspace: the caller keeps
 * those buffers alive and at the same addresses (refill the inputs in place before each replay) and re-captures when a
 * shape (B, T, Ty, max_len), a scalar argument or a dtype switch changes.  bv2_graph_launch replays on any stream.
 * Taps and profiling must be off while capturing. */
typedef struct bv2_graph bv2_graph;
int bv2_graph_capture_encode(bv2_handle* h, bv2_stream stream, const bv2_encode_in* in, const bv2_encode_out* out,
                             void* workspace, int64_t workspace_bytes, bv2_graph** graph);
int bv2_graph_capture_decode(bv2_handle* h, bv2_stream stream, const bv2_decode_in* in, const bv2_decode_out* out,
                             void* workspace, int64_t workspace_bytes, bv2_graph** graph);
int bv2_graph_launch(bv2_graph* graph, bv2_stream stream);
int bv2_graph_num_nodes(const bv2_graph* graph);
void bv2_graph_destroy(bv2_graph* graph);

/* ---- debugging / measurement ------------------------------------------------------------------------------ */
/* Kernel-selection switches, for tests that hold the fused kernels to the layer-wise ones (default 1 = fused):
 *   "fused_resblock"  the narrow Generator stages as whole-ResBlock / fused-pair kernels (0: one conv per launch; in bf16 mode this
 *                     also takes the C = 32 stage off the pair kernel, i.e. 0 = every narrow stage layer-wise)
 *   "conv_x6"         the ResBlock convs of the fp32 Generator stages with C >= 32 on the bf16 matrix core: operands split exactly
 *                     into three bf16 planes, six cross products accumulated in fp32 — fp32 accuracy (dropped terms < 2^-23 of a
 *                     product) at 6/16 of the fp32-MFMA time (kernels/conv_x6.hip).  0: v_mfma_f32_32x32x2_f32 (conv_mfma.hip)
 *   "conv_x3"         those convs (layer-wise at C % 128 == 0, and the pair kernel of the narrower stages) on the TWO-plane fp16 form:
 *                     operands scaled by powers of two into fp16's range and split into two fp16 halves, three cross products — the
 *                     same order of error (dropped term <= 2^-24 of a product) at half the matrix work.  The activation scale comes
 *                     from the data: max |x| of the whole input tensor, published by the launch that wrote it (layer-wise form), or of
 *                     the workgroup's own tile (pair kernel), so an element more than 2^15 below that maximum keeps an absolute, not a
 *                     relative, error (2^-40 of the maximum), and a non-finite input leaves the whole tensor non-finite or zero.
 *                     0: the three-plane bf16 form above (exact splits, fp32's exponent range)
 *   "conv_x6_c32"     also the C = 32 stage layer-wise on conv_x6.hip (two launches per ResBlock pair, 44.6 us each at batch 1) instead
 *                     of the fused fp32-MFMA pair kernel (one launch, 110 us); 0: resblock_fused.hip.  Only consulted when the C = 32
 *                     stage is NOT on the split-bf16 pair kernel, i.e. together with "x6_pair" = 0 (the default runs respair_x6.hip there)
 *   "fused_respair"   the wide bf16 Generator stages (C = 64 / 128 / 256) one (dilated conv, conv) ResBlock pair per launch, the
 *                     intermediate in LDS (kernels/respair_cl_bf16.hip; bit-identical to the layer-wise path); 0: one conv per launch
 *   "respair_c32"     1 (default): also the C = 32 stage pair by pair (one wave owns all channels); 0: whole-ResBlock launches
 *   "f16_kv"          1 (default): in the fp16 Encoder stacks the fused q/k/v projection writes K (channels-last) and V (channel-major) as fp16 in the
 *                     layouts the attention kernel's matrix products consume (kernels/attention.hip KV16: 6 loads of 16 B per key tile instead of 48
 *                     dword loads + conversions, 80 fewer registers); 0: fp32 rows, rounded to fp16 in the attention kernel's registers.  Same fp16
 *                     values either way; measured -0.6 us per attention launch at B = 32 and nothing on the projection (profiles/r06_fam_f16_kv.txt)
 *   "f16_wn"          tile tuning of the fp16 Encoder convs (0 = the launcher's choice): waves per workgroup (4 / 6 / 8, taken when it wastes no more wave
 *                     slots than the default).  Measured round 6 (profiles/r06_ab_f16_tiles.txt): four-wave workgroups for the FFN conv_1 at B = 32
 *                     give 14.56 vs 14.62 ms — inside the spread, default unchanged
 *   "f16_ni"          the same for the time steps per workgroup (2 = 64, 4 = 128; 0 = the launcher's choice)
 *   "stage_sum"       0 (default, measured: no gain — what the consumers save the producers lose, see bv2_internal.h); 1: the launch that finishes a bf16 Generator stage's ResBlocks (the last pair launch of the wide stages, the whole-ResBlock
 *                     launch at C = 16) runs the stage's n branches tile by tile in one workgroup and writes ONE tensor — the branch mean of reference
 *                     models.py:545-552 (`xs / self.num_kernels`) — so the next ConvTranspose1d / conv_post reads one tensor instead of n; 0: n
 *                     tensors, the consumer forms the mean.  Same rounding points either way (kernels/cl_bf16.h stage_mean): bit-identical results
 *   "resblock_c16"    1 (default): the C = 16 bf16 stage's whole-ResBlock launch on v_mfma_f32_16x16x32_bf16 (two taps x 16 channels per
 *                     instruction, unpadded 32-byte LDS rows, two workgroups per CU: kernels/resblock_c16_bf16.hip); 0: the 32x32x16
 *                     whole-ResBlock kernel (resblock_cl_bf16.hip), whose MFMA block is half zero padding at this width
 *   "f16_fused_ln"    1 (default): in the fp16 Encoder stacks LayerNorm-1 runs in conv_o's epilogue and every LayerNorm-2 but the stack's last (with the speaker add of the
 *                     conditioning layer where it follows) in the FFN conv_2's — the workgroup owns all 192 channels of
 *                     its columns (kernels/enc_f16.hip); 0: LayerNorm launches of their own (layernorm.hip)
 *   "f16_ksplit"      1 (default): the fp16 Encoder stacks' FFN conv_2 (C_in = 768 -> 192 rows) on 64-column tiles splits K inside the workgroup:
 *                     the whole 768-channel tile staged once, 12 waves = 6 output tiles x 2 channel halves, partial sums merged through LDS
 *                     (kernels/enc_f16.hip); 0: 6 waves over three staged 256-channel chunks.  Same products, another fp32 summation order
 *   "conv_post_rows"  1 (default): the bf16 path's conv_post + tanh at C = 16, k = 7 row-wise (a thread owns one 32-byte input row per branch, K floats
 *                     per thread through LDS); 0: the any-width kernel (C*K scalar LDS reads per output sample).  fp32 arithmetic in both
 *   "ups_phase_taps"  1 (default): a bf16 ConvTranspose1d launch (one conv with C_out' = u*C_out over the union of the phases' tap windows) runs,
 *                     per wave, only the window taps of the phases its output channels belong to — the zero weights that pad the other
 *                     phases' taps are stepped over (1/3 of the matrix work at k = 2u); 0: every tap of the window (same results)
 *   "prefetch"        default 0 (measured: within noise at config 2 — the batch-1 launches do not wait for their weights; the BERT extractor,
 *                     where it is worth 0.6 %, has it on: bv2_bert_set_option).  Batch 1 (small-N regime): bit 0 — a LayerNorm launch carries the NEXT launch's weight stream (FFN conv_1 behind
 *                     LayerNorm-1, the next layer's q/k/v projection behind LayerNorm-2), bit 1 — a split-K conv launch does (conv_2 behind
 *                     conv_1): 128 spare workgroups touch it line by line into the L2 of the XCD whose workgroups will read it
 *                     (bv2_kernels.h Prefetch).  0: every launch fetches its own weights from HBM / the Infinity Cache when it starts
 *   "xcd_affine"      1 (default): in the fp16 Encoder stacks at batch 8, 16, 24 and any batch >= 32 (B >= 8 && (B % 8 == 0 || B >= 32): the batch fills the eight XCDs about evenly), batch item b runs on XCD b % 8 in EVERY kernel of a layer (q/k/v,
 *                     attention, conv_o, LayerNorms, FFN convs), so a layer's tensors are handed on inside one XCD's L2 (the eight L2s are
 *                     not coherent with each other); 0: plain grids
 *   "respair_form"    1 (default): 64-channel x 128-row wave tiles on the XOR-swizzled tile; 0: 32-channel waves on the padded tile
 *   "respair_mix"     1 (default): the k = 11 / 7 / 3 branches of a pair launch interleaved in dispatch order; 0: branch after branch
 *   "fused_dds"       one launch per DDSConv layer incl. the projection / spline that follows (0: 3 launches per layer)
 *   "x6_pair"         fp32 mode, the C = 64 / 32 / 16 Generator stages one (dilated conv, conv) ResBlock pair per launch with both convs on
 *                     the bf16 matrix core and the intermediate planes in LDS (kernels/respair_x6.hip; bit-identical to the two conv_x6
 *                     launches); 0: two launches per pair
 *   "x6_pair_c64"     1 (default); 0: "x6_pair" without the C = 64 stage
 *   "x6_pair_c16"     1 (default); 0: "x6_pair" without the C = 16 stage (back on resblock_fused.hip)
 *   "x6_pair_c128"    0 (default); 1: "x6_pair" also on the C = 128 stage (one 8-wave workgroup per CU: measured, no gain)
 *   "fused_boundary"  transformer flow, small-batch fp32 regime: LayerNorm-2 of a coupling's last Encoder layer, its `post` and the next
 *                     coupling's `pre` in one launch (kernels/flow_boundary.hip); 0: three launches
 *   "fused_attn_o"    MultiHeadAttention.conv_o inside the attention kernel in the small-batch fp32 regime: head h writes partial
 *                     slab h, the LayerNorm sums the slabs (0: conv_o as its own launch)
 *   "attn_ksplit"     -1 (default): the key ranges per (head, query tile) of the fused attention are picked per shape (2 or 4 at batch 1 and long
 *                     sequences: each range's workgroup writes its own partial slab, LayerNorm-1 merges them with the flash-decoding weights);
 *                     0 / 1: no key split; 2 / 4: that many ranges where the slabs allow it
 *   "overlap_dp"      default 0: 1 runs the DurationPredictor on an internal side stream beside the stochastic one (fork / join
 *                     with events on the caller's stream; measured slower at batch 1, kept for experiments)
 * Captured graphs keep whatever was selected when they were recorded. */
int bv2_set_option(bv2_handle* h, const char* key, int value);

/* Ask the executor to copy a named intermediate (e.g. "dec.ups.0", "dec.stage.2", "flow.3.h", "enc.layer.1")
 * into dev_dst (capacity in floats) the next time it is produced.  name==NULL clears all taps. */
int bv2_set_tap(bv2_handle* h, const char* name, float* dev_dst, int64_t capacity_floats);

/* Per-kernel-family timing with HIP events recorded on the caller's stream (bench.py's roofline leg). */
typedef struct bv2_profile_row {
  char name[96];          /* kernel family, e.g. "conv1d_mfma<128x128>" (mode 3: "site|family shape") */
  int64_t launches;
  double total_ms;        /* sum of event-timed durations */
  double flops;           /* algorithmic FLOPs (2*MAC) over those launches */
  double bytes;           /* algorithmic bytes (inputs read once + outputs written once + weights) */
} bv2_profile_row;
int bv2_profile_enable(bv2_handle* h, int on);          /* 0 off, 1 every MFMA kernel launch, 2 Generator launches only,
                                                            3 every MFMA launch with one row per launch site and shape,
                                                            4 the Generator's ConvTranspose1d ("dec.ups") launches only, one row each */
int bv2_profile_reset(bv2_handle* h);
/* Synchronises the recorded events and aggregates them; returns the number of rows written (<= max_rows). */
int bv2_profile_report(bv2_handle* h, bv2_profile_row* rows, int max_rows);

#ifdef __cplusplus
}
#endif
#endif /* BV2_H */

/* bv2_bert.h — C ABI of the BERT feature extractor in front of SynthesizerTrn.infer(), libbv2.so.
 *
 * What it replaces (reference, 100 % Python): text/chinese_bert.py:15-37 —
 *     models[device] = AutoModelForMaskedLM.from_pretrained("./bert/chinese-roberta-wwm-ext-large").to(device)
 *     res = models[device](**tokenizer(text, return_tensors="pt"), output_hidden_states=True)
 *     res = torch.cat(res["hidden_states"][-3:-2], -1)[0].cpu()
 * i.e. the forward pass of a HuggingFace `BertModel` (chinese-roberta-wwm-ext-large is model_type "bert": 24 layers, hidden 1024,
 * 16 heads, intermediate 4096, learned absolute positions, post-LayerNorm, erf-GELU) up to hidden_states[-3] = the output of
 * encoder layer 22 — the last two layers and the MLM head never run here.  The algorithm lives in a third-party dependency that
 * is not in /root/reference (`transformers`, unpinned in the reference's requirements.txt:11; 5.15.0 in the build image): the
 * oracle (oracle/bert_oracle.py) restates BertModel's published forward pass and is pinned by goldens that the REAL
 * transformers.BertModel produced (oracle/gen_bert_golden.py).  The Japanese / English extractors (text/japanese_bert.py:34-43,
 * english_bert_mock.py:30-41) are DeBERTa-v2 models (`DebertaV2ForMaskedLM` / `DebertaV2Model`, disentangled attention): the same
 * entry points run them with bv2_bert_config.arch = BV2_BERT_ARCH_DEBERTA_V2 (oracle/deberta_oracle.py, goldens of the real classes).
 *
 * Same conventions as bv2.h: plain pointers and sizes, every device buffer caller-allocated, launches only on the caller's stream,
 * no allocation, no host sync (capturable), int status + bv2_bert_last_error.  Output layout is the one the TextEncoder front of
 * bv2_encode_durations consumes at WORD level (bv2_encode_in.bert + bert_index): fp32 [B][hidden][S], S contiguous — the hidden
 * state never leaves the device and is never transposed or repeated (chinese_bert.py:37 `.cpu()`, :48-58 repeat loop, :60 `.T`).
 */
#ifndef BV2_BERT_H
#define BV2_BERT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bv2_bert bv2_bert;

typedef struct bv2_bert_config {
  int32_t struct_bytes;          /* sizeof(bv2_bert_config) */
  int32_t vocab_size;            /* BertConfig.vocab_size (21128) */
  int32_t hidden_size;           /* 1024; a multiple of 128, <= 1024 */
  int32_t num_heads;             /* 16; hidden_size / num_heads in {32, 64, 96, 128} */
  int32_t intermediate_size;     /* 4096 */
  int32_t max_position;          /* 512 */
  int32_t type_vocab_size;       /* 2 */
  int32_t num_layers_run;        /* encoder layers to execute: num_hidden_layers - 2 = 22 gives hidden_states[-3] */
  float layer_norm_eps;          /* 1e-12 (BERT), 1e-7 (DeBERTa-v2) */
  /* ---- architecture.  BV2_BERT_ARCH_BERT: the fields above are everything.  BV2_BERT_ARCH_DEBERTA_V2 (the reference's Japanese /
   * English extractors, text/japanese_bert.py:9, text/english_bert_mock.py:9; configs under /root/reference/bert/): word embeddings
   * only (position_biased_input = false, type_vocab_size = 0: max_position / type_vocab_size above are then the relative-position
   * table length and ignored), disentangled attention with relative positions (relative_attention, share_att_key, pos_att_type
   * c2p|p2c, norm_rel_ebd layer_norm), optional ConvLayer after the first encoder layer. */
  int32_t arch;
  int32_t att_span;              /* DeBERTa: position_buckets (256), or max_relative_positions when buckets are off */
  int32_t conv_kernel_size;      /* DeBERTa: ConvLayer kernel size (3 for deberta-v2-large-japanese-char-wwm), 0 = none; conv_act gelu */
} bv2_bert_config;
#define BV2_BERT_ARCH_BERT 0
#define BV2_BERT_ARCH_DEBERTA_V2 1

int bv2_bert_create(const bv2_bert_config* cfg, bv2_bert** out);
void bv2_bert_destroy(bv2_bert* h);
const char* bv2_bert_last_error(const bv2_bert* h);

/* Size of the packed weight blob (bytes; a pure function of the config). */
int64_t bv2_bert_packed_bytes(const bv2_bert* h);

/* Pack ONE tensor of the HuggingFace state_dict (fp32, HOST memory, PyTorch layout) into the HOST blob: GEMM weights go to the MFMA
 * fragment order of the conv kernels, query/key/value are fused into one projection (1/sqrt(head_dim) folded into the query rows).
 * Keys as in BertModel.state_dict() ("embeddings.word_embeddings.weight", "encoder.layer.3.attention.self.query.bias", ...); a
 * leading "bert." (BertForMaskedLM, what the reference loads) is accepted.  Returns 0 = packed, 1 = key not used by this path
 * (pooler, cls head, position_ids, layers >= num_layers_run), < 0 = error (unknown layout / shape mismatch). */
int bv2_bert_pack_tensor(bv2_bert* h, void* host_blob, int64_t blob_bytes, const char* hf_key, const float* data,
                         const int64_t* shape, int ndim);
/* DeBERTa-v2 additionally needs three tensors that are functions of the weights only — the caller (bert_encoder.py) derives them
 * once from the state_dict and packs them like any other tensor:
 *   "encoder.layer.N.attention.self.pos_key"    fp32 [2*att_span][hidden] = key_proj(LayerNorm(encoder.rel_embeddings.weight))
 *   "encoder.layer.N.attention.self.pos_query"  fp32 [2*att_span][hidden] = query_proj(LayerNorm(encoder.rel_embeddings.weight))
 *   "encoder.relative_index"                    fp32 [2*max_position - 1]: clamp(bucket(r) + att_span, 0, 2*att_span - 1) for
 *                                               r = -(max_position-1) .. max_position-1 (make_log_bucket_position)
 * (1/sqrt(3*head_dim) is folded into the query rows and into pos_query here, not by the caller.) */
/* Number of tensors the blob still misses (0 = complete); names via bv2_bert_last_error when > 0. */
int bv2_bert_missing(bv2_bert* h);

/* Kernel-selection switches for A/B measurements.  "prefetch" (default 3; batch 1 only): bit 0 — the embedding / LayerNorm launches, bit 1 —
 * the GEMM launches carry the NEXT GEMM's packed weight stream, which spare workgroups touch into the L2 of the XCD that will read it, so
 * a 12-17 MB weight set is on chip when its GEMM starts (bv2_kernels.h Prefetch).  0: every GEMM fetches its weights when it starts. */
int bv2_bert_set_option(bv2_bert* h, const char* key, int value);

/* Attach the blob after the caller copied it to DEVICE memory (the library never allocates). */
int bv2_bert_attach_weights(bv2_bert* h, const void* dev_blob, int64_t bytes);

int64_t bv2_bert_workspace_bytes(const bv2_bert* h, int B, int S);

/* hidden_states[num_layers_run] of BertModel(input_ids, token_type_ids, attention_mask) for B sentences padded to S tokens.
 *   input_ids      int64 [B][S] (DEVICE)           token_type_ids  int64 [B][S] or NULL (= all 0, what the tokenizer emits here)
 *   lengths        int64 [B] or NULL (= all S): attention_mask[b][s] = s < lengths[b]; padded keys get BertModel's additive
 *                  mask semantics (their probability is exactly 0), padded query columns are computed but meaningless
 *   out            fp32 [B][hidden][S] (DEVICE)
 * `stream` is a hipStream_t.  Workspace: bv2_bert_workspace_bytes(B, S) bytes of DEVICE memory. */
int bv2_bert_forward(bv2_bert* h, void* stream, int B, int S, const int64_t* input_ids, const int64_t* token_type_ids,
                     const int64_t* lengths, float* out, void* workspace, int64_t workspace_bytes);

#ifdef __cplusplus
}
#endif
#endif

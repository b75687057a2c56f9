// bv2_bert.cpp — the BERT feature extractor behind include/bv2_bert.h: HuggingFace BertModel's forward pass up to
// hidden_states[num_layers_run] (reference call site text/chinese_bert.py:34-37; algorithm: transformers' modeling_bert —
// BertEmbeddings, BertSelfAttention, BertSelfOutput, BertIntermediate, BertOutput), as a fixed launch sequence of libbv2's kernels:
//
//   embeddings      kernels/bert.hip  bert_embed_ln     word + token-type + position rows, LayerNorm            1 launch
//   per layer       conv_mfma.hip     split-K GEMM      fused query/key/value projection (1/sqrt(d) folded)      \
//                   attention.hip     attention         softmax(q k^T + mask) v, flash-style (window 0)           |
//                   conv_mfma.hip     split-K GEMM      attention.output.dense + bias + residual -> partial slabs | 7 launches
//                   kernels/bert.hip  bert_ln           slab sum + LayerNorm                                      |
//                   conv_mfma.hip     split-K GEMM      intermediate.dense + erf-GELU                             |
//                   conv_mfma.hip     split-K GEMM      output.dense + bias + residual -> partial slabs           |
//                   kernels/bert.hip  bert_ln           slab sum + LayerNorm                                     /
//
// A sentence is S ~ 20-100 tokens, so every GEMM has a few dozen columns: the layer is bound by streaming its 12.6 M fp32 weights
// (50 MB) from HBM, which is what the split-K kernel is built for (weights global -> registers, K split over waves and
// workgroups, no barrier in the main loop).  Activations fp32 [B][C][S]; the output is the word-level feature matrix the
// TextEncoder front gathers from (bv2_encode_in.bert + bert_index).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <set>
#include <string>

#include "../../include/bv2_bert.h"
#include "bv2_internal.h"

using namespace bv2;

namespace {

constexpr int64_t kBertMagic = 0x42563242455254ll;      // "BV2BERT"
constexpr int kBertLayout = 1;

struct Lin { int cin = 0, cin_pad = 0, cout = 0, cout_pad = 0, w_ld = 0, k = 1; int64_t w_off = -1, b_off = -1; };
struct LayerW { Lin qkv, o, ffn1, ffn2; int64_t g1 = -1, b1 = -1, g2 = -1, b2 = -1, pk = -1, pq = -1; };   // pk / pq: DeBERTa [H][D][2 span]

inline int rup(int x, int m) { return (x + m - 1) / m * m; }

struct Carve {
  char* base; int64_t off = 0, cap;
  Carve(void* b, int64_t c) : base(static_cast<char*>(b)), cap(c) {}
  float* get(int64_t n) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += (n * 4 + 255) / 256 * 256;
    return p;
  }
  bool ok() const { return !base || off <= cap; }
};

}  // namespace

struct bv2_bert {
  bv2_bert_config cfg;
  std::string err;
  const float* blob = nullptr;
  int64_t word = 0, pos = 0, type = 0, emb_g = 0, emb_b = 0, erv = 0, total = 0;
  int64_t tab = -1, conv_g = -1, conv_b = -1;           // DeBERTa-v2: relative index table, ConvLayer LayerNorm
  Lin conv;                                             // DeBERTa-v2 ConvLayer (k = conv_kernel_size)
  std::vector<LayerW> layer;
  int prefetch = 3;                                     // bv2_bert_set_option("prefetch"): bit 0 LayerNorm / embedding launches, bit 1 GEMM launches carry the next GEMM's weights
  bool deberta() const { return cfg.arch == BV2_BERT_ARCH_DEBERTA_V2; }
  std::set<std::string> packed, wanted;
  int D() const { return cfg.hidden_size / cfg.num_heads; }
  // BERT: + one (all-zero) relative-key row per head (attention.hip with window 0); DeBERTa: its own attention kernel, q/k/v only
  int qkv_rows() const { return 3 * cfg.hidden_size + (deberta() ? 0 : cfg.num_heads); }
};

static thread_local std::string g_bert_create_err;

static void lay_lin(Lin& l, int cin, int cout, int64_t& off, int k = 1) {
  l.cin = cin; l.cin_pad = rup(cin, 16); l.cout = cout; l.cout_pad = rup(cout, 32); l.w_ld = rup(cout, 128); l.k = k;
  l.w_off = off; off += (int64_t)k * l.cin_pad * l.w_ld;
  l.b_off = off; off += l.w_ld;
}

static void build_layout(bv2_bert* h) {
  const bv2_bert_config& c = h->cfg;
  const int C = c.hidden_size, I = c.intermediate_size;
  const bool deb = h->deberta();
  int64_t off = 64;                                  // header: 16 int32 = 64 bytes = 16 floats; keep 64 floats for alignment
  auto vec = [&](int64_t n) { const int64_t o = off; off += (n + 63) / 64 * 64; return o; };
  h->word = vec((int64_t)c.vocab_size * C);
  h->wanted = {"embeddings.word_embeddings.weight", "embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias"};
  if (deb) {
    h->pos = h->type = -1;
    h->tab = vec(2 * (int64_t)c.max_position - 1);
    h->wanted.insert("encoder.relative_index");
    if (c.conv_kernel_size > 0) {
      lay_lin(h->conv, C, C, off, c.conv_kernel_size);
      h->conv_g = vec(C); h->conv_b = vec(C);
      for (const char* k : {"encoder.conv.conv.weight", "encoder.conv.conv.bias", "encoder.conv.LayerNorm.weight", "encoder.conv.LayerNorm.bias"})
        h->wanted.insert(k);
    }
  } else {
    h->pos = vec((int64_t)c.max_position * C);
    h->type = vec((int64_t)c.type_vocab_size * C);
    h->wanted.insert("embeddings.position_embeddings.weight");
    h->wanted.insert("embeddings.token_type_embeddings.weight");
  }
  h->emb_g = vec(C); h->emb_b = vec(C);
  h->erv = vec(h->D());
  h->layer.resize(c.num_layers_run);
  for (int i = 0; i < c.num_layers_run; ++i) {
    LayerW& L = h->layer[i];
    lay_lin(L.qkv, C, h->qkv_rows(), off);
    if (deb) { L.pk = vec((int64_t)C * 2 * c.att_span); L.pq = vec((int64_t)C * 2 * c.att_span); }
    lay_lin(L.o, C, C, off);
    L.g1 = vec(C); L.b1 = vec(C);
    lay_lin(L.ffn1, C, I, off);
    lay_lin(L.ffn2, I, C, off);
    L.g2 = vec(C); L.b2 = vec(C);
    const std::string p = "encoder.layer." + std::to_string(i) + ".";
    const char* qn = deb ? "attention.self.query_proj" : "attention.self.query";
    const char* kn = deb ? "attention.self.key_proj" : "attention.self.key";
    const char* vn = deb ? "attention.self.value_proj" : "attention.self.value";
    for (const char* s : {qn, kn, vn, "attention.output.dense", "attention.output.LayerNorm", "intermediate.dense", "output.dense",
                          "output.LayerNorm"}) {
      h->wanted.insert(p + s + ".weight");
      h->wanted.insert(p + s + ".bias");
    }
    if (deb) { h->wanted.insert(p + "attention.self.pos_key"); h->wanted.insert(p + "attention.self.pos_query"); }
  }
  h->total = off;
}

static uint32_t cfg_hash(const bv2_bert_config& c) {
  const int32_t v[11] = {c.vocab_size, c.hidden_size, c.num_heads, c.intermediate_size, c.max_position, c.type_vocab_size,
                         c.num_layers_run, kBertLayout, c.arch, c.att_span, c.conv_kernel_size};
  uint32_t hsh = 2166136261u;
  for (int i = 0; i < 11; ++i) { hsh ^= (uint32_t)v[i]; hsh *= 16777619u; }
  return hsh;
}

extern "C" {

int bv2_bert_create(const bv2_bert_config* cfg, bv2_bert** out) {
  if (!cfg || !out) { g_bert_create_err = "bv2_bert_create: null argument"; return -1; }
  if (cfg->struct_bytes != (int32_t)sizeof(bv2_bert_config)) { g_bert_create_err = "bv2_bert_create: struct_bytes mismatch"; return -1; }
  const int C = cfg->hidden_size, H = cfg->num_heads;
  if (C < 128 || C > 1024 || C % 128 || H < 1 || C % H || (C / H) % 32 || C / H > 128 || cfg->intermediate_size < 16 ||
      cfg->intermediate_size % 16 || cfg->vocab_size < 1 || cfg->max_position < 1 ||
      (cfg->arch == BV2_BERT_ARCH_BERT && cfg->type_vocab_size < 1) ||
      cfg->num_layers_run < 1 || !(cfg->layer_norm_eps > 0.f) ||
      (cfg->arch != BV2_BERT_ARCH_BERT && cfg->arch != BV2_BERT_ARCH_DEBERTA_V2) ||
      (cfg->arch == BV2_BERT_ARCH_DEBERTA_V2 && (cfg->att_span < 1 || cfg->att_span > 4096 || cfg->conv_kernel_size < 0 ||
                                                 (cfg->conv_kernel_size > 0 && cfg->conv_kernel_size % 2 == 0) || cfg->conv_kernel_size > 15))) {
    g_bert_create_err = "bv2_bert_create: unsupported config (hidden a multiple of 128 up to 1024, head_dim in {32,64,96,128})";
    return -2;
  }
  try {
    bv2_bert* h = new bv2_bert();
    h->cfg = *cfg;
    build_layout(h);
    *out = h;
    return 0;
  } catch (...) { g_bert_create_err = "bv2_bert_create: out of memory"; return -100; }
}

void bv2_bert_destroy(bv2_bert* h) { delete h; }

const char* bv2_bert_last_error(const bv2_bert* h) { return h ? h->err.c_str() : g_bert_create_err.c_str(); }

int64_t bv2_bert_packed_bytes(const bv2_bert* h) { return h ? h->total * 4 : -1; }

// place W [cout_src][cin] (PyTorch Linear) at output rows [row0, row0 + cout_src) of a fused projection, fragment order
static void put_linear(float* blob, const Lin& l, int row0, const float* w, int rows, float scale) {
  for (int co = 0; co < rows; ++co)
    for (int ci = 0; ci < l.cin; ++ci)
      blob[l.w_off + conv_w_index(0, ci, row0 + co, l.cin_pad, 1)] = w[(size_t)co * l.cin + ci] * scale;
}

int bv2_bert_pack_tensor(bv2_bert* h, void* host_blob, int64_t blob_bytes, const char* hf_key, const float* data,
                         const int64_t* shape, int ndim) {
  if (!h) return -1;
  try {
    if (!host_blob || !hf_key || !data || !shape || blob_bytes < h->total * 4) { h->err = "bv2_bert_pack_tensor: bad argument"; return -1; }
    std::string k = hf_key;
    if (k.rfind("bert.", 0) == 0) k = k.substr(5);
    else if (k.rfind("deberta.", 0) == 0) k = k.substr(8);
    if (!h->wanted.count(k)) return 1;
    float* blob = static_cast<float*>(host_blob);
    int64_t* hdr = reinterpret_cast<int64_t*>(blob);
    hdr[0] = kBertMagic; hdr[1] = kBertLayout; hdr[2] = cfg_hash(h->cfg); hdr[3] = h->total;
    const bv2_bert_config& c = h->cfg;
    const int C = c.hidden_size, I = c.intermediate_size;
    auto is2 = [&](int64_t a, int64_t b) { return ndim == 2 && shape[0] == a && shape[1] == b; };
    auto is1 = [&](int64_t a) { return ndim == 1 && shape[0] == a; };
    auto bad = [&]() { h->err = "bv2_bert_pack_tensor: shape mismatch for '" + k + "'"; return -3; };
    auto copy = [&](int64_t off, int64_t n) { std::memcpy(blob + off, data, sizeof(float) * (size_t)n); };
    if (k == "embeddings.word_embeddings.weight") { if (!is2(c.vocab_size, C)) return bad(); copy(h->word, (int64_t)c.vocab_size * C); }
    else if (k == "embeddings.position_embeddings.weight") { if (h->pos < 0 || !is2(c.max_position, C)) return bad(); copy(h->pos, (int64_t)c.max_position * C); }
    else if (k == "embeddings.token_type_embeddings.weight") { if (h->type < 0 || !is2(c.type_vocab_size, C)) return bad(); copy(h->type, (int64_t)c.type_vocab_size * C); }
    else if (k == "embeddings.LayerNorm.weight") { if (!is1(C)) return bad(); copy(h->emb_g, C); }
    else if (k == "embeddings.LayerNorm.bias") { if (!is1(C)) return bad(); copy(h->emb_b, C); }
    else if (k == "encoder.relative_index") {
      const int64_t n = 2 * (int64_t)c.max_position - 1;
      if (!is1(n)) return bad();
      // kernels/deberta_attn.hip stages at most 63 consecutive table rows per 32x32 tile pair: t(r) must be monotone with slope <= 1
      // (true for every log-bucket table with (max_relative_positions-1)/(position_buckets/2) >= e, i.e. the reference's models;
      // the device-side clamp is only a backstop and would silently return wrong scores for a steeper table)
      for (int64_t i = 1; i < n; ++i) {
        const float d = data[i] - data[i - 1];
        if (!(d >= 0.f && d <= 1.f)) {
          h->err = "encoder.relative_index: the relative-position bucket table must be non-decreasing with steps <= 1 "
                   "(position_buckets too large for max_relative_positions); unsupported by the disentangled-attention kernel";
          return -2;
        }
      }
      copy(h->tab, n);
    }
    else if (k == "encoder.conv.conv.weight") {
      const int kk = c.conv_kernel_size;
      if (!(ndim == 3 && shape[0] == C && shape[1] == C && shape[2] == kk)) return bad();
      for (int co = 0; co < C; ++co)
        for (int ci = 0; ci < C; ++ci)
          for (int j = 0; j < kk; ++j)
            blob[h->conv.w_off + conv_w_index(j, ci, co, h->conv.cin_pad, kk)] = data[((size_t)co * C + ci) * kk + j];
    }
    else if (k == "encoder.conv.conv.bias") { if (!is1(C)) return bad(); copy(h->conv.b_off, C); }
    else if (k == "encoder.conv.LayerNorm.weight") { if (!is1(C)) return bad(); copy(h->conv_g, C); }
    else if (k == "encoder.conv.LayerNorm.bias") { if (!is1(C)) return bad(); copy(h->conv_b, C); }
    else {
      int li = -1, consumed = 0;
      if (std::sscanf(k.c_str(), "encoder.layer.%d.%n", &li, &consumed) != 1 || li < 0 || li >= c.num_layers_run) return 1;
      const std::string rest = k.substr(consumed);
      LayerW& L = h->layer[li];
      // BERT: scores / sqrt(d); DeBERTa-v2: (QK + c2p + p2c) / sqrt(3 d) — folded into the query rows (QK and c2p) and into pos_query (p2c)
      const float qs = 1.0f / std::sqrt((float)h->D() * (h->deberta() ? 3.0f : 1.0f));
      auto lin = [&](const Lin& l, int row0, int rows, int cin, float scale, bool is_w) {
        if (is_w) { if (!is2(rows, cin)) return bad(); put_linear(blob, l, row0, data, rows, scale); }
        else { if (!is1(rows)) return bad(); for (int r = 0; r < rows; ++r) blob[l.b_off + row0 + r] = data[r] * scale; }
        return 0;
      };
      if (rest == "attention.self.pos_key" || rest == "attention.self.pos_query") {
        // [2 span][C] -> per head transposed [H][D][2 span]: a run of relative indices is a contiguous row (kernels/deberta_attn.hip)
        const int R2 = 2 * c.att_span, Dh = h->D();
        if (!is2(R2, C)) return bad();
        const bool isq = rest == "attention.self.pos_query";
        float* dst = blob + (isq ? L.pq : L.pk);
        for (int r = 0; r < R2; ++r)
          for (int ch = 0; ch < C; ++ch)
            dst[((size_t)(ch / Dh) * Dh + ch % Dh) * R2 + r] = data[(size_t)r * C + ch] * (isq ? qs : 1.f);
        h->packed.insert(k);
        return 0;
      }
      const bool w = rest.size() > 7 && rest.compare(rest.size() - 7, 7, ".weight") == 0;
      const std::string mod = rest.substr(0, rest.rfind('.'));
      int rc = 0;
      if (mod == "attention.self.query" || mod == "attention.self.query_proj") rc = lin(L.qkv, 0, C, C, qs, w);
      else if (mod == "attention.self.key" || mod == "attention.self.key_proj") rc = lin(L.qkv, C, C, C, 1.f, w);
      else if (mod == "attention.self.value" || mod == "attention.self.value_proj") rc = lin(L.qkv, 2 * C, C, C, 1.f, w);
      else if (mod == "attention.output.dense") rc = lin(L.o, 0, C, C, 1.f, w);
      else if (mod == "intermediate.dense") rc = lin(L.ffn1, 0, I, C, 1.f, w);
      else if (mod == "output.dense") rc = lin(L.ffn2, 0, C, I, 1.f, w);
      else if (mod == "attention.output.LayerNorm") { if (!is1(C)) return bad(); copy(w ? L.g1 : L.b1, C); }
      else if (mod == "output.LayerNorm") { if (!is1(C)) return bad(); copy(w ? L.g2 : L.b2, C); }
      else return 1;
      if (rc) return rc;
    }
    h->packed.insert(k);
    return 0;
  } catch (const std::exception& e) { h->err = std::string("exception: ") + e.what(); return -100; }
}

int bv2_bert_missing(bv2_bert* h) {
  if (!h) return -1;
  int n = 0;
  std::string names;
  for (const std::string& k : h->wanted)
    if (!h->packed.count(k)) { if (n < 8) names += (n ? ", " : "") + k; ++n; }
  if (n) h->err = "missing tensors: " + names + (n > 8 ? ", ..." : "");
  return n;
}

int bv2_bert_attach_weights(bv2_bert* h, const void* dev_blob, int64_t bytes) {
  if (!h) return -1;
  if (!dev_blob) { h->blob = nullptr; return 0; }
  if (bytes < h->total * 4) { h->err = "bv2_bert_attach_weights: blob too small"; return -1; }
  int64_t hdr[4];
  if (hipMemcpy(hdr, dev_blob, sizeof(hdr), hipMemcpyDeviceToHost) != hipSuccess) { h->err = "bv2_bert_attach_weights: cannot read the blob header"; return -6; }
  if (hdr[0] != kBertMagic || hdr[1] != kBertLayout || hdr[2] != (int64_t)cfg_hash(h->cfg) || hdr[3] != h->total) {
    h->err = "bv2_bert_attach_weights: blob was packed for another config / layout";
    return -4;
  }
  h->blob = static_cast<const float*>(dev_blob);
  return 0;
}

struct BertPlan { float *x, *x1, *att, *qkv, *s, *f1, *mask, *emb; int64_t slab; int ld; };
static BertPlan plan(const bv2_bert* h, Carve& A, int B, int S) {
  const bv2_bert_config& c = h->cfg;
  BertPlan p;
  const int64_t C = c.hidden_size, BS = (int64_t)B * S;
  p.ld = rup(S, 32);
  p.slab = BS * C;
  p.x = A.get(BS * C);
  p.x1 = A.get(BS * C);
  p.att = A.get(BS * C);
  p.qkv = A.get((int64_t)B * h->qkv_rows() * p.ld);
  p.s = A.get(BV2_MAX_KSPLIT * BS * C);
  p.f1 = A.get(BS * c.intermediate_size);
  p.mask = A.get(BS);
  p.emb = (h->deberta() && c.conv_kernel_size > 0) ? A.get(BS * C) : nullptr;     // the ConvLayer reads the embeddings after layer 0
  return p;
}

int bv2_bert_set_option(bv2_bert* h, const char* key, int value) {
  if (!h || !key) return -1;
  if (std::string(key) == "prefetch") { h->prefetch = value & 3; return 0; }
  h->err = std::string("bv2_bert_set_option: unknown key ") + key;
  return -1;
}

int64_t bv2_bert_workspace_bytes(const bv2_bert* h, int B, int S) {
  if (!h || B < 1 || S < 1) return -1;
  Carve A(nullptr, 0);
  (void)plan(h, A, B, S);
  return A.off;
}

int bv2_bert_forward(bv2_bert* h, void* stream, int B, int S, const int64_t* input_ids, const int64_t* token_type_ids,
                     const int64_t* lengths, float* out, void* workspace, int64_t workspace_bytes) {
  if (!h) return -1;
  try {
    const bv2_bert_config& c = h->cfg;
    if (!h->blob) { h->err = "bv2_bert_forward: no weights attached"; return -2; }
    if (B < 1 || S < 1 || S > c.max_position || !input_ids || !out || !workspace) { h->err = "bv2_bert_forward: bad argument (S <= max_position)"; return -1; }
    Carve A(workspace, workspace_bytes);
    const BertPlan P = plan(h, A, B, S);
    if (!A.ok()) { h->err = "workspace too small for bv2_bert_forward"; return -5; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float* W = h->blob;
    const int C = c.hidden_size, I = c.intermediate_size;
    int rc = 0;
    auto chk = [&](int r, const char* what) { if (r && !rc) { rc = r; h->err = std::string("kernel launch failed: ") + what; } };

    const bool deb = h->deberta();
    if (deb && S > c.max_position) { h->err = "bv2_bert_forward: S exceeds the relative-position table (max_position)"; return -1; }
    // the packed stream of a GEMM's weights as a prefetch target (bv2_kernels.h Prefetch): batch 1 only
    auto pf_of = [&](const Lin* l, int bit) -> Prefetch {
      if (!l || B != 1 || !(h->prefetch & bit) || l->w_off < 0) return Prefetch{nullptr, 0};
      return Prefetch{W + l->w_off, (unsigned)((int64_t)(l->cout_pad / 32) * (l->cin_pad / 8) * l->k * 1024)};
    };
    float* x_in = P.emb ? P.emb : P.x;               // layer 0 reads the embeddings from here (kept for the DeBERTa ConvLayer)
    BertEmbedArgs e;
    e.input_ids = input_ids; e.token_type_ids = token_type_ids;
    e.word = W + h->word; e.pos = h->pos >= 0 ? W + h->pos : nullptr; e.type = h->type >= 0 ? W + h->type : nullptr;
    e.lengths = deb ? lengths : nullptr;             // DebertaV2Embeddings multiplies by the mask, BertEmbeddings does not
    e.gamma = W + h->emb_g; e.beta = W + h->emb_b; e.eps = c.layer_norm_eps;
    e.out = x_in; e.B = B; e.S = S; e.C = C; e.vocab = c.vocab_size; e.max_pos = c.max_position; e.type_vocab = c.type_vocab_size;
    if (c.num_layers_run > 0) e.pf = pf_of(&h->layer[0].qkv, 1);
    chk(launch_bert_embed_ln(s, e), "bert.embeddings");
    chk(launch_seq_mask(s, lengths, P.mask, B, S), "bert.mask");

    // y = W x + b as a k = 1 conv on [B][cin][S]; slabs > 1: K split across workgroups, the LayerNorm sums the partial slabs
    auto gemm = [&](const Lin& l, const float* x, float* y, int act, const float* res, bool slabs, int out_rs, int64_t out_bs,
                    const char* what, const float* out_mask = nullptr, const Lin* next = nullptr) -> int {
      ConvLaunch cl;
      std::memset(&cl, 0, sizeof(cl));
      ConvProb& p = cl.p[0];
      p.x[0] = x; p.nsrc = 1; p.in_scale = 1.f;
      p.x_bstride = (int64_t)l.cin * S; p.x_rstride = S; p.Lin = S;
      p.in_mask_bstride = S; p.out_mask_bstride = S;
      p.w = W + l.w_off; p.bias = W + l.b_off;
      p.out = y; p.out_bstride = out_bs; p.out_rstride = out_rs; p.out_tstride = 1; p.out_toff = 0;
      p.res = res; p.res_bstride = out_bs; p.res_mode = res ? RES_ADD : RES_NONE;
      p.cin = l.cin; p.cin_pad = l.cin_pad; p.cout = l.cout; p.cout_pad = l.cout_pad; p.w_ld = l.w_ld;
      p.k = l.k; p.dil = 1; p.pad_left = (l.k - 1) / 2; p.slope = 0.1f; p.act = act;
      p.out_mask = out_mask; p.mask_pre = out_mask ? 1 : 0;          // (act(Wx + b)) * mask, then + residual
      cl.nprob = 1; cl.B = B; cl.L = S; cl.ksplit = 1; cl.slab_stride = P.slab;
      cl.pf = pf_of(next, 2);
      if (slabs && conv_use_splitk(cl)) cl.ksplit = conv_pick_ksplit(cl, 4);   // 4 slabs: measured 1.77 ms per forward against 1.91 (8) and 1.85 (2) at B = 1, S = 53
      chk(launch_conv1d(s, cl, TILE_AUTO, nullptr), what);
      return cl.ksplit;
    };
    auto ln = [&](const float* a, int nslab, int64_t g, int64_t b, float* y, const char* what, const float* mask = nullptr,
                  const Lin* next = nullptr) {
      BertLnArgs l;
      l.pf = pf_of(next, 1);
      l.a = a; l.nslab = nslab; l.slab_stride = P.slab; l.gamma = W + g; l.beta = W + b; l.eps = c.layer_norm_eps; l.mask = mask;
      l.out = y; l.B = B; l.C = C; l.T = S;
      chk(launch_bert_ln(s, l), what);
    };

    const int R = h->qkv_rows();
    for (int i = 0; i < c.num_layers_run && !rc; ++i) {
      const LayerW& L = h->layer[i];
      const float* xin = i == 0 ? x_in : P.x;
      const bool last = i + 1 == c.num_layers_run;
      const bool conv_here = deb && i == 0 && c.conv_kernel_size > 0;
      const Lin* next_qkv = (!last && !conv_here) ? &h->layer[i + 1].qkv : nullptr;
      gemm(L.qkv, xin, P.qkv, ACT_NONE, nullptr, false, P.ld, (int64_t)R * P.ld, "bert.qkv", nullptr, &L.o);
      if (deb) {
        DebertaAttnArgs a;
        a.qkv = P.qkv; a.ld = P.ld; a.mask = P.mask; a.pk = W + L.pk; a.pq = W + L.pq; a.tab = W + h->tab; a.out = P.att;
        a.B = B; a.H = c.num_heads; a.D = h->D(); a.T = S; a.P = c.max_position; a.span = c.att_span;
        chk(launch_deberta_attn(s, a), "deberta.attention");
      } else {
        AttnArgs a;
        a.qkv = P.qkv; a.ld = P.ld; a.mask = P.mask; a.erv = W + h->erv; a.out = P.att;
        a.B = B; a.H = c.num_heads; a.D = h->D(); a.T = S; a.W = 0; a.f16 = 0;
        chk(launch_attention(s, a), "bert.attention");
      }
      int ns = gemm(L.o, P.att, P.s, ACT_NONE, xin, true, S, (int64_t)C * S, "bert.attention.output");
      ln(P.s, ns, L.g1, L.b1, P.x1, "bert.attention.output.LayerNorm", nullptr, &L.ffn1);
      gemm(L.ffn1, P.x1, P.f1, ACT_GELU, nullptr, false, S, (int64_t)I * S, "bert.intermediate", nullptr, &L.ffn2);
      ns = gemm(L.ffn2, P.f1, P.s, ACT_NONE, P.x1, true, S, (int64_t)C * S, "bert.output");
      ln(P.s, ns, L.g2, L.b2, (last && !conv_here) ? out : P.x, "bert.output.LayerNorm", nullptr, next_qkv);
      if (conv_here) {
        // DebertaV2Encoder.forward: after layer 0, output = ConvLayer(embeddings, layer-0 output, mask)
        //   = LayerNorm(layer0 + gelu(conv1d_k(embeddings)) [masked]) * mask     (conv_act = gelu; masked BEFORE the activation in HF:
        //   gelu(0) = 0, so (gelu(.)) * mask is the same tensor)
        gemm(h->conv, P.emb, P.s, ACT_GELU, P.x, false, S, (int64_t)C * S, "deberta.conv", P.mask);
        ln(P.s, 1, h->conv_g, h->conv_b, last ? out : P.x, "deberta.conv.LayerNorm", P.mask);
      }
    }
    return rc;
  } catch (const std::exception& e) { h->err = std::string("exception: ") + e.what(); return -100; }
}

}  // extern "C"

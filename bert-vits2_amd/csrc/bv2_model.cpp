// bv2_model.cpp — packed-weight layout and the host-side packer.
//
// What the reference does on EVERY forward and we do ONCE at load:
//   * weight_norm fold  w = g * v / ||v||  over all dims but 0 (dim 0 = C_out for Conv1d, C_in for ConvTranspose1d;
//     reference models.py:513, modules.py:160-182, 226-292) — also accepts already folded `.weight` checkpoints
//     (Generator.remove_weight_norm, models.py:559-564);
// and what only a from-scratch layout can do:
//   * conv weights re-laid as [C_out/32][C_in/8][tap][C_in&1][C_out%32][(C_in%8)/2] = the MFMA A-operand fragment order,
//     one contiguous stream per 32-row output tile (bv2_kernels.h conv_w_index);
//   * conv_q/k/v fused into one projection that also emits, per head, the 2W+1 relative-key logits
//     q_i·Ek[r]/sqrt(d) (they are linear in the layer input: rows Ek·Wq/sqrt(d)); 1/sqrt(d) folded into the q rows;
//   * ConvTranspose1d split into its u polyphase stride-1 convolutions (k/u taps each);
//   * the flows' channel Flip (modules.py:374-381) folded into input/output channel permutations of pre/post,
//     so no flip ever touches HBM;
//   * enc_p.proj split into its m / logs halves (models.py:397-399).
// The layout is a pure function of bv2_config, so every rank derives identical offsets and the blob can be broadcast.
#include <cmath>
#include <cstring>
#include <functional>

#include "bv2_internal.h"

namespace bv2 {

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// fp32 -> bf16, round to nearest even (what torch's .to(torch.bfloat16) and v_cvt_pk_bf16_f32 do)
static inline uint16_t f2bf(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// fp32 -> fp16, round to nearest even (what torch's .to(torch.float16) and v_cvt_f16_f32 do)
static inline uint16_t f2h(float f) {
  const _Float16 h = (_Float16)f;
  uint16_t u;
  std::memcpy(&u, &h, 2);
  return u;
}

static inline float h2f(uint16_t u) {
  _Float16 h;
  std::memcpy(&h, &u, 2);
  return (float)h;
}

namespace {

struct Packer {
  const std::map<std::string, HostTensor>* T = nullptr;   // null => layout only
  float* blob = nullptr;
  int64_t cursor = kBlobHeaderFloats;
  std::vector<std::string> missing;
  bool emit_bf16 = false;                                  // also write the channels-last bf16 fragment stream (dec.*)
  bool emit_f16 = false;                                   // also write the fp16 fragment stream (flow Encoder convs)
  bool emit_x6 = false;                                    // also write the three-plane bf16 split of the fp32 weights (conv_x6.hip)

  bool fill() const { return blob != nullptr; }

  int64_t alloc(int64_t n) {
    const int64_t off = cursor;
    cursor += (n + 63) / 64 * 64;                          // 256-byte alignment
    if (fill()) std::memset(blob + off, 0, sizeof(float) * (size_t)((n + 63) / 64 * 64));
    return off;
  }

  const HostTensor* get(const std::string& key, std::initializer_list<int64_t> shape) {
    if (!fill()) return nullptr;
    auto it = T->find(key);
    if (it == T->end()) { missing.push_back(key); return nullptr; }
    std::vector<int64_t> want(shape);
    if (it->second.shape != want) {
      std::string s = key + " (shape mismatch: got [";
      for (auto d : it->second.shape) s += std::to_string(d) + ",";
      s += "] want [";
      for (auto d : want) s += std::to_string(d) + ",";
      s += "])";
      missing.push_back(s);
      return nullptr;
    }
    return &it->second;
  }
  bool has(const std::string& key) const { return fill() && T->count(key); }

  // folded weight [d0][d1][k] of a (possibly weight-normed) conv; norm over all dims but 0
  bool folded(const std::string& p, int d0, int d1, int k, std::vector<float>& w) {
    if (!fill()) return false;
    if (has(p + ".weight_v") || !has(p + ".weight")) {
      const HostTensor* v = get(p + ".weight_v", {d0, d1, k});
      const HostTensor* g = get(p + ".weight_g", {d0, 1, 1});
      if (!v || !g) return false;
      w.resize(v->data.size());
      const int64_t inner = (int64_t)d1 * k;
      for (int a = 0; a < d0; ++a) {
        double ss = 0;
        for (int64_t i = 0; i < inner; ++i) { const double x = v->data[a * inner + i]; ss += x * x; }
        // torch computes the norm in fp32; a double sqrt rounded to fp32 agrees to <= 1 ulp
        const float nrm = (float)std::sqrt(ss);
        const float sc = g->data[a] / nrm;
        for (int64_t i = 0; i < inner; ++i) w[a * inner + i] = v->data[a * inner + i] * sc;
      }
      return true;
    }
    const HostTensor* t = get(p + ".weight", {d0, d1, k});
    if (!t) return false;
    w = t->data;
    return true;
  }

  VecW vec(const std::string& key, std::initializer_list<int64_t> shape) {
    int64_t n = 1;
    for (auto d : shape) n *= d;
    VecW v; v.n = n; v.off = alloc(n);
    if (const HostTensor* t = get(key, shape)) std::memcpy(blob + v.off, t->data.data(), sizeof(float) * (size_t)n);
    return v;
  }

  // generic conv packer: src(co, ci, j) gives the folded weight, bsrc(co) the bias
  ConvW conv(int cout, int cin, int k, bool bias, const std::function<float(int, int, int)>& src,
             const std::function<float(int)>& bsrc, bool ok) {
    ConvW c;
    c.cin = cin; c.cout = cout; c.k = k;
    c.cin_pad = round_up(cin, 16); c.cout_pad = round_up(cout, 32); c.w_ld = round_up(cout, 128);
    c.w_off = alloc((int64_t)k * c.cin_pad * c.w_ld);
    c.b_off = bias ? alloc(c.cout_pad) : -1;
    if (emit_bf16 && cin % 16 == 0) c.wb_off = alloc((cl_w_elems(cin, c.cout_pad, k) + 1) / 2);
    if (emit_f16 && cin % 16 == 0) c.wh_off = alloc((cl_w_elems(cin, c.cout_pad, k) + 1) / 2);
    if (emit_x6 && cin % 16 == 0) c.wx_off = alloc((x6_w_elems(cin, c.cout_pad, k) + 1) / 2);
    // ... and the two scaled fp16 planes of the "x3" form of conv_x6.hip / respair_x6.hip (1 / S_w in front)
    if (emit_x6 && cin % 16 == 0) c.wy_off = alloc(X3_HDR_FLOATS + (x3_w_elems(cin, c.cout_pad, k) + 1) / 2);
    if (fill() && ok) {
      for (int j = 0; j < k; ++j)
        for (int ci = 0; ci < cin; ++ci)
          for (int co = 0; co < cout; ++co)
            blob[c.w_off + conv_w_index(j, ci, co, c.cin_pad, k)] = src(co, ci, j);
      if (bias)
        for (int co = 0; co < cout; ++co) blob[c.b_off + co] = bsrc(co);
      if (c.wb_off >= 0) {
        uint16_t* wb = reinterpret_cast<uint16_t*>(blob + c.wb_off);
        for (int j = 0; j < k; ++j)
          for (int ci = 0; ci < cin; ++ci)
            for (int co = 0; co < cout; ++co) wb[cl_w_index(j, ci, co, cin, k)] = f2bf(src(co, ci, j));
      }
      if (c.wx_off >= 0) {
        // w = h1 + h2 + h3 exactly, one bf16 plane each (x6_split / x6_w_index, bv2_kernels.h)
        uint16_t* wx = reinterpret_cast<uint16_t*>(blob + c.wx_off);
        for (int j = 0; j < k; ++j)
          for (int ci = 0; ci < cin; ++ci)
            for (int co = 0; co < cout; ++co) {
              uint16_t h[3];
              x6_split(src(co, ci, j), h);
              for (int pl = 0; pl < 3; ++pl) wx[x6_w_index(j, ci, co, cin, k, pl)] = h[pl];
            }
      }
      if (c.wy_off >= 0) {
        // w * S_w = g0 + g1 (x3_w_index, bv2_kernels.h); S_w from the tensor's largest magnitude
        float wmax = 0.f;
        for (int j = 0; j < k; ++j)
          for (int ci = 0; ci < cin; ++ci)
            for (int co = 0; co < cout; ++co) wmax = std::max(wmax, std::fabs(src(co, ci, j)));
        uint32_t mb;
        std::memcpy(&mb, &wmax, 4);
        const unsigned e = x3_scale_exp(mb);
        const float S = x3_scale(e);
        blob[c.wy_off] = x3_scale_inv(e);
        uint16_t* wy = reinterpret_cast<uint16_t*>(blob + c.wy_off + X3_HDR_FLOATS);
        for (int j = 0; j < k; ++j)
          for (int ci = 0; ci < cin; ++ci)
            for (int co = 0; co < cout; ++co) {
              const float v = src(co, ci, j) * S;                    // exact: S is a power of two, |v| < 2^15
              const uint16_t g0 = f2h(v);
              wy[x3_w_index(j, ci, co, cin, k, 0)] = g0;
              wy[x3_w_index(j, ci, co, cin, k, 1)] = f2h(v - h2f(g0));
            }
      }
      if (c.wh_off >= 0) {
        uint16_t* wh = reinterpret_cast<uint16_t*>(blob + c.wh_off);
        for (int j = 0; j < k; ++j)
          for (int ci = 0; ci < cin; ++ci)
            for (int co = 0; co < cout; ++co) wh[cl_w_index(j, ci, co, cin, k)] = f2h(src(co, ci, j));
      }
    }
    return c;
  }

  // bf16-only conv (no fp32 copy): weights as a channels-last fragment stream + fp32 bias
  ConvW conv_cl(int cout, int cin, int k, const std::function<float(int, int, int)>& src,
                const std::function<float(int)>& bsrc, bool ok) {
    ConvW c;
    c.cin = cin; c.cout = cout; c.k = k;
    c.cin_pad = cin; c.cout_pad = round_up(cout, 32); c.w_ld = c.cout_pad;
    c.b_off = alloc(c.cout_pad);
    c.wb_off = alloc((cl_w_elems(cin, c.cout_pad, k) + 1) / 2);
    if (fill() && ok) {
      uint16_t* wb = reinterpret_cast<uint16_t*>(blob + c.wb_off);
      for (int j = 0; j < k; ++j)
        for (int ci = 0; ci < cin; ++ci)
          for (int co = 0; co < cout; ++co) wb[cl_w_index(j, ci, co, cin, k)] = f2bf(src(co, ci, j));
      for (int co = 0; co < cout; ++co) blob[c.b_off + co] = bsrc(co);
    }
    return c;
  }

  // plain nn.Conv1d `p` ([cout][cin][k] + bias), optional row range / channel permutations
  ConvW conv1d(const std::string& p, int cout, int cin, int k, bool bias = true, bool wn = false, int row0 = 0,
               int rows = -1, bool rev_in = false, bool rev_out = false, const std::function<int(int)>* rowmap = nullptr) {
    if (rows < 0) rows = cout;
    std::vector<float> w;
    const HostTensor* wt = nullptr;
    bool ok = true;
    if (fill()) {
      if (wn) ok = folded(p, cout, cin, k, w);
      else { wt = get(p + ".weight", {cout, cin, k}); ok = wt != nullptr; }
    }
    const float* wd = wn ? w.data() : (wt ? wt->data.data() : nullptr);
    const HostTensor* bt = bias ? get(p + ".bias", {cout}) : nullptr;
    if (bias && fill() && !bt) ok = false;
    auto src = [&](int co, int ci, int j) {
      const int sco = rowmap ? (*rowmap)(co) : row0 + (rev_out ? rows - 1 - co : co);
      const int sci = rev_in ? cin - 1 - ci : ci;
      return wd[((int64_t)sco * cin + sci) * k + j];
    };
    auto bsrc = [&](int co) { return bt->data[rowmap ? (*rowmap)(co) : row0 + (rev_out ? rows - 1 - co : co)]; };
    return conv(rows, cin, k, bias, src, bsrc, ok);
  }

  GemvW gemv(const std::string& p, int cout, int cin, bool conv_shape, bool wn = false,
             const std::function<int(int)>* rowmap = nullptr) {      // rowmap: packed row r is source row (*rowmap)(r)
    GemvW g; g.cout = cout; g.cin = cin;
    g.w_off = alloc((int64_t)cout * cin);
    g.b_off = alloc(cout);
    if (fill()) {
      std::vector<float> w;
      const float* wd = nullptr;
      if (wn) { if (folded(p, cout, cin, 1, w)) wd = w.data(); }
      else if (conv_shape) { if (auto* t = get(p + ".weight", {cout, cin, 1})) wd = t->data.data(); }
      else { if (auto* t = get(p + ".weight", {cout, cin})) wd = t->data.data(); }
      const HostTensor* b = get(p + ".bias", {cout});
      for (int r = 0; r < cout; ++r) {
        const int sr = rowmap ? (*rowmap)(r) : r;
        if (wd) std::memcpy(blob + g.w_off + (int64_t)r * cin, wd + (int64_t)sr * cin, sizeof(float) * (size_t)cin);
        if (b) blob[g.b_off + r] = b->data[sr];
      }
    }
    return g;
  }
};

EncoderW pack_encoder(Packer& P, const std::string& p, int hidden, int filter, int heads, int layers, int ksize, int gin) {
  EncoderW e;
  e.n_layers = layers; e.ksize = ksize; e.hidden = hidden; e.filter = filter; e.heads = heads;
  const int dk = hidden / heads, nr = 2 * kAttnWindow + 1;
  e.spk = P.gemv(p + ".spk_emb_linear", hidden, gin, /*conv_shape=*/false);
  for (int i = 0; i < layers; ++i) {
    EncLayerW& L = e.layer[i];
    const std::string a = p + ".attn_layers." + std::to_string(i);
    // fused q/k/v projection (reference attentions.py:264-266 runs three 1x1 convs) + relative-key logit rows
    const HostTensor *wq = P.get(a + ".conv_q.weight", {hidden, hidden, 1}), *wk = P.get(a + ".conv_k.weight", {hidden, hidden, 1}),
                     *wv = P.get(a + ".conv_v.weight", {hidden, hidden, 1});
    const HostTensor *bq = P.get(a + ".conv_q.bias", {hidden}), *bk = P.get(a + ".conv_k.bias", {hidden}),
                     *bv = P.get(a + ".conv_v.bias", {hidden});
    const HostTensor* ek = P.get(a + ".emb_rel_k", {1, nr, dk});
    const bool ok = wq && wk && wv && bq && bk && bv && ek;
    const double isq = 1.0 / std::sqrt((double)dk);        // query / sqrt(k_channels), attentions.py:280
    auto wsrc = [&](int co, int ci, int) -> float {
      if (co < hidden) return (float)(wq->data[(int64_t)co * hidden + ci] * isq);
      if (co < 2 * hidden) return wk->data[(int64_t)(co - hidden) * hidden + ci];
      if (co < 3 * hidden) return wv->data[(int64_t)(co - 2 * hidden) * hidden + ci];
      const int hh = (co - 3 * hidden) / nr, r = (co - 3 * hidden) % nr;
      double acc = 0;
      for (int c = 0; c < dk; ++c) acc += (double)ek->data[r * dk + c] * wq->data[(int64_t)(hh * dk + c) * hidden + ci];
      return (float)(acc * isq);
    };
    auto bsrc = [&](int co) -> float {
      if (co < hidden) return (float)(bq->data[co] * isq);
      if (co < 2 * hidden) return bk->data[co - hidden];
      if (co < 3 * hidden) return bv->data[co - 2 * hidden];
      const int hh = (co - 3 * hidden) / nr, r = (co - 3 * hidden) % nr;
      double acc = 0;
      for (int c = 0; c < dk; ++c) acc += (double)ek->data[r * dk + c] * bq->data[hh * dk + c];
      return (float)(acc * isq);
    };
    L.qkv = P.conv(3 * hidden + heads * nr, hidden, 1, true, wsrc, bsrc, ok);
    L.o = P.conv1d(a + ".conv_o", hidden, hidden, 1);
    L.erv = P.vec(a + ".emb_rel_v", {1, nr, dk});
    L.g1 = P.vec(p + ".norm_layers_1." + std::to_string(i) + ".gamma", {hidden});
    L.b1 = P.vec(p + ".norm_layers_1." + std::to_string(i) + ".beta", {hidden});
    const std::string f = p + ".ffn_layers." + std::to_string(i);
    // (Round 5, measured and not kept: these two convs also as the three split-bf16 planes, so that at batch 1 their split-K launches
    // run on the bf16 matrix core — config 2 3.611 / 3.596 -> 3.584 / 3.588 ms for +195 MB of blob: a split-K launch at batch 1 is 10k
    // ticks of prologue (first bytes of x and of the weights), 10k of K loop and 2k of epilogue behind a ~3 us launch gap, and only the
    // loop gets shorter; profiles/r05_ab_splitk_x6_not_kept.txt.  The kernel was deleted in round 6.)
    L.ffn1 = P.conv1d(f + ".conv_1", filter, hidden, ksize);
    L.ffn2 = P.conv1d(f + ".conv_2", hidden, filter, ksize);
    L.g2 = P.vec(p + ".norm_layers_2." + std::to_string(i) + ".gamma", {hidden});
    L.b2 = P.vec(p + ".norm_layers_2." + std::to_string(i) + ".beta", {hidden});
  }
  return e;
}

DDSW pack_dds(Packer& P, const std::string& p, int c) {
  DDSW d;
  int dil = 1;
  for (int i = 0; i < kSdpLayers; ++i) {
    DDSLayerW& L = d.l[i];
    const std::string si = std::to_string(i);
    L.dil = dil; dil *= kSdpKernel;
    L.dww = P.vec(p + ".convs_sep." + si + ".weight", {c, 1, kSdpKernel});
    L.dwb = P.vec(p + ".convs_sep." + si + ".bias", {c});
    L.c1x1 = P.conv1d(p + ".convs_1x1." + si, c, c, 1);
    L.g1 = P.vec(p + ".norms_1." + si + ".gamma", {c});
    L.b1 = P.vec(p + ".norms_1." + si + ".beta", {c});
    L.g2 = P.vec(p + ".norms_2." + si + ".gamma", {c});
    L.b2 = P.vec(p + ".norms_2." + si + ".beta", {c});
  }
  return d;
}

uint32_t hash_cfg(const bv2_config& c) {
  const unsigned char* p = reinterpret_cast<const unsigned char*>(&c);
  uint32_t h = 2166136261u;
  for (size_t i = 0; i < sizeof(c); ++i) { h ^= p[i]; h *= 16777619u; }
  return h;
}

int pack_all(Model& m, Packer& P) {
  const bv2_config& c = m.cfg;
  const int hid = c.hidden_channels, inter = c.inter_channels, filt = c.filter_channels, gin = c.gin_channels;
  const int half = inter / 2;

  // ---- enc_p (reference models.py:333-400)
  m.emb = P.vec("enc_p.emb.weight", {c.n_vocab, hid});
  m.tone_emb = P.vec("enc_p.tone_emb.weight", {c.n_tones, hid});
  m.lang_emb = P.vec("enc_p.language_emb.weight", {c.n_languages, hid});
  const char* bn[3] = {"enc_p.bert_proj", "enc_p.ja_bert_proj", "enc_p.en_bert_proj"};
  // (Round 5, measured and not kept: these convs also as split-bf16 planes so that at batch >= 16, where they leave the split-K regime,
  // they take conv_x6.hip — config 3 15.12 / 15.08 -> 15.06 / 15.08 ms, nothing, for +42 MB of blob: N = 128 columns per batch item gives
  // the 64x128 / 128x64 tiles one or two column tiles per item; profiles/r05_ab_x6_enc_not_kept.txt.)
  for (int i = 0; i < 3; ++i) m.bert[i] = P.conv1d(bn[i], hid, c.bert_dim, 1);
  m.enc = pack_encoder(P, "enc_p.encoder", hid, filt, c.n_heads, c.n_layers, c.kernel_size, gin);
  m.proj_m = P.conv1d("enc_p.proj", 2 * inter, hid, 1, true, false, 0, inter);
  m.proj_logs = P.conv1d("enc_p.proj", 2 * inter, hid, 1, true, false, inter, inter);

  // ---- sdp (reference models.py:148-204, 245-256)
  m.sdp_pre = P.conv1d("sdp.pre", hid, hid, 1);
  m.sdp_proj = P.conv1d("sdp.proj", hid, hid, 1);
  m.sdp_cond = P.gemv("sdp.cond", hid, gin, true);
  m.sdp_convs = pack_dds(P, "sdp.convs", hid);
  const int cf_idx[kSdpFlowsUsed] = {7, 5, 3};      // reversed(flows) minus the "useless vflow" (models.py:246-247)
  for (int i = 0; i < kSdpFlowsUsed; ++i) {
    const std::string p = "sdp.flows." + std::to_string(cf_idx[i]);
    m.cf[i].pre_w = P.vec(p + ".pre.weight", {hid, 1, 1});
    m.cf[i].pre_b = P.vec(p + ".pre.bias", {hid});
    m.cf[i].convs = pack_dds(P, p + ".convs", hid);
    m.cf[i].proj = P.conv1d(p + ".proj", 3 * kSdpBins - 1, hid, 1);
  }
  m.ea_m = P.vec("sdp.flows.0.m", {2, 1});
  m.ea_logs = P.vec("sdp.flows.0.logs", {2, 1});

  // ---- dp (reference models.py:259-299)
  m.dp_cond = P.gemv("dp.cond", hid, gin, true);
  m.dp_c1 = P.conv1d("dp.conv_1", kDpFilter, hid, kDpKernel);
  m.dp_g1 = P.vec("dp.norm_1.gamma", {kDpFilter});
  m.dp_b1 = P.vec("dp.norm_1.beta", {kDpFilter});
  m.dp_c2 = P.conv1d("dp.conv_2", kDpFilter, kDpFilter, kDpKernel);
  m.dp_g2 = P.vec("dp.norm_2.gamma", {kDpFilter});
  m.dp_b2 = P.vec("dp.norm_2.beta", {kDpFilter});
  m.dp_proj = P.conv1d("dp.proj", 1, kDpFilter, 1);

  m.emb_g = P.vec("emb_g.weight", {c.n_speakers, gin});

  // ---- flow, in APPLICATION order of the reverse pass (reference models.py:143-144 / 443-444)
  m.n_coupling = c.use_transformer_flow ? c.n_flow_layer : 4;
  for (int a = 0; a < m.n_coupling; ++a) {
    CouplingW& C = m.coupling[a];
    const int f = m.n_coupling - 1 - a;
    const std::string p = "flow.flows." + std::to_string(2 * f);
    // a+1 Flips have been applied before coupling a (reverse pass: Flip, C_{n-1}, Flip, ..., C_0).  They are folded into pre's input /
    // post's output channel order instead of moving data — which leaves the tensor un-flipped at the end only if their number n is even.
    // Odd n: the FIRST Flip is real data movement (flow_core, one launch per pass), the remaining n - 1 are folded
    m.flow_flip_first = (m.n_coupling % 2) == 1;
    C.flipped = m.flow_flip_first ? (a % 2) == 1 : (a % 2) == 0;
    C.pre = P.conv1d(p + ".pre", hid, half, 1, true, false, 0, -1, /*rev_in=*/C.flipped, false);
    if (c.use_transformer_flow) {
      P.emit_f16 = true;                           // BASELINE config 5 "fp16 flow": q/k/v/o and FFN convs also as fp16 streams
      C.enc = pack_encoder(P, p + ".enc", hid, filt, c.n_heads, c.n_layers_trans_flow, kFlowKernel, gin);
      P.emit_f16 = false;
    } else {
      const int nl = c.n_flow_layer;
      C.wn_layers = nl;
      // WN gate (commons.py:98-105: tanh(first half) * sigmoid(second half)) fused into in_layer's epilogue (ACT_GATE): rows are
      // packed so that every 32-row tile holds the tanh rows of 16 channels followed by their sigmoid rows; the conditioning
      // slice g_l (cond_layer rows [2H*i, 2H*(i+1)), modules.py:189-197) is permuted the same way
      const std::function<int(int)> gate_row = [hid](int n) { const int mt = n / 32, j = n % 32; return j < 16 ? 16 * mt + j : hid + 16 * mt + (j - 16); };
      const std::function<int(int)> cond_row = [hid, &gate_row](int n) { return (n / (2 * hid)) * 2 * hid + gate_row(n % (2 * hid)); };
      C.wn_cond = P.gemv(p + ".enc.cond_layer", 2 * hid * nl, gin, true, /*wn=*/true, &cond_row);
      P.emit_f16 = true;                           // "fp16 flow": in_layers / res_skip_layers also as fp16 streams (same row order)
      for (int i = 0; i < nl; ++i) {
        const std::string rsn = p + ".enc.res_skip_layers." + std::to_string(i);
        C.wn_in[i] = P.conv1d(p + ".enc.in_layers." + std::to_string(i), 2 * hid, hid, kFlowKernel, true, true, 0, -1, false, false, &gate_row);
        // res_skip (modules.py:203-210): rows [0,H) update x, rows [H,2H) accumulate into the output; the last layer has H rows,
        // all of them output.  Two problems of ONE launch, each with its own residual target (no separate res/skip kernel).
        if (i < nl - 1) {
          C.wn_res[i] = P.conv1d(rsn, 2 * hid, hid, 1, true, true, 0, hid);
          C.wn_skip[i] = P.conv1d(rsn, 2 * hid, hid, 1, true, true, hid, hid);
        } else {
          C.wn_skip[i] = P.conv1d(rsn, hid, hid, 1, true, true);
        }
      }
      P.emit_f16 = false;
    }
    C.post = P.conv1d(p + ".post", half, hid, 1, true, false, 0, -1, false, /*rev_out=*/C.flipped);
  }

  // ---- dec (reference models.py:490-564)
  const int c0 = c.upsample_initial_channel;
  P.emit_bf16 = true;
  m.conv_pre = P.conv1d("dec.conv_pre", c0, inter, 7);
  m.dec_cond = P.gemv("dec.cond", c0, gin, true);
  m.n_ups = c.n_upsamples; m.n_rbk = c.n_resblock_kernels; m.n_rbd = c.n_resblock_dilations; m.rb_type = c.resblock_type == 2 ? 2 : 1;
  m.total_up = 1;
  int ch = c0;
  for (int i = 0; i < m.n_ups; ++i) {
    UpW& U = m.ups[i];
    U.u = c.upsample_rates[i]; U.k = c.upsample_kernel_sizes[i];
    U.cin = c0 >> i; U.cout = c0 >> (i + 1);
    U.ntaps = U.k / U.u;
    const int pad = (U.k - U.u) / 2;
    m.total_up *= U.u;
    std::vector<float> w;                          // folded ConvTranspose1d weight [cin][cout][k]
    const bool ok = P.folded("dec.ups." + std::to_string(i), U.cin, U.cout, U.k, w);
    const HostTensor* bt = P.get("dec.ups." + std::to_string(i) + ".bias", {U.cout});
    P.emit_bf16 = false;                           // the bf16 path uses the single channels-last form below, not the phases
    for (int ph = 0; ph < U.u; ++ph) {
      // output n = u*s + ph gathers x[s + shift - mtap] * W[ci][co][pp + u*mtap]   
      const int pp = (ph + pad) % U.u, shift = (ph + pad) / U.u;
      U.pad_left[ph] = (U.ntaps - 1) - shift;
      const int nt = U.ntaps, uu = U.u, kk = U.k, cout = U.cout;
      U.phase[ph] = P.conv(U.cout, U.cin, nt, true,
                           [&, pp, nt, uu, kk, cout](int co, int ci, int j) {
                             return w[((int64_t)ci * cout + co) * kk + pp + uu * (nt - 1 - j)];
                           },
                           [&](int co) { return bt->data[co]; }, ok && bt);
    }
    P.emit_bf16 = true;
    {
      // channels-last form: tap window = union over phases; phase ph uses window taps [off, off + ntaps), off = plmax - pad_left[ph]
      int plmax = 0, right = 0;
      for (int ph = 0; ph < U.u; ++ph) {
        plmax = U.pad_left[ph] > plmax ? U.pad_left[ph] : plmax;
        const int r = U.ntaps - 1 - U.pad_left[ph];
        right = r > right ? r : right;
      }
      const int kk = plmax + right + 1, nt = U.ntaps, uu = U.u, kfull = U.k, cout = U.cout, pad_ = pad;
      U.cl_pad_left = plmax;
      const int* pl = U.pad_left;
      U.cl = P.conv_cl(U.u * U.cout, U.cin, kk,
                       [&, plmax, nt, uu, kfull, cout, pad_, pl](int cop, int ci, int jw) -> float {
                         const int ph = cop / cout, co = cop % cout;
                         const int j = jw - (plmax - pl[ph]);
                         if (j < 0 || j >= nt) return 0.f;
                         const int pp = (ph + pad_) % uu;
                         return w[((int64_t)ci * cout + co) * kfull + pp + uu * (nt - 1 - j)];
                       },
                       [&, cout](int cop) { return bt->data[cop % cout]; }, ok && bt);
    }
    ch = U.cout;
    for (int j = 0; j < m.n_rbk; ++j) {
      const int k = c.resblock_kernel_sizes[j];
      const std::string rp = "dec.resblocks." + std::to_string(i * m.n_rbk + j);
      // wide stages (the ones the LDS-tiled fp32 conv runs): the weights also as the three bf16 planes of conv_x6.hip
      P.emit_x6 = ch >= 16 && ch % 16 == 0;        // C = 16: the pair kernel only (respair_x6.hip; conv_x6.hip wants whole 32-channel chunks)
      const bool rb2 = c.resblock_type == 2;       // modules.ResBlock2: one weight-normed conv per dilation, `convs.<d>`
      for (int d = 0; d < m.n_rbd; ++d) {
        if (rb2) {
          m.rb[i][j][d][0] = P.conv1d(rp + ".convs." + std::to_string(d), ch, ch, k, true, true);
          m.rb[i][j][d][1] = ConvW();
        } else {
          m.rb[i][j][d][0] = P.conv1d(rp + ".convs1." + std::to_string(d), ch, ch, k, true, true);
          m.rb[i][j][d][1] = P.conv1d(rp + ".convs2." + std::to_string(d), ch, ch, k, true, true);
        }
      }
      P.emit_x6 = false;
      if (rb2) {                                   // the fused / whole-ResBlock streams describe (conv, conv) pairs: ResBlock1 only
        m.rbcl_w_off[i][j] = m.rbcl_b_off[i][j] = m.rb16_w_off[i][j] = m.rb16_b_off[i][j] = -1;
        continue;
      }
      // narrow stages: the 2*n_rbd convs of the block once more as ONE contiguous bf16 stream (+ one bias block) for the
      // whole-ResBlock kernel; copied out of the per-conv streams just written (m-tile 0 = the whole channel dim)
      m.rbcl_w_off[i][j] = m.rbcl_b_off[i][j] = -1;
      if (m.n_rbd <= BV2_RBCL_MAX_D && resblock_cl_bf16_supported(ch, k, c.resblock_dilation_sizes[j], m.n_rbd)) {
        const int U = (ch / 16) * k, Upad = resblock_cl_bf16_units(ch, k);
        m.rbcl_w_off[i][j] = P.alloc(((int64_t)(2 * m.n_rbd * Upad + RBCL_PD) * 512 + 1) / 2);
        m.rbcl_b_off[i][j] = P.alloc((int64_t)2 * m.n_rbd * 32);
        if (P.fill()) {
          uint16_t* dst = reinterpret_cast<uint16_t*>(P.blob + m.rbcl_w_off[i][j]);
          for (int d = 0; d < m.n_rbd; ++d)
            for (int e = 0; e < 2; ++e) {
              const ConvW& cw = m.rb[i][j][d][e];
              // TAP-MAJOR unit order (tap j outer, 16-channel group s inner): the kernel's LDS walk then needs one pointer add
              // per tap and immediate offsets for the groups; the per-conv stream is group-major (unit = s*k + j)
              const uint16_t* src = reinterpret_cast<const uint16_t*>(P.blob + cw.wb_off);
              const int G = ch / 16;
              for (int j2 = 0; j2 < k; ++j2)
                for (int s2 = 0; s2 < G; ++s2)
                  std::memcpy(dst + ((int64_t)(2 * d + e) * Upad + (int64_t)j2 * G + s2) * 512, src + ((int64_t)s2 * k + j2) * 512,
                              sizeof(uint16_t) * 512);
              (void)U;
              std::memcpy(P.blob + m.rbcl_b_off[i][j] + (2 * d + e) * 32, P.blob + cw.b_off, sizeof(float) * (size_t)ch);
            }
        }
      }
      // C = 16: the tap-pair stream of resblock_c16_bf16.hip, rearranged from the per-conv bf16 stream just written (same bf16 values)
      m.rb16_w_off[i][j] = m.rb16_b_off[i][j] = -1;
      if (m.n_rbd <= BV2_RBCL_MAX_D && resblock_c16_bf16_supported(ch, k, c.resblock_dilation_sizes[j], m.n_rbd)) {
        const int KU = rb16_units(k);
        m.rb16_w_off[i][j] = P.alloc(((int64_t)2 * m.n_rbd * KU * 512 + 1) / 2);
        m.rb16_b_off[i][j] = P.alloc((int64_t)2 * m.n_rbd * 16);
        if (P.fill()) {
          uint16_t* dst = reinterpret_cast<uint16_t*>(P.blob + m.rb16_w_off[i][j]);
          std::memset(dst, 0, sizeof(uint16_t) * (size_t)2 * m.n_rbd * KU * 512);
          for (int d = 0; d < m.n_rbd; ++d)
            for (int e = 0; e < 2; ++e) {
              const ConvW& cw = m.rb[i][j][d][e];
              const uint16_t* src = reinterpret_cast<const uint16_t*>(P.blob + cw.wb_off);
              uint16_t* cd = dst + (int64_t)(2 * d + e) * KU * 512;
              for (int co = 0; co < 16; ++co)
                for (int ci = 0; ci < 16; ++ci)
                  for (int j2 = 0; j2 < k; ++j2) cd[rb16_w_index(j2, ci, co)] = src[cl_w_index(j2, ci, co, 16, k)];
              std::memcpy(P.blob + m.rb16_b_off[i][j] + (2 * d + e) * 16, P.blob + cw.b_off, sizeof(float) * 16);
            }
        }
      }
    }
  }
  P.emit_bf16 = false;
  m.post_c = ch;
  m.conv_post = P.vec("dec.conv_post.weight", {1, ch, m.post_k});
  P.alloc(2048);                                  // slack: weight prefetch rings run up to 4 units (4 KB) past a stream's end
  m.total_floats = P.cursor;
  return 0;
}

// The accepted hyper-parameter envelope == the tested one (bert_vits2_amd/hparams.py ENVELOPE holds the same numbers; tests/test_envelope_cpu.py
// checks both sides agree and that every bound below rejects).  Its corners are pinned on the GPU against goldens of the real reference
// (oracle/cases.ENVELOPE), its interior against the oracle (cases.random_hparams).  Round 5 accepted far more than anything ran — and an accepted,
// never-run config (odd coupling count) had been wrong for four rounds.
int validate(const bv2_config& c, std::string& err) {
  auto bad = [&](const char* s) { err = s; return -1; };
  auto in = [](int v, int lo, int hi) { return v >= lo && v <= hi; };
  if (c.struct_bytes != (int32_t)sizeof(bv2_config)) return bad("bv2_config.struct_bytes does not match this library");
  if (c.n_vocab < 1 || c.n_tones < 1 || c.n_languages < 1 || c.bert_dim < 8 || c.bert_dim % 8)
    return bad("n_vocab / n_tones / n_languages must be >= 1 and bert_dim a multiple of 8");
  if (c.n_heads < 1 || c.hidden_channels % c.n_heads) return bad("hidden_channels must be divisible by n_heads (attentions.py:223)");
  if (!in(c.hidden_channels, 96, 256) || c.hidden_channels % 32) return bad("hidden_channels must be a multiple of 32 in 96..256");
  const int dk = c.hidden_channels / c.n_heads;
  if (dk % 32 || dk > 128) return bad("head dim (hidden_channels / n_heads) must be 32, 64, 96 or 128");
  if (!in(c.filter_channels, 128, 1024) || c.filter_channels % 64) return bad("filter_channels must be a multiple of 64 in 128..1024");
  if (!in(c.inter_channels, 64, 256) || c.inter_channels % 32) return bad("inter_channels must be a multiple of 32 in 64..256");
  if (c.kernel_size != 1 && c.kernel_size != 3 && c.kernel_size != 5 && c.kernel_size != 7) return bad("FFN kernel_size must be 1, 3, 5 or 7");
  if (!in(c.n_layers, kCondLayer + 1, 8)) return bad("n_layers must be 3..8 (cond_layer_idx = 2 < n_layers, attentions.py:69-75)");
  if (c.use_transformer_flow && !in(c.n_layers_trans_flow, kCondLayer + 1, 8)) return bad("n_layers_trans_flow must be 3..8");
  if (!in(c.n_flow_layer, 1, kMaxFlows)) return bad("n_flow_layer must be 1..8");
  if (c.n_speakers < 1) return bad("n_speakers must be >= 1 (the ReferenceEncoder path, models.py:1047-1048, is out of scope)");
  if (!in(c.gin_channels, 64, 768) || c.gin_channels % 64) return bad("gin_channels must be a multiple of 64 in 64..768");
  if (!in(c.n_upsamples, 2, 5)) return bad("2..5 upsampling stages supported");
  if (!in(c.n_resblock_kernels, 1, 3)) return bad("1..3 resblock kernels supported");
  if (c.resblock_type != 1 && c.resblock_type != 2) return bad("resblock_type must be 1 (ResBlock1) or 2 (ResBlock2)");
  // modules.ResBlock1 reads dilation[0..2], ResBlock2 dilation[0..1] and ignores the rest (modules.py:208-258, 318-346): hand over exactly those
  if (c.resblock_type == 1 && c.n_resblock_dilations != 3) return bad("ResBlock1 has exactly three (dilated conv, conv) pairs: n_resblock_dilations must be 3");
  if (c.resblock_type == 2 && c.n_resblock_dilations != 2) return bad("ResBlock2 has exactly two convs (dilation[0], dilation[1]: modules.py:318-346)");
  for (int i = 0; i < c.n_upsamples; ++i) {
    const int u = c.upsample_rates[i], k = c.upsample_kernel_sizes[i];
    if (!in(u, 2, BV2_MAX_UPS)) return bad("upsample rates must be 2..8");
    if (k < u || k % u || (k - u) % 2 || k / u > 4) return bad("upsample kernel must be 1..4 x its rate with even (kernel - rate)");
  }
  // 512 = the released width; a 1024-channel channels-last bf16 ConvTranspose tile does not fit the 160 KB LDS (gen_bf16.hip conv_cl_bf16_supported)
  if (!in(c.upsample_initial_channel, 64, 512) || c.upsample_initial_channel % (1 << c.n_upsamples))
    return bad("upsample_initial_channel must be 64..512 and divisible by 2^n_upsamples");
  const int fw = c.upsample_initial_channel >> c.n_upsamples;
  if (fw != 16 && fw != 32 && fw != 64) return bad("final Generator width (upsample_initial_channel >> n_upsamples) must be 16, 32 or 64");
  for (int j = 0; j < c.n_resblock_kernels; ++j) {
    if (c.resblock_kernel_sizes[j] % 2 == 0 || !in(c.resblock_kernel_sizes[j], 3, 11)) return bad("resblock kernels must be odd, 3..11");
    for (int d = 0; d < c.n_resblock_dilations; ++d)
      if (!in(c.resblock_dilation_sizes[j][d], 1, 12)) return bad("resblock dilations must be 1..12");
  }
  return 0;
}

}  // namespace

int build_layout(Model& m, std::string& err) {
  if (int rc = validate(m.cfg, err)) return rc;
  Packer P;
  pack_all(m, P);
  m.cfg_hash = hash_cfg(m.cfg);
  return 0;
}

int pack_blob(const Model& m_in, const std::map<std::string, HostTensor>& t, float* blob, std::string& err) {
  Model m = m_in;
  Packer P;
  P.T = &t;
  P.blob = blob;
  std::memset(blob, 0, sizeof(float) * kBlobHeaderFloats);
  pack_all(m, P);
  if (!P.missing.empty()) {
    err = "missing/invalid tensors (" + std::to_string(P.missing.size()) + "): ";
    for (size_t i = 0; i < P.missing.size() && i < 12; ++i) err += P.missing[i] + "; ";
    return -2;
  }
  if (m.total_floats != m_in.total_floats) { err = "internal: layout drift between passes"; return -4; }
  uint32_t* hdr = reinterpret_cast<uint32_t*>(blob);
  hdr[0] = kBlobMagic; hdr[1] = BV2_ABI_VERSION; hdr[2] = m_in.cfg_hash; hdr[3] = BV2_PACK_LAYOUT;
  int64_t tf = m.total_floats;
  std::memcpy(hdr + 4, &tf, sizeof(tf));
  return 0;
}

bool key_in_schema(const Model& m, const std::string& key) {
  (void)m;
  static const char* ignored[] = {"enc_q.", "sdp.post_", "sdp.flows.1."};
  for (const char* p : ignored)
    if (key.compare(0, std::strlen(p), p) == 0) return false;
  return true;
}

}  // namespace bv2

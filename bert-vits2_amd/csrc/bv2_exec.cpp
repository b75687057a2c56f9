// bv2_exec.cpp — the host executor: SynthesizerTrn.infer() (reference models.py:1026-1074) as a fixed sequence of
// kernel launches on the caller's stream.  No allocation, no host sync, no device->host copy happens in here
// (the one data-dependent size, T_y, is read by the CALLER between run_encode and run_decode — the same single
// sync the reference has at commons.py:120-122), so each phase is hipGraph-capturable.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "bv2_internal.h"

namespace bv2 {

namespace {

struct Arena {
  char* base;
  int64_t off = 0, cap;
  Arena(void* b, int64_t c) : base(static_cast<char*>(b)), cap(c) {}
  template <class T> T* get(int64_t n) {
    const int64_t bytes = (n * (int64_t)sizeof(T) + 255) / 256 * 256;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += bytes;
    return p;
  }
  bool ok() const { return base == nullptr || off <= cap; }
};

struct Ctx {
  bv2_handle* h;
  hipStream_t s;
  const Model& m;
  const float* blob;
  int rc = 0;
  const float* W(int64_t off) const { return off < 0 ? nullptr : blob + off; }

  void fail(const char* what, int code) {
    if (!rc) { rc = code ? code : -1; h->err = std::string("kernel launch failed: ") + what; }
  }

  // ---- profiling: HIP events on the caller's stream around one kernel family launch
  const char* cur_tag = "";
  std::string cur_shape;
  int prof_begin(const char* tag) {
    cur_tag = tag;
    if (!h->prof_on || h->prof_used >= h->prof_pool.size()) return -1;
    if (h->prof_mode == 2 && std::strncmp(tag, "dec.", 4) != 0) return -1;   // Generator kernels only
    if (h->prof_mode == 4 && std::strcmp(tag, "dec.ups") != 0) return -1;     // the Generator's ConvTranspose1d launches only
    const int i = (int)h->prof_used++;
    (void)hipEventRecord(h->prof_pool[i].e0, s);
    return i;
  }
  void prof_end(int i, const char* name, double flops, double bytes) {
    if (i < 0) return;
    (void)hipEventRecord(h->prof_pool[i].e1, s);
    std::string key = name;
    if (h->prof_mode >= 3) key = std::string(cur_tag) + "|" + name + cur_shape;   // one row per launch site and shape
    int fam = -1;
    for (size_t k = 0; k < h->prof_names.size(); ++k)
      if (h->prof_names[k] == key) fam = (int)k;
    if (fam < 0) { fam = (int)h->prof_names.size(); h->prof_names.push_back(key); }
    h->prof_pool[i].fam = fam; h->prof_pool[i].flops = flops; h->prof_pool[i].bytes = bytes;
  }

  void tap(const char* name, const float* src, int64_t n) {
    if (h->taps.empty()) return;
    auto it = h->taps.find(name);
    if (it == h->taps.end()) return;
    const int64_t c = n < it->second.cap ? n : it->second.cap;
    (void)hipMemcpyAsync(it->second.dst, src, sizeof(float) * (size_t)c, hipMemcpyDeviceToDevice, s);
  }

  // conv problem with the defaults of a "same"-padded Conv1d on a dense [B,C,L] tensor
  ConvProb prob(const ConvW& w, const float* x, float* out, int L, int dil = 1) const {
    ConvProb p;
    std::memset(&p, 0, sizeof(p));
    p.x[0] = x; p.nsrc = 1; p.in_scale = 1.f;
    p.x_bstride = (int64_t)w.cin * L; p.x_rstride = L; p.Lin = L;
    p.in_mask_bstride = L; p.out_mask_bstride = L;
    p.w = W(w.w_off); p.bias = W(w.b_off);
    p.out = out; p.out_bstride = (int64_t)w.cout * L; p.out_rstride = L; p.out_tstride = 1; p.out_toff = 0;
    p.res_bstride = p.out_bstride;
    p.cin = w.cin; p.cin_pad = w.cin_pad; p.cout = w.cout; p.cout_pad = w.cout_pad; p.w_ld = w.w_ld;
    p.k = w.k; p.dil = dil; p.pad_left = ((w.k - 1) / 2) * dil;
    p.slope = 0.1f;
    return p;
  }
  // Launch; returns the number of partial slabs written per output (1 unless the split-K kernel was allowed to split K
  // across workgroups: max_split > 1 means the CONSUMER sums `slab_stride`-spaced slabs).
  int conv(ConvLaunch& L, const char* tag, int max_split = 1, int64_t slab_stride = 0) {
    if (rc) return 1;
    L.ksplit = 1; L.slab_stride = slab_stride;
    if (max_split > 1 && conv_use_splitk(L)) L.ksplit = conv_pick_ksplit(L, max_split);
    const char* vn = "conv1d_mfma";
    const int pi = prof_begin(tag);
    if (pi >= 0 && h->prof_mode >= 3) {
      const ConvProb& q = L.p[0];
      cur_shape = " n" + std::to_string(L.nprob) + " " + std::to_string(q.cin) + ">" + std::to_string(q.cout) + " k" +
                  std::to_string(q.k) + " L" + std::to_string(L.L) + " B" + std::to_string(L.B) + " s" + std::to_string(L.ksplit);
    }
    const int r = launch_conv1d(s, L, TILE_AUTO, &vn);
    prof_end(pi, vn, conv_flops(L), conv_bytes(L));
    if (r) fail(tag, r);
    return L.ksplit;
  }
  int conv1(const ConvProb& p, int B, int L, const char* tag, int max_split = 1, int64_t slab_stride = 0,
            const int64_t* lens = nullptr, int len_mul = 1, Prefetch pf = Prefetch{nullptr, 0}) {
    ConvLaunch cl;
    cl.p[0] = p; cl.nprob = 1; cl.B = B; cl.L = L; cl.lens = lens; cl.len_mul = len_mul; cl.pf = pf;
    return conv(cl, tag, max_split, slab_stride);
  }
  // the packed fp32 stream of a conv (m-tile-major: what the split-K kernel's XCDs read in contiguous eighths) as a prefetch target;
  // bit 0 of the "prefetch" option: LayerNorm launches carry one, bit 1: split-K launches do.  Batch 1 only.
  Prefetch pf_of(const ConvW& w, int B, int bit) const {
    if (B != 1 || !(h->prefetch & bit) || w.w_off < 0) return Prefetch{nullptr, 0};
    return Prefetch{W(w.w_off), (unsigned)((int64_t)(w.cout_pad / 32) * (w.cin_pad / 8) * w.k * 1024)};
  }
  // fp16 Encoder conv (kernels/enc_f16.hip) on dense tensors: in_ct/out_ct select fp32 [B][C][T] vs fp16 [B][T][C]
  HcProb hprob(const ConvW& w, const void* x, bool in_ct, void* out, bool out_ct, int L) const {
    HcProb p;
    std::memset(&p, 0, sizeof(p));
    p.x = x; p.in_ct = in_ct; p.x_bstride = (int64_t)w.cin * L; p.x_rstride = L; p.Lin = L;
    p.in_mask_bstride = L; p.out_mask_bstride = L;
    p.w = reinterpret_cast<const uint16_t*>(W(w.wh_off)); p.bias = W(w.b_off);
    p.out = out; p.out_ct = out_ct; p.out_bstride = (int64_t)w.cout * L; p.out_rstride = L;
    p.res_bstride = p.out_bstride;
    p.cin = w.cin; p.cout = w.cout; p.cout_pad = w.cout_pad; p.k = w.k; p.dil = 1; p.pad_left = (w.k - 1) / 2;
    return p;
  }
  // batch item -> XCD affinity for the fp16 Encoder stacks (HcLaunch::xcd_b): every kernel of a layer keeps item b on XCD b % 8.  Only when
  // the batch fills the eight XCDs about evenly
  bool xcd_affine(int B) const { return !h->no_xcd_affine && B >= 8 && (B % 8 == 0 || B >= 32); }
  void conv_h(const HcProb& p, int B, int L, const char* tag, const HcProb* p2 = nullptr, bool xcd = false) {
    if (rc) return;
    HcLaunch hl;
    hl.p[0] = p; hl.nprob = 1; hl.B = B; hl.L = L; hl.xcd_b = xcd ? 1 : 0; hl.no_ksplit = h->no_f16_ksplit ? 1 : 0;
    hl.wn_pref = h->f16_wn; hl.ni_pref = h->f16_ni;
    if (p2) { hl.p[1] = *p2; hl.nprob = 2; }
    const char* vn = "conv_f16";
    const int pi = prof_begin(tag);
    if (pi >= 0 && h->prof_mode >= 3)
      cur_shape = " n1 " + std::to_string(p.cin) + ">" + std::to_string(p.cout) + " k" + std::to_string(p.k) + " L" +
                  std::to_string(L) + " B" + std::to_string(B);
    const int r = launch_conv_f16(s, hl, &vn);
    prof_end(pi, vn, conv_f16_flops(hl), conv_f16_bytes(hl));
    if (r) fail(tag, r);
  }
  void ln(const LnArgs& a, const char* tag) {
    if (rc) return;
    if (int r = launch_layernorm(s, a)) fail(tag, r);
  }
  void chk(int r, const char* tag) { if (r && !rc) fail(tag, r); }
};

// ---------------------------------------------------------------------------------------------------------------
// attentions.Encoder.forward (reference attentions.py:103-120): 5 launches per layer
//   qkv = W_qkv x                         (fused 1x1, MFMA)
//   att = relpos_attention(qkv)           (flash-style, MFMA)
//   s   = x + W_o att                     (MFMA, residual in the epilogue)
//   x   = LN1(s)
//   f   = relu(conv_k(x*mask))            (MFMA, input mask + ReLU fused)
//   s   = conv_k(f*mask)*mask + x         (MFMA, masks + residual fused)
//   x   = LN2(s) [ + spk, *mask when the NEXT layer is the conditioning layer; *mask after the last layer ]
struct EncBufs { float *x, *s, *att, *qkv, *f1, *ml; int64_t slab; uint16_t *k16 = nullptr, *v16 = nullptr; };   // k16 / v16: fp16 K / V of the fp16 stacks (attention.hip KV16)   // s holds kSlabs slabs of `slab` floats; ml: key-split (max, sum) pairs

constexpr int kSlabs = BV2_MAX_KSPLIT;
inline int attn_ld(int T) { return (T + 31) / 32 * 32; }
inline int qkv_rows(const EncoderW& e) { return 3 * e.hidden + e.heads * (2 * kAttnWindow + 1); }

inline int n_slabs(int B, int T);
void run_encoder(Ctx& c, const EncoderW& e, const EncBufs& b, const float* mask, const float* spk, int spk_bstride,
                 int B, int T, const char* tapname, bool f16 = false, float* out2 = nullptr, const float* vec2 = nullptr,
                 int vec2_bstride = 0, FbArgs* fb = nullptr) {
  const int H = e.hidden, ld = attn_ld(T), R = qkv_rows(e);
  const bool xcd = f16 && c.xcd_affine(B);
  // fp16 stacks: the LayerNorms that need nothing but the conv's own columns run in conv_o's / conv_2's epilogue (enc_f16.hip)
  const bool ln_in_conv = f16 && !c.h->no_f16_fused_ln && conv_f16_ln_supported(H);
  // cond_layer_idx == 2 > 0: the speaker add always rides on the previous layer's LN2 epilogue
  for (int i = 0; i < e.n_layers; ++i) {
    const EncLayerW& L = e.layer[i];
    bool ln2_in_conv = false;
    ConvProb p = c.prob(L.qkv, b.x, b.qkv, T);
    p.out_rstride = ld; p.out_bstride = (int64_t)R * ld;      // rows padded to 32 columns: aligned tile loads
    const bool kv16 = f16 && b.k16 && b.v16 && !c.h->no_f16_kv && H % 32 == 0;
    if (f16) {
      HcProb q = c.hprob(L.qkv, b.x, true, b.qkv, true, T);
      q.out_rstride = ld; q.out_bstride = (int64_t)R * ld;
      if (kv16) { q.k16 = b.k16; q.v16 = b.v16; q.kv_row0 = H; q.kv_rows = H; q.k16_ld = ld; }   // K / V rows as fp16 in the attention kernel's layouts
      c.conv_h(q, B, T, "enc.qkv", nullptr, xcd);
    } else {
      c.conv1(p, B, T, "enc.qkv");
    }
    AttnArgs a;
    a.qkv = b.qkv; a.ld = ld; a.mask = mask; a.erv = c.W(L.erv.off); a.out = b.att;
    a.B = B; a.H = e.heads; a.D = H / e.heads; a.T = T; a.W = kAttnWindow; a.f16 = f16 ? 1 : 0; a.xcd_b = xcd ? 1 : 0;
    if (kv16) { a.kh = b.k16; a.vh = b.v16; }
    // small-N regime (the one where `s` holds partial slabs): conv_o runs inside the attention kernel, head h -> slab h
    const bool fuse_o = !f16 && !c.h->no_fused_attn_o && n_slabs(B, T) >= e.heads && L.o.k == 1 && L.o.cin == H;
    // key split (batch 1, long sequences): the key tiles of a (head, query tile) go to `ks` workgroups, each writing its own partial
    // slab; LayerNorm-1 merges them with the flash-decoding weights and adds conv_o's bias and the residual itself
    int ks = 1;
    if (fuse_o && b.ml) {
      ks = c.h->attn_ksplit < 0 ? attention_pick_ksplit(B, e.heads, T, n_slabs(B, T)) : c.h->attn_ksplit;
      if (ks < 2 || e.heads * ks > n_slabs(B, T) || (e.heads * ks != 4 && e.heads * ks != 8) || (ks != 2 && ks != 4) ||
          ks > (T + 31) / 32)
        ks = 1;
    }
    if (fuse_o) {
      a.wo = c.W(L.o.w_off); a.o_out = b.s; a.o_slab_stride = b.slab;
      a.Co = L.o.cout; a.wo_groups = L.o.cin_pad / 8;
      if (ks > 1) { a.ksplit = ks; a.ml_out = b.ml; }
      else { a.bo = c.W(L.o.b_off); a.res = b.x; }
    }
    if (!c.rc) {
      const int pi = c.prof_begin("attention");
      const int r = launch_attention(c.s, a);
      c.prof_end(pi, "attention_relpos", attention_flops(a), 4.0 * B * 4 * H * (double)T);
      if (r) c.fail("attention", r);
    }
    int ns = 1;
    if (fuse_o) {
      ns = e.heads * ks;
    } else if (f16) {
      HcProb q = c.hprob(L.o, b.att, true, b.s, true, T);
      q.res = b.x; q.res_mode = RES_ADD;
      if (ln_in_conv) {                            // LayerNorm-1 in conv_o's epilogue: x = LN1(x + conv_o(att)) written in place
        q.out = b.x; q.ln_gamma = c.W(L.g1.off); q.ln_beta = c.W(L.b1.off); q.ln_eps = 1e-5f;
      }
      c.conv_h(q, B, T, "enc.o", nullptr, xcd);
    } else {
      p = c.prob(L.o, b.att, b.s, T);
      p.res = b.x; p.res_mode = RES_ADD;
      ns = c.conv1(p, B, T, "enc.o", kSlabs, b.slab);
    }
    LnArgs l;
    std::memset(&l, 0, sizeof(l));
    l.a = b.s; l.nslab = ns; l.slab_stride = b.slab;
    l.gamma = c.W(L.g1.off); l.beta = c.W(L.b1.off); l.eps = 1e-5f; l.out = b.x; l.B = B; l.C = H; l.T = T; l.xcd_b = xcd ? 1 : 0;
    if (ks > 1) { l.ml = b.ml; l.ml_H = e.heads; l.ml_ks = ks; l.bias = c.W(L.o.b_off); l.add = b.x; }
    if (!f16) l.pf = c.pf_of(L.ffn1, B, 1);                       // the FFN's first weight set lands in L2 under this LayerNorm
    if (!(f16 && ln_in_conv)) c.ln(l, "enc.ln1");
    l.ml = nullptr; l.bias = nullptr; l.add = nullptr; l.ml_H = l.ml_ks = 0; l.pf = Prefetch{nullptr, 0};
    if (f16) {
      // FFN (attentions.py:438-446): hidden activation relu(conv_1(x*mask))*mask kept as fp16 channels-last in b.f1
      HcProb q = c.hprob(L.ffn1, b.x, true, b.f1, false, T);
      q.in_mask = mask; q.act = ACT_RELU; q.out_mask = mask; q.mask_post = 1;
      c.conv_h(q, B, T, "enc.ffn1", nullptr, xcd);
      q = c.hprob(L.ffn2, b.f1, false, b.s, true, T);
      q.out_mask = mask; q.mask_pre = 1; q.res = b.x; q.res_mode = RES_ADD;
      // every LayerNorm-2 but the stack's last (masks, second output, flow_boundary.hip) rides in conv_2's epilogue the same way
      ln2_in_conv = ln_in_conv && i + 1 < e.n_layers;
      if (ln2_in_conv) {
        q.out = b.x; q.ln_gamma = c.W(L.g2.off); q.ln_beta = c.W(L.b2.off); q.ln_eps = 1e-5f;
        if (i + 1 == kCondLayer) { q.ln_vec = spk; q.ln_vec_bstride = spk_bstride; q.ln_mask = mask; }   // the next layer is the conditioning layer
      }
      c.conv_h(q, B, T, "enc.ffn2", nullptr, xcd);
      ns = 1;
    } else {
      p = c.prob(L.ffn1, b.x, b.f1, T);
      p.in_mask = mask; p.act = ACT_RELU;
      c.conv1(p, B, T, "enc.ffn1", 1, 0, nullptr, 1, c.pf_of(L.ffn2, B, 2));   // conv_2's weights under conv_1 (whose own are in L2 by now)
      p = c.prob(L.ffn2, b.f1, b.s, T);
      p.in_mask = mask; p.out_mask = mask; p.mask_pre = 1; p.res = b.x; p.res_mode = RES_ADD;
      ns = c.conv1(p, B, T, "enc.ffn2", kSlabs, b.slab);
    }
    if (ln2_in_conv) {                             // x already holds LN2's output
      if (tapname) c.tap((std::string(tapname) + ".layer." + std::to_string(i)).c_str(), b.x, (int64_t)B * H * T);
      continue;
    }
    l.a = b.s; l.nslab = ns; l.gamma = c.W(L.g2.off); l.beta = c.W(L.b2.off);
    if (!f16 && i + 1 < e.n_layers) l.pf = c.pf_of(e.layer[i + 1].qkv, B, 1);   // the next layer's q/k/v projection
    if (i + 1 == kCondLayer && i + 1 < e.n_layers) { l.vec = spk; l.vec_bstride = spk_bstride; l.mask = mask; }
    if (i + 1 == e.n_layers) {
      l.mask = mask;
      if (out2) { l.out2 = out2; l.vec2 = vec2; l.vec2_bstride = vec2_bstride; }   // (x + vec2) * mask beside x * mask
    }
    if (fb && i + 1 == e.n_layers) {
      // the flow: this LayerNorm, the coupling's post and the next coupling's pre in one launch (flow_boundary.hip)
      fb->a = l.a; fb->nslab = l.nslab; fb->slab_stride = l.slab_stride; fb->gamma = l.gamma; fb->beta = l.beta; fb->eps = l.eps;
      fb->mask = mask; fb->B = B; fb->C = H; fb->T = T;
      fb->launched = flow_boundary_supported(*fb) ? 1 : 0;
    }
    if (fb && i + 1 == e.n_layers && fb->launched) {
      if (!c.rc) {
        const int pi = c.prof_begin("flow.boundary");
        const int r = launch_flow_boundary(c.s, *fb);
        c.prof_end(pi, "flow_boundary", 2.0 * B * T * (double)H * fb->C1 * (fb->pre_w ? 2 : 1),
                   4.0 * B * T * ((double)H * (l.nslab + (fb->pre_w ? 1 : 0)) + 2.0 * fb->C1));
        if (r) c.fail("flow.boundary", r);
      }
    } else
    c.ln(l, "enc.ln2");
    if (tapname) {
      const std::string tn = std::string(tapname) + ".layer." + std::to_string(i);
      c.tap(tn.c_str(), b.x, (int64_t)B * H * T);
    }
  }
}

// modules.DDSConv.forward (reference modules.py:118-130).  Fused form (kernels/dds_fused.hip): ONE launch per layer, the
// layers ping-pong between `ha` and `hb` (a tile reads its neighbours' input columns, so a layer never updates in place), the
// first layer can take ConvFlow.pre as its input transform and the last one applies the projection that follows the DDSConv.
struct DdsPost {                      // what follows the DDSConv
  const ConvW* proj = nullptr;        // sdp.proj (post_out) or ConvFlow.proj (+ spline on z)
  float* post_out = nullptr;
  float* zio = nullptr; int z_src = 0, z_dst = 0;
};
struct DdsPre { const float* w = nullptr; const float* b = nullptr; const float* z = nullptr; int z_src = 0; const float* g = nullptr; };

void run_dds_fused(Ctx& c, const DDSW& d, const float* x_in, const DdsPre& pre, float* ha, float* hb, const float* mask,
                   const DdsPost& post, int B, int C, int T, float sqrt_fc) {
  const float* cur = x_in;
  for (int i = 0; i < kSdpLayers; ++i) {
    const DDSLayerW& L = d.l[i];
    const bool last = i + 1 == kSdpLayers;
    DdsArgs a;
    std::memset(&a, 0, sizeof(a));
    if (i == 0 && pre.w) { a.pre_w = pre.w; a.pre_b = pre.b; a.z = pre.z; a.z_src = pre.z_src; a.g = pre.g; }
    else a.x = cur;
    a.mask = mask;
    a.dww = c.W(L.dww.off); a.dwb = c.W(L.dwb.off); a.g1 = c.W(L.g1.off); a.b1 = c.W(L.b1.off);
    a.g2 = c.W(L.g2.off); a.b2 = c.W(L.b2.off);
    a.w = c.W(L.c1x1.w_off); a.bias = c.W(L.c1x1.b_off);
    float* out = (cur == ha) ? hb : ha;
    a.out = out; a.dil = L.dil; a.last_mask = last ? 1 : 0; a.eps = 1e-5f;
    a.B = B; a.C = C; a.T = T;
    if (last && post.proj) {
      a.post_w = c.W(post.proj->w_off); a.post_b = c.W(post.proj->b_off);
      a.post_cout = post.proj->cout; a.post_cout_pad = post.proj->cout_pad;
      a.post_out = post.post_out; a.zio = post.zio; a.z_src = post.z_src; a.z_dst = post.z_dst;
      a.sqrt_fc = sqrt_fc; a.tail = 5.0f;
      a.out = nullptr;                                         // only the projection's result is consumed
    }
    if (!c.rc) { if (int r = launch_dds_layer(c.s, a)) c.fail("dds.layer", r); }
    cur = out;
  }
}

// Unfused form (3 launches per layer), for channel counts the fused kernel does not cover:
//   y1 = gelu(LN1(dwconv_k3,dil(x*mask)))   y2 = W_1x1 y1 (MFMA)   x = x + gelu(LN2(y2))   [*mask after the last]
void run_dds(Ctx& c, const DDSW& d, float* x, float* y1, float* y2, int64_t slab, const float* mask, int B, int C, int T) {
  for (int i = 0; i < kSdpLayers; ++i) {
    const DDSLayerW& L = d.l[i];
    LnArgs l;
    std::memset(&l, 0, sizeof(l));
    l.a = x; l.mode = 1; l.dww = c.W(L.dww.off); l.dwb = c.W(L.dwb.off); l.dil = L.dil; l.in_mask = mask;
    l.gamma = c.W(L.g1.off); l.beta = c.W(L.b1.off); l.eps = 1e-5f; l.post_gelu = 1; l.out = y1; l.B = B; l.C = C; l.T = T;
    c.ln(l, "dds.ln1");
    ConvProb p = c.prob(L.c1x1, y1, y2, T);
    const int ns = c.conv1(p, B, T, "dds.1x1", kSlabs, slab);
    std::memset(&l, 0, sizeof(l));
    l.a = y2; l.nslab = ns; l.slab_stride = slab; l.gamma = c.W(L.g2.off); l.beta = c.W(L.b2.off); l.eps = 1e-5f; l.post_gelu = 1; l.res = x; l.out = x;
    l.B = B; l.C = C; l.T = T;
    if (i + 1 == kSdpLayers) l.mask = mask;
    c.ln(l, "dds.ln2");
  }
}

// ---------------------------------------------------------------------------------------------------------------
// slabs a split-K consumer may have to sum: only the small-N regime (conv_use_splitk) ever splits K across workgroups
inline int n_slabs(int B, int T) { return (int64_t)B * T <= 4096 ? kSlabs : 1; }

struct PlanA {
  int64_t slab;
  float *gv, *bsum, *dp0, *dp1, *dp2, *logw_dp, *sdp_h, *sdp_x, *y1, *y2, *z, *params, *logw_sdp;
  EncBufs enc;
};
PlanA plan_a(Arena& A, const Model& m, int B, int T) {
  const bv2_config& c = m.cfg;
  const int64_t H = c.hidden_channels, BT = (int64_t)B * T;
  PlanA p;
  p.gv = A.get<float>((int64_t)B * 3 * H);
  const int ns = n_slabs(B, T);
  p.slab = BT * H;
  p.bsum = A.get<float>(3 * ns * BT * H);
  p.enc.x = nullptr;   // encoder state lives in out.x
  p.enc.slab = BT * H;
  p.enc.s = A.get<float>(ns * BT * H);
  p.enc.att = A.get<float>(BT * H);
  p.enc.qkv = A.get<float>((int64_t)B * qkv_rows(m.enc) * attn_ld(T));
  p.enc.f1 = A.get<float>(BT * c.filter_channels);
  p.enc.ml = ns > 1 ? A.get<float>((int64_t)B * kSlabs * 2 * T) : nullptr;
  p.dp0 = A.get<float>(BT * H);
  p.dp1 = A.get<float>(BT * kDpFilter);
  p.dp2 = A.get<float>(BT * kDpFilter);
  p.logw_dp = A.get<float>(BT);
  p.logw_sdp = A.get<float>(BT);
  p.sdp_h = A.get<float>(BT * H);
  p.sdp_x = A.get<float>(BT * H);
  p.y1 = A.get<float>(BT * H);
  p.y2 = A.get<float>(ns * BT * H);
  p.z = A.get<float>(BT * 2);
  p.params = A.get<float>(BT * 32);
  return p;
}

constexpr int kX3Slots = 264;          // = BV2_MAX_UPS * (1 + 2 * BV2_MAX_RESBLOCK_KERNELS * BV2_MAX_RESBLOCK_DILATIONS), 1 KB each
struct PlanB {
  float *gv, *zp, *z, *h, *acts, *outacc, *pre, *ymask;
  int* fidx;
  unsigned* xslots;                  // kX3Slots max |x| slots of the fp32 Generator's x3 convs (conv_x6.hip), zeroed at the start of every decode
  int64_t* len_cap;                  // [B] the batch's longest y_length, broadcast (exact_lengths == 2)
  EncBufs enc;
  float* set[2][7];
  int gv_stride;
};
int64_t gen_unit(const Model& m) {      // max over stages of C_i * (samples per frame at stage i)
  int64_t best = 0, up = 1;
  for (int i = 0; i < m.n_ups; ++i) {
    up *= m.ups[i].u;
    const int64_t v = (int64_t)m.ups[i].cout * up;
    best = v > best ? v : best;
  }
  return best;
}
PlanB plan_b(Arena& A, const Model& m, int B, int Ty) {
  const bv2_config& c = m.cfg;
  const int64_t H = c.hidden_channels, BT = (int64_t)B * Ty;
  PlanB p;
  std::memset(&p, 0, sizeof(p));
  p.gv_stride = c.upsample_initial_channel +
                (c.use_transformer_flow ? m.n_coupling * (int)H : m.n_coupling * 2 * (int)H * c.n_flow_layer);
  p.gv = A.get<float>((int64_t)B * p.gv_stride);
  p.fidx = A.get<int>(BT);
  p.xslots = A.get<unsigned>((int64_t)kX3Slots * X3_SLOT_WORDS);
  p.ymask = A.get<float>(BT);
  p.len_cap = A.get<int64_t>(B);
  p.zp = A.get<float>(BT * c.inter_channels);
  p.z = A.get<float>(BT * c.inter_channels);
  p.h = A.get<float>(BT * H);
  if (c.use_transformer_flow) {
    p.enc.x = p.h;
    p.enc.slab = BT * H;
    p.enc.s = A.get<float>(n_slabs(B, Ty) * BT * H);
    p.enc.att = A.get<float>(BT * H);
    p.enc.qkv = A.get<float>((int64_t)B * qkv_rows(m.coupling[0].enc) * attn_ld(Ty));
    p.enc.f1 = A.get<float>(BT * c.filter_channels);
    p.enc.ml = n_slabs(B, Ty) > 1 ? A.get<float>((int64_t)B * kSlabs * 2 * Ty) : nullptr;
    p.enc.k16 = A.get<uint16_t>((int64_t)B * attn_ld(Ty) * H);
    p.enc.v16 = A.get<uint16_t>((int64_t)B * attn_ld(Ty) * H);
  } else {
    p.acts = A.get<float>(BT * H);
    p.outacc = A.get<float>(BT * H);
  }
  p.pre = A.get<float>(BT * c.upsample_initial_channel);
  const int64_t unit = gen_unit(m) * BT;
  for (int s = 0; s < 2; ++s)
    for (int i = 0; i < 7; ++i) p.set[s][i] = A.get<float>(unit);
  return p;
}

}  // namespace

int64_t workspace_bytes(const Model& m, int B, int T, int Ty) {
  Arena a(nullptr, 0), b(nullptr, 0);
  plan_a(a, m, B, T);
  plan_b(b, m, B, Ty);
  return (a.off > b.off ? a.off : b.off) + 256;
}

// ===============================================================================================================
// phase A pieces (each is also one of the reference's ONNX stages, include/bv2.h "single stages")

// the front launch: emb_g lookup (or a given g), the speaker-conditioning GEMVs, x_mask and the scaled SDP noise
static void run_front(Ctx& c, const int64_t* sid, const float* g_in, float* g_out, const GemvW* const* gw, float* const* go,
                      int n, int out_stride, const int64_t* x_lengths, float* x_mask, int B, int T, const float* noise,
                      float* z, float noise_scale) {
  const bv2_config& cf = c.m.cfg;
  FrontArgs F;
  std::memset(&F, 0, sizeof(F));
  for (int i = 0; i < n; ++i) {
    F.p[i].w = c.W(gw[i]->w_off); F.p[i].bias = c.W(gw[i]->b_off); F.p[i].out = go[i];
    F.p[i].cout = gw[i]->cout; F.p[i].cin = gw[i]->cin; F.p[i].out_bstride = out_stride;
  }
  F.nprob = n; F.B = B; F.g = g_in; F.g_bstride = cf.gin_channels; F.gin = cf.gin_channels;
  if (sid) { F.table = c.W(c.m.emb_g.off); F.sid = sid; F.nrows = cf.n_speakers; F.g_out = g_out; }
  F.lengths = x_lengths; F.mask = x_mask; F.T = T;
  F.noise = noise; F.z = z; F.noise_scale = noise_scale; F.nz = z ? (int64_t)B * 2 * T : 0;
  c.chk(launch_front(c.s, F), "front");
}

// TextEncoder (reference models.py:377-400).  dp0/dp_c: the DurationPredictor's `x + cond(g)` input rides on the last LayerNorm.
static void enc_p_core(Ctx& c, PlanA& P, const int64_t* x, const int64_t* tone, const int64_t* lang, const float* const* berts,
                       const float* mask, const float* spk, int spk_stride, int B, int T, float* out_x, float* out_m,
                       float* out_logs, float* dp0, const float* dp_c, const int32_t* const* bert_index = nullptr,
                       const int32_t* bert_cols = nullptr) {
  const Model& m = c.m;
  const bv2_config& cf = m.cfg;
  const int H = cf.hidden_channels;
  P.enc.x = out_x;
  int bert_slabs;
  {
    // the three 1024 -> hidden projections in ONE launch; each writes its own partial slab(s), the embed kernel sums them
    ConvLaunch cl;
    cl.nprob = 3; cl.B = B; cl.L = T;
    cl.ksplit = 1; cl.slab_stride = P.slab;
    for (int i = 0; i < 3; ++i) {
      cl.p[i] = c.prob(m.bert[i], berts[i], P.bsum, T);
      if (bert_index && bert_index[i]) {
        // word-level feature [B, 1024, S]: the projection runs over its S columns (reads past S are the conv's zero padding);
        // the embed kernel gathers column bert_index[b][t] of the result for symbol t
        const int S = bert_cols[i];
        cl.p[i].x_bstride = (int64_t)m.bert[i].cin * S; cl.p[i].x_rstride = S; cl.p[i].Lin = S;
      }
    }
    const int ks = conv_use_splitk(cl) ? conv_pick_ksplit(cl, kSlabs / 2) : 1;
    for (int i = 0; i < 3; ++i) cl.p[i].out = P.bsum + (int64_t)i * ks * P.slab;
    c.conv(cl, "enc_p.bert_proj", ks > 1 ? ks : 1, P.slab);
    if (cl.ksplit != ks) c.fail("enc_p.bert_proj (split mismatch)", -1);
    bert_slabs = 3 * ks;
  }
  {
    EmbedArgs e;
    e.x = x; e.tone = tone; e.lang = lang;
    e.emb = c.W(m.emb.off); e.tone_emb = c.W(m.tone_emb.off); e.lang_emb = c.W(m.lang_emb.off);
    e.n_vocab = cf.n_vocab; e.n_tones = cf.n_tones; e.n_langs = cf.n_languages;
    e.bsum = P.bsum; e.nslab = bert_slabs; e.slab_stride = P.slab;
    e.mask = mask; e.out = out_x; e.scale = (float)std::sqrt((double)H); e.B = B; e.C = H; e.T = T;
    for (int i = 0; i < 3; ++i) {
      e.idx[i] = bert_index ? bert_index[i] : nullptr;
      e.cols[i] = (bert_index && bert_index[i] && bert_cols) ? bert_cols[i] : 0;
    }
    c.chk(launch_embed(c.s, e), "embed");
  }
  c.tap("enc.x0", out_x, (int64_t)B * H * T);
  run_encoder(c, m.enc, P.enc, mask, spk, spk_stride, B, T, "enc", false, dp0, dp_c, spk_stride);
  {
    ConvLaunch cl;
    cl.nprob = 2; cl.B = B; cl.L = T;
    cl.p[0] = c.prob(m.proj_m, out_x, out_m, T);
    cl.p[1] = c.prob(m.proj_logs, out_x, out_logs, T);
    for (int i = 0; i < 2; ++i) { cl.p[i].out_mask = mask; cl.p[i].mask_post = 1; }
    c.conv(cl, "enc_p.proj");
  }
}

// DurationPredictor (reference models.py:285-299); dp0 = (x + cond(g)) [* mask is applied by conv_1's input mask]
static void dp_core(Ctx& c, PlanA& P, const float* dp0, const float* mask, int B, int T, float* logw_dp) {
  const Model& m = c.m;
  ConvProb p = c.prob(m.dp_c1, dp0, P.dp1, T);
  p.in_mask = mask; p.act = ACT_RELU;
  c.conv1(p, B, T, "dp.conv_1");
  LnArgs l;
  std::memset(&l, 0, sizeof(l));
  l.a = P.dp1; l.gamma = c.W(m.dp_g1.off); l.beta = c.W(m.dp_b1.off); l.eps = 1e-5f; l.out = P.dp1;
  l.B = B; l.C = kDpFilter; l.T = T;
  c.ln(l, "dp.norm_1");
  p = c.prob(m.dp_c2, P.dp1, P.dp2, T);
  p.in_mask = mask; p.act = ACT_RELU;
  c.conv1(p, B, T, "dp.conv_2");
  l.a = P.dp2; l.gamma = c.W(m.dp_g2.off); l.beta = c.W(m.dp_b2.off); l.out = P.dp2;
  c.ln(l, "dp.norm_2");
  p = c.prob(m.dp_proj, P.dp2, logw_dp, T);
  p.in_mask = mask; p.out_mask = mask; p.mask_post = 1;
  c.conv1(p, B, T, "dp.proj");
}

// StochasticDurationPredictor, reverse (reference models.py:197-204, 245-256) up to the last ConvFlow: on return P.z [B,2,T]
// holds the flow state whose channel 0 the ElementwiseAffine inverse (durations kernel) turns into logw.  P.z must already hold
// the SCALED noise (randn * noise_scale_w).
static void sdp_core(Ctx& c, PlanA& P, const float* x, const float* mask, const float* sdp_c, int sdp_c_stride, int B, int T) {
  const Model& m = c.m;
  const int H = m.cfg.hidden_channels;
  const float sqrt_fc = (float)std::sqrt((double)H);
  ConvProb p = c.prob(m.sdp_pre, x, P.sdp_h, T);
  p.bias2 = sdp_c; p.bias2_bstride = sdp_c_stride;              // x = pre(x) + cond(g)
  c.conv1(p, B, T, "sdp.pre");
  const bool fused = dds_fused_supported(H) && !c.h->no_fused_dds;
  if (fused) {
    DdsPost post;
    post.proj = &m.sdp_proj; post.post_out = P.sdp_x;             // x = proj(convs(x)) * mask: the ConvFlows' conditioning
    run_dds_fused(c, m.sdp_convs, P.sdp_h, DdsPre(), P.y1, P.y2, mask, post, B, H, T, sqrt_fc);
  } else {
    run_dds(c, m.sdp_convs, P.sdp_h, P.y1, P.y2, P.slab, mask, B, H, T);
    p = c.prob(m.sdp_proj, P.sdp_h, P.sdp_x, T);
    p.out_mask = mask; p.mask_post = 1;
    c.conv1(p, B, T, "sdp.proj");
  }
  c.tap("sdp.x", P.sdp_x, (int64_t)B * H * T);
  // Flip, CF, Flip, CF, Flip, CF, Flip, EA: the 2-channel flips are index swaps (src/dst), never data movement
  for (int i = 0; i < kSdpFlowsUsed; ++i) {
    const int src = (i % 2 == 0) ? 1 : 0, dst = 1 - src;
    const ConvFlowW& F = m.cf[i];
    if (fused) {
      DdsPre pre;
      pre.w = c.W(F.pre_w.off); pre.b = c.W(F.pre_b.off); pre.z = P.z; pre.z_src = src; pre.g = P.sdp_x;
      DdsPost post;
      post.proj = &F.proj; post.zio = P.z; post.z_src = src; post.z_dst = dst;
      run_dds_fused(c, F.convs, nullptr, pre, P.sdp_h, P.y1, mask, post, B, H, T, sqrt_fc);
    } else {
      c.chk(launch_convflow_pre(c.s, P.z, src, c.W(F.pre_w.off), c.W(F.pre_b.off), P.sdp_x, P.sdp_h, B, H, T), "cf.pre");
      run_dds(c, F.convs, P.sdp_h, P.y1, P.y2, P.slab, mask, B, H, T);
      p = c.prob(F.proj, P.sdp_h, P.params, T);
      p.out_bstride = (int64_t)32 * T;
      p.out_mask = mask; p.mask_post = 1;
      c.conv1(p, B, T, "cf.proj");
      c.chk(launch_spline(c.s, P.z, src, dst, P.params, 32, mask, sqrt_fc, 5.0f, B, T), "cf.spline");
    }
    const std::string tn = "sdp.z." + std::to_string(i);
    c.tap(tn.c_str(), P.z, (int64_t)B * 2 * T);
  }
}

static void run_durations(Ctx& c, const PlanA& P, const float* logw_dp, const float* mask, float sdp_ratio, float length_scale,
                          float* logw_sdp, float* logw, float* w_ceil, int64_t* y_lengths, int B, int T) {
  const Model& m = c.m;
  DurArgs d;
  d.z = P.z; d.ea_m = c.W(m.ea_m.off); d.ea_logs = c.W(m.ea_logs.off);
  d.logw_dp = logw_dp; d.mask = mask;
  d.sdp_ratio = sdp_ratio; d.one_minus_ratio = (float)(1.0 - (double)sdp_ratio); d.length_scale = length_scale;
  d.logw_sdp = logw_sdp; d.logw = logw; d.w_ceil = w_ceil; d.y_lengths = y_lengths;
  d.B = B; d.T = T;
  c.chk(launch_durations(c.s, d), "durations");
}

// ===============================================================================================================
// phase A: emb_g, enc_p, sdp, dp, durations
int run_encode(bv2_handle* h, hipStream_t s, const bv2_encode_in& in, const bv2_encode_out& out, void* ws, int64_t wsb) {
  const Model& m = h->model;
  const bv2_config& cf = m.cfg;
  const int B = in.B, T = in.T, H = cf.hidden_channels;
  Arena A(ws, wsb);
  PlanA P = plan_a(A, m, B, T);
  if (!A.ok()) { h->err = "workspace too small for bv2_encode_durations"; return -5; }
  Ctx c{h, s, m, h->blob};
  const float* mask = out.x_mask;

  // ONE front launch: g = emb_g(sid), x_mask, every g-conditioned vector of this phase, and the scaled SDP noise
  float *spk = P.gv, *sdp_c = P.gv + H, *dp_c = P.gv + 2 * H;
  {
    const GemvW* gw[3] = {&m.enc.spk, &m.sdp_cond, &m.dp_cond};
    float* go[3] = {spk, sdp_c, dp_c};
    run_front(c, in.sid, nullptr, out.g, gw, go, 3, 3 * H, in.x_lengths, out.x_mask, B, T, in.noise_w, P.z, in.noise_scale_w);
  }
  const float* berts[3] = {in.bert, in.ja_bert, in.en_bert};
  enc_p_core(c, P, in.x, in.tone, in.language, berts, mask, spk, 3 * H, B, T, out.x, out.m_p, out.logs_p, P.dp0, dp_c,
             in.bert_index, in.bert_cols);
  float* logw_dp = out.logw_dp ? out.logw_dp : P.logw_dp;
  // The two duration predictors only share their input (the encoder output).  Optional ("overlap_dp", default off — measured
  // slower, see bv2_internal.h): the deterministic one (5 short launches) on the handle's side stream beside the stochastic one
  // (17 launches), joined before the durations kernel.  Disjoint workspace.
  bool forked = false;
  if (!h->no_overlap_dp && !c.rc) {
    if (!h->side_stream) {
      if (hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        h->no_overlap_dp = true;                   // no side stream on this system: stay on the caller's stream
      }
    }
    if (!h->no_overlap_dp && hipEventRecord(h->ev_fork, s) == hipSuccess &&
        hipStreamWaitEvent(h->side_stream, h->ev_fork, 0) == hipSuccess) {
      Ctx c2{h, h->side_stream, m, h->blob};
      dp_core(c2, P, P.dp0, mask, B, T, logw_dp);
      if (c2.rc && !c.rc) c.rc = c2.rc;
      if (hipEventRecord(h->ev_join, h->side_stream) != hipSuccess) c.fail("dp join", -6);
      forked = true;
    }
  }
  if (!forked) dp_core(c, P, P.dp0, mask, B, T, logw_dp);
  sdp_core(c, P, out.x, mask, sdp_c, 3 * H, B, T);
  if (forked && hipStreamWaitEvent(s, h->ev_join, 0) != hipSuccess) c.fail("dp join", -6);
  run_durations(c, P, logw_dp, mask, in.sdp_ratio, in.length_scale, out.logw_sdp ? out.logw_sdp : P.logw_sdp, out.logw,
                out.w_ceil, out.y_lengths, B, T);
  return c.rc;
}

// ---- the reference's ONNX stages of phase A (include/bv2.h) --------------------------------------------------------
int run_stage_emb_g(bv2_handle* h, hipStream_t s, int B, const int64_t* sid, float* g) {
  const Model& m = h->model;
  Ctx c{h, s, m, h->blob};
  c.chk(launch_gather_rows(s, c.W(m.emb_g.off), sid, g, B, m.cfg.gin_channels, m.cfg.n_speakers), "emb_g");
  return c.rc;
}

int run_stage_enc_p(bv2_handle* h, hipStream_t s, int B, int T, const int64_t* x, const int64_t* tone, const int64_t* lang,
                    const float* b0, const float* b1, const float* b2, const float* g, const int64_t* x_lengths, float* xout,
                    float* m_p, float* logs_p, float* x_mask, void* ws, int64_t wsb) {
  const Model& m = h->model;
  const int H = m.cfg.hidden_channels;
  Arena A(ws, wsb);
  PlanA P = plan_a(A, m, B, T);
  if (!A.ok()) { h->err = "workspace too small for bv2_stage_enc_p"; return -5; }
  Ctx c{h, s, m, h->blob};
  const GemvW* gw[1] = {&m.enc.spk};
  float* go[1] = {P.gv};
  run_front(c, nullptr, g, nullptr, gw, go, 1, 3 * H, x_lengths, x_mask, B, T, nullptr, nullptr, 0.f);
  const float* berts[3] = {b0, b1, b2};
  enc_p_core(c, P, x, tone, lang, berts, x_mask, P.gv, 3 * H, B, T, xout, m_p, logs_p, nullptr, nullptr);
  return c.rc;
}

int run_stage_sdp(bv2_handle* h, hipStream_t s, int B, int T, const float* x, const float* x_mask, const float* zin,
                  const float* g, float* logw, void* ws, int64_t wsb) {
  const Model& m = h->model;
  const int H = m.cfg.hidden_channels;
  Arena A(ws, wsb);
  PlanA P = plan_a(A, m, B, T);
  if (!A.ok()) { h->err = "workspace too small for bv2_stage_sdp"; return -5; }
  Ctx c{h, s, m, h->blob};
  const GemvW* gw[1] = {&m.sdp_cond};
  float* go[1] = {P.gv + H};
  run_front(c, nullptr, g, nullptr, gw, go, 1, 3 * H, nullptr, nullptr, B, T, zin, P.z, 1.0f);   // zin is already scaled
  sdp_core(c, P, x, x_mask, P.gv + H, 3 * H, B, T);
  run_durations(c, P, nullptr, x_mask, 1.f, 1.f, logw, nullptr, nullptr, nullptr, B, T);
  return c.rc;
}

int run_stage_dp(bv2_handle* h, hipStream_t s, int B, int T, const float* x, const float* x_mask, const float* g, float* logw,
                 void* ws, int64_t wsb) {
  const Model& m = h->model;
  const int H = m.cfg.hidden_channels;
  Arena A(ws, wsb);
  PlanA P = plan_a(A, m, B, T);
  if (!A.ok()) { h->err = "workspace too small for bv2_stage_dp"; return -5; }
  Ctx c{h, s, m, h->blob};
  const GemvW* gw[1] = {&m.dp_cond};
  float* go[1] = {P.gv + 2 * H};
  run_front(c, nullptr, g, nullptr, gw, go, 1, 3 * H, nullptr, nullptr, B, T, nullptr, nullptr, 0.f);
  c.chk(launch_add_vec_mask(s, x, P.gv + 2 * H, 3 * H, nullptr, P.dp0, B, H, T), "dp.cond");
  dp_core(c, P, P.dp0, x_mask, B, T, logw);
  return c.rc;
}

// ===============================================================================================================
// flow reverse (reference models.py:138-145 / 438-445); z is updated in place
static void flow_core(Ctx& c, const PlanB& P, float* z, const float* ymask, const float* g, int B, int Ty) {
  const Model& m = c.m;
  const bv2_config& cf = m.cfg;
  const int H = cf.hidden_channels, C = cf.inter_channels, half = C / 2;
  float* gv_flow = P.gv + cf.upsample_initial_channel;
  bool pre_done = false;                            // the previous coupling's boundary launch already wrote this coupling's h
  if (m.flow_flip_first) c.chk(launch_flip_channels(c.s, z, B, C, Ty), "flow.flip");   // odd coupling count: see bv2_model.cpp
  for (int a = 0; a < m.n_coupling; ++a) {
    const CouplingW& K = m.coupling[a];
    float* x0 = z + (K.flipped ? (int64_t)half * Ty : 0);
    float* x1 = z + (K.flipped ? 0 : (int64_t)half * Ty);
    // small-N fp32 regime: LayerNorm-2 of the last Encoder layer + post + the NEXT coupling's pre run as one launch (flow_boundary.hip);
    // the next coupling's x0 is this coupling's x1 (the Flip is folded into the weights), so its `h` is ready when its turn comes
    bool fuse_b = cf.use_transformer_flow && c.h->flow_dtype != BV2_F16 && !c.h->no_fused_boundary && c.h->taps.empty() &&
                  n_slabs(B, Ty) > 1 && H == 192 && half * 2 == C && half * 2 == H && K.post.cin == H && K.post.cout == half && K.post.k == 1 &&
                  K.pre.cin == half && K.pre.cout == H && K.pre.k == 1 && K.post.cin_pad == H && K.pre.cin_pad == half;
    const CouplingW* Kn = a + 1 < m.n_coupling ? &m.coupling[a + 1] : nullptr;
    if (fuse_b && Kn) {
      const float* x0n = z + (Kn->flipped ? (int64_t)half * Ty : 0);
      fuse_b = x0n == x1 && Kn->pre.cin == half && Kn->pre.cout == H && Kn->pre.k == 1 && Kn->pre.cin_pad == half;
    }
    ConvProb p;
    if (!pre_done) {
      p = c.prob(K.pre, x0, P.h, Ty);
      p.x_bstride = (int64_t)C * Ty;
      p.out_mask = ymask; p.mask_post = 1;
      c.conv1(p, B, Ty, "flow.pre");
    }
    pre_done = false;
    const float* hres = P.h;
    if (cf.use_transformer_flow) {
      FbArgs F;
      std::memset(&F, 0, sizeof(F));
      if (fuse_b) {
        F.x1 = x1; F.x1_out = x1; F.z_bstride = (int64_t)C * Ty; F.C1 = half;
        F.post_w = c.W(K.post.w_off); F.post_b = c.W(K.post.b_off);
        if (Kn) { F.pre_w = c.W(Kn->pre.w_off); F.pre_b = c.W(Kn->pre.b_off); F.pre_out = P.h; pre_done = true; }
      }
      run_encoder(c, K.enc, P.enc, ymask, gv_flow + a * H, P.gv_stride, B, Ty, nullptr, c.h->flow_dtype == BV2_F16, nullptr, nullptr, 0,
                  fuse_b ? &F : nullptr);
      if (fuse_b && F.launched) continue;                        // post (and the next pre) are done
      pre_done = false;                                          // the boundary kernel declined these arguments: LayerNorm ran, post follows
    } else if (c.h->flow_dtype == BV2_F16) {
      // WN.forward on the fp16 matrix core (bv2_set_flow_dtype(BV2_F16)): in_layer reads the fp32 x (rounded while staged), adds
      // bias + g_l and gates in fp32, writes the gate output as fp16 channels-last — it is only ever res_skip's input; res_skip
      // runs as two problems of one launch on that tensor with the fp32 epilogues of the fp32 path (x update in place with the
      // mask, skip sum).  x, the skip sum, pre / post stay fp32.
      const int nl = K.wn_layers;
      const float* gl = gv_flow + (int64_t)a * 2 * H * nl;
      uint16_t* actsh = reinterpret_cast<uint16_t*>(P.acts);
      for (int i = 0; i < nl; ++i) {
        const bool last = i == nl - 1;
        HcProb q = c.hprob(K.wn_in[i], P.h, true, actsh, false, Ty);
        q.bias2 = gl + (int64_t)i * 2 * H; q.bias2_bstride = P.gv_stride;
        q.act = ACT_GATE; q.out_bstride = (int64_t)H * Ty;
        c.conv_h(q, B, Ty, "wn.in+gate");
        HcProb sk = c.hprob(K.wn_skip[i], actsh, false, P.outacc, true, Ty);
        if (i > 0) { sk.res = P.outacc; sk.res_mode = RES_ADD; }
        if (last) { sk.out_mask = ymask; sk.mask_post = 1; }
        if (!last) {
          HcProb r = c.hprob(K.wn_res[i], actsh, false, P.h, true, Ty);
          r.res = P.h; r.res_mode = RES_ADD; r.out_mask = ymask; r.mask_post = 1;
          c.conv_h(r, B, Ty, "wn.res_skip", &sk);
        } else {
          c.conv_h(sk, B, Ty, "wn.res_skip");
        }
      }
      hres = P.outacc;
    } else {
      const int nl = K.wn_layers;
      const float* gl = gv_flow + (int64_t)a * 2 * H * nl;
      // WN.forward (modules.py:185-210), 2 launches per layer: in_layer + g_l + gate (ACT_GATE epilogue), then res_skip as two
      // problems of one launch — x = (x + rs[:H]) * mask in place, output += rs[H:] (first layer: =; last layer: all H rows, * mask)
      for (int i = 0; i < nl; ++i) {
        const bool last = i == nl - 1;
        p = c.prob(K.wn_in[i], P.h, P.acts, Ty);
        p.bias2 = gl + (int64_t)i * 2 * H; p.bias2_bstride = P.gv_stride;
        p.act = ACT_GATE; p.out_bstride = (int64_t)H * Ty;
        c.conv1(p, B, Ty, "wn.in+gate");
        ConvLaunch cl;
        cl.B = B; cl.L = Ty; cl.nprob = 0;
        if (!last) {
          ConvProb r = c.prob(K.wn_res[i], P.acts, P.h, Ty);
          r.res = P.h; r.res_mode = RES_ADD; r.out_mask = ymask; r.mask_post = 1;
          cl.p[cl.nprob++] = r;
        }
        ConvProb sk = c.prob(K.wn_skip[i], P.acts, P.outacc, Ty);
        if (i > 0) { sk.res = P.outacc; sk.res_mode = RES_ADD; }
        if (last) { sk.out_mask = ymask; sk.mask_post = 1; }
        cl.p[cl.nprob++] = sk;
        c.conv(cl, "wn.res_skip");
      }
      hres = P.outacc;
    }
    {
      const std::string tn = "flow." + std::to_string(a) + ".enc";
      c.tap(tn.c_str(), hres, (int64_t)B * H * Ty);
    }
    p = c.prob(K.post, hres, x1, Ty);                          // x1 = (x1 - post(h)) * mask, written in place
    p.out_bstride = (int64_t)C * Ty; p.res = x1; p.res_bstride = (int64_t)C * Ty; p.res_mode = RES_RSUB;
    p.out_mask = ymask; p.mask_post = 1;
    c.conv1(p, B, Ty, "flow.post");
    {
      const std::string tn = "flow." + std::to_string(a) + ".z";
      c.tap(tn.c_str(), z, (int64_t)B * C * Ty);
    }
  }
}

// Generator.forward (reference models.py:538-557)
static void gen_core(Ctx& c, const PlanB& P, const float* z, int z_rstride, const float* ymask, int B, int L, float* o,
                     const int64_t* lens) {
  const Model& m = c.m;
  const bv2_config& cf = m.cfg;
  const int C = cf.inter_channels, c0 = cf.upsample_initial_channel;
  {
    // conv_pre((z*y_mask)[:, :, :L]) + cond(g)
    ConvProb p = c.prob(m.conv_pre, z, P.pre, L);
    p.x_bstride = (int64_t)C * z_rstride; p.x_rstride = z_rstride; p.Lin = L;
    p.in_mask = ymask; p.in_mask_bstride = z_rstride;
    p.bias2 = P.gv; p.bias2_bstride = P.gv_stride;
    c.conv1(p, B, L, "dec.conv_pre", 1, 0, lens, 1);
  }
  c.tap("dec.pre", P.pre, (int64_t)B * c0 * L);
  const float* src[3] = {P.pre, nullptr, nullptr};
  int nsrc = 1;
  int Lc = L;
  int up = 1;                                     // samples per latent frame at the current resolution
  // x3 form of the wide-stage convs (conv_x6.hip, two scaled fp16 planes): every conv input needs the slot its producer's epilogue
  // filled with max |x|; the slots of one decode are distinct and zeroed here
  int next_slot = 0;
  int n_slots = 0;                                // slots this decode can use: stages with 128-row tiles
  for (int i = 0; i < m.n_ups && !c.h->no_conv_x3 && !c.h->no_conv_x6; ++i)
    if (m.ups[i].cout % 128 == 0 && m.rb[i][0][0][0].wy_off >= 0) n_slots += 1 + 2 * m.n_rbk * m.n_rbd;
  const bool any_x3 = n_slots > 0;
  if (any_x3 && !c.rc) c.chk(launch_x3_zero_slots(c.s, P.xslots, n_slots), "dec.x3_slots");
  for (int i = 0; i < m.n_ups; ++i) {
    const UpW& U = m.ups[i];
    float* const* S = P.set[i & 1];
    float* x = S[0];
    const int Lo = Lc * U.u;
    const int nb = m.n_rbk;
    // the n_rbk ResBlock1 branches run side by side
    const bool rb2 = m.rb_type == 2;
    bool fused = U.cout <= 32 && nb <= 3 && !c.h->no_fused_resblock && !rb2;
    // C = 32: two split-bf16 launches per pair (HBM-bound, 5 tensor passes) against one fused fp32-MFMA launch (MFMA-bound, 3 passes)
    // C = 32 with the planes packed: the pair in ONE launch on the bf16 matrix core (respair_x6.hip: two passes AND the fast pipe)
    bool x6pair = !rb2 && nb <= 3 && !c.h->no_fused_resblock && !c.h->no_conv_x6 && !c.h->no_x6_pair &&
                  (U.cout == 32 || (U.cout == 16 && !c.h->no_x6_pair_c16) || (U.cout == 64 && !c.h->no_x6_pair_c64) ||
                   (U.cout == 128 && c.h->x6_pair_c128));
    for (int j = 0; j < nb && x6pair; ++j)
      for (int d = 0; d < m.n_rbd && x6pair; ++d)
        x6pair = m.rb[i][j][d][0].wx_off >= 0 && m.rb[i][j][d][1].wx_off >= 0 && m.rb[i][j][d][0].k == m.rb[i][j][d][1].k &&
                 respair_x6_supported(U.cout, cf.resblock_kernel_sizes[j], cf.resblock_dilation_sizes[j][d]);
    if (x6pair) fused = true;
    if (fused && !x6pair && c.h->x6_narrow && !c.h->no_conv_x6 && U.cout == 32 && m.rb[i][0][0][0].wx_off >= 0) fused = false;
    for (int j = 0; j < nb && fused && !x6pair; ++j)
      for (int d = 0; d < m.n_rbd; ++d)
        fused = fused && resblock_fused_supported(U.cout, cf.resblock_kernel_sizes[j], cf.resblock_dilation_sizes[j][d]);
    // x3 slots only for a stage that runs layer-wise on conv_x6.hip's 128-row tiles (the pair kernel scales per workgroup tile)
    bool x3 = any_x3 && !rb2 && !fused && U.cout % 128 == 0;
    for (int j = 0; j < nb && x3; ++j)
      for (int d = 0; d < m.n_rbd && x3; ++d) x3 = m.rb[i][j][d][0].wy_off >= 0 && m.rb[i][j][d][1].wy_off >= 0;
    unsigned* const slot_x = x3 ? P.xslots + X3_SLOT_WORDS * next_slot++ : nullptr;
    {
      // x = ConvTranspose1d(leaky_relu(mean of the previous stage's branches)) as U.u polyphase stride-1 convs
      ConvLaunch cl;
      cl.nprob = U.u; cl.B = B; cl.L = Lc; cl.lens = lens; cl.len_mul = up;
      for (int ph = 0; ph < U.u; ++ph) {
        ConvProb p = c.prob(U.phase[ph], src[0], x, Lc);
        p.x[1] = src[1]; p.x[2] = src[2]; p.nsrc = nsrc; p.in_scale = 1.f / (float)nsrc;
        p.pre_act = PRE_LRELU; p.slope = 0.1f;
        p.pad_left = U.pad_left[ph];
        p.out_bstride = (int64_t)U.cout * Lo; p.out_rstride = Lo; p.out_tstride = U.u; p.out_toff = ph;
        p.omax = slot_x;
        cl.p[ph] = p;
      }
      c.conv(cl, "dec.ups");
    }
    {
      const std::string tn = "dec.ups." + std::to_string(i);
      c.tap(tn.c_str(), x, (int64_t)B * U.cout * Lo);
    }
    float* branch_out[BV2_MAX_RESBLOCK_KERNELS];
    if (rb2) {
      // modules.ResBlock2 (reference modules.py:348-357): for each of the two dilations x = x + conv_d(lrelu(x)) — one conv per launch
      // with the residual in its epilogue, the branches side by side; branch j ping-pongs between S[1 + nb + j] and S[1 + j] and ends in S[1 + j]
      for (int d = 0; d < m.n_rbd; ++d) {
        ConvLaunch c1;
        c1.nprob = nb; c1.B = B; c1.L = Lo; c1.lens = lens; c1.len_mul = up * U.u;
        const bool to_cur = ((m.n_rbd - 1 - d) & 1) == 0;
        for (int jj = 0; jj < nb; ++jj) {
          const int j = nb - 1 - jj;                            // widest kernel first
          float* cur = S[1 + j];
          float* tmp = S[1 + nb + j];
          const float* xin = d == 0 ? x : (to_cur ? tmp : cur);
          ConvProb p = c.prob(m.rb[i][j][d][0], xin, to_cur ? cur : tmp, Lo, cf.resblock_dilation_sizes[j][d]);
          p.pre_act = PRE_LRELU; p.slope = 0.1f;
          p.res = xin; p.res_mode = RES_ADD;
          p.w6 = (!c.h->no_conv_x6 && m.rb[i][j][d][0].wx_off >= 0) ? reinterpret_cast<const uint16_t*>(c.W(m.rb[i][j][d][0].wx_off)) : nullptr;
          c1.p[jj] = p;
        }
        c.conv(c1, "dec.resblock2.conv");
      }
      for (int j = 0; j < nb; ++j) branch_out[j] = S[1 + j];
    } else if (fused) {
      // narrow stages: ONE launch per dilation step = the whole (conv, conv) pair of every branch, intermediate in LDS;
      // the pair ping-pongs between the branch's two buffers (a tile reads its neighbours' halo: no in-place update)
      for (int j = 0; j < nb; ++j) branch_out[j] = nullptr;
      for (int d = 0; d < m.n_rbd; ++d) {
        FusedLaunch F;
        std::memset(&F, 0, sizeof(F));
        F.nprob = nb; F.B = B; F.C = U.cout; F.L = Lo; F.slope = 0.1f; F.lens = lens; F.len_mul = up * U.u;
        for (int jj = 0; jj < nb; ++jj) {
          const int j = nb - 1 - jj;                            // widest kernel first
          float* a = S[1 + j];
          float* b2 = S[1 + nb + j];
          const float* xin = d == 0 ? x : branch_out[j];
          float* xout = (xin == a) ? b2 : a;
          FusedProb& p = F.p[jj];
          p.x = xin; p.out = xout;
          p.w1 = c.W(m.rb[i][j][d][0].w_off); p.b1 = c.W(m.rb[i][j][d][0].b_off);
          p.w2 = c.W(m.rb[i][j][d][1].w_off); p.b2 = c.W(m.rb[i][j][d][1].b_off);
          p.k = m.rb[i][j][d][0].k; p.dil = cf.resblock_dilation_sizes[j][d];
          if (x6pair) {
            p.w61 = reinterpret_cast<const uint16_t*>(c.W(m.rb[i][j][d][0].wx_off));
            p.w62 = reinterpret_cast<const uint16_t*>(c.W(m.rb[i][j][d][1].wx_off));
            if (!c.h->no_conv_x3 && m.rb[i][j][d][0].wy_off >= 0 && m.rb[i][j][d][1].wy_off >= 0) {   // the x3 form: scaled fp16 planes
              p.w3inv1 = c.W(m.rb[i][j][d][0].wy_off); p.w3inv2 = c.W(m.rb[i][j][d][1].wy_off);
              p.w31 = reinterpret_cast<const uint16_t*>(p.w3inv1 + X3_HDR_FLOATS);
              p.w32 = reinterpret_cast<const uint16_t*>(p.w3inv2 + X3_HDR_FLOATS);
            }
          }
          branch_out[j] = xout;
        }
        if (!c.rc) {
          const int pi = c.prof_begin("dec.resblock.fused");
          if (pi >= 0 && c.h->prof_mode >= 3)
            c.cur_shape = " n" + std::to_string(nb) + " C" + std::to_string(U.cout) + " k" + std::to_string(F.p[0].k) + " L" +
                          std::to_string(Lo) + " B" + std::to_string(B);
          const int r = x6pair ? launch_respair_x6(c.s, F) : launch_resblock_fused(c.s, F);
          c.prof_end(pi, x6pair ? (F.p[0].w31 ? (U.cout == 16 ? "respair_x3<16>" : U.cout == 32 ? "respair_x3<32>" : (U.cout == 64 ? "respair_x3<64>" : "respair_x3<128>"))
                                              : (U.cout == 16 ? "respair_x6<16>" : U.cout == 32 ? "respair_x6<32>" : (U.cout == 64 ? "respair_x6<64>" : "respair_x6<128>"))) : "resblock_fused", resblock_fused_flops(F), resblock_fused_bytes(F));
          if (r) c.fail("dec.resblock.fused", r);
        }
      }
    } else {
      // wide stages: 2 launches per dilation step, each carrying all branches
      unsigned* slot_cur[BV2_MAX_RESBLOCK_KERNELS] = {nullptr, nullptr, nullptr, nullptr};   // max |cur_j| after the previous dilation step
      for (int d = 0; d < m.n_rbd; ++d) {
        ConvLaunch c1, c2;
        c1.nprob = c2.nprob = nb; c1.B = c2.B = B; c1.L = c2.L = Lo;
        c1.lens = c2.lens = lens; c1.len_mul = c2.len_mul = up * U.u;
        for (int jj = 0; jj < nb; ++jj) {
          const int j = nb - 1 - jj;                            // widest kernel first: longest workgroups start first
          float* cur = S[1 + j];
          float* tmp = S[1 + nb + j];
          const float* xin = d == 0 ? x : cur;
          const bool x6 = !c.h->no_conv_x6;                       // the split-bf16 planes ride along: launch_conv1d picks conv_x6.hip
          auto planes3 = [&](ConvProb& q, const ConvW& w, const unsigned* in_slot, unsigned* out_slot) {
            if (!x3) return;
            q.w3inv = c.W(w.wy_off);
            q.w3 = reinterpret_cast<const uint16_t*>(c.W(w.wy_off) + X3_HDR_FLOATS);
            q.xmax = in_slot; q.omax = out_slot;
          };
          unsigned* const slot_tmp = x3 ? P.xslots + X3_SLOT_WORDS * next_slot++ : nullptr;
          unsigned* const slot_out = (x3 && d + 1 < m.n_rbd) ? P.xslots + X3_SLOT_WORDS * next_slot++ : nullptr;   // the last step's output feeds no x3 conv
          ConvProb p = c.prob(m.rb[i][j][d][0], xin, tmp, Lo, cf.resblock_dilation_sizes[j][d]);
          p.pre_act = PRE_LRELU; p.slope = 0.1f;
          p.w6 = (x6 && m.rb[i][j][d][0].wx_off >= 0) ? reinterpret_cast<const uint16_t*>(c.W(m.rb[i][j][d][0].wx_off)) : nullptr;
          planes3(p, m.rb[i][j][d][0], d == 0 ? slot_x : slot_cur[j], slot_tmp);
          c1.p[jj] = p;
          p = c.prob(m.rb[i][j][d][1], tmp, cur, Lo, 1);
          p.pre_act = PRE_LRELU; p.slope = 0.1f;
          p.res = xin; p.res_mode = RES_ADD;
          p.w6 = (x6 && m.rb[i][j][d][1].wx_off >= 0) ? reinterpret_cast<const uint16_t*>(c.W(m.rb[i][j][d][1].wx_off)) : nullptr;
          planes3(p, m.rb[i][j][d][1], slot_tmp, slot_out);
          c2.p[jj] = p;
          slot_cur[j] = slot_out;
        }
        c.conv(c1, "dec.resblock.convs1");
        c.conv(c2, "dec.resblock.convs2");
      }
      for (int j = 0; j < nb; ++j) branch_out[j] = S[1 + j];
    }
    for (int j = 0; j < nb; ++j) {
      const std::string tn = "dec.rb." + std::to_string(i) + "." + std::to_string(j);
      c.tap(tn.c_str(), branch_out[j], (int64_t)B * U.cout * Lo);
    }
    for (int j = 0; j < 3; ++j) src[j] = j < nb ? branch_out[j] : nullptr;
    nsrc = nb;
    Lc = Lo;
    up *= U.u;
  }
  ConvPostArgs a;
  std::memset(&a, 0, sizeof(a));
  a.lens = lens; a.len_mul = up;
  for (int j = 0; j < 3; ++j) a.x[j] = src[j];
  a.nsrc = nsrc; a.in_scale = 1.f / (float)nsrc; a.x_bstride = (int64_t)m.post_c * Lc; a.x_rstride = Lc;
  a.w = c.W(m.conv_post.off); a.out = o; a.out_bstride = Lc; a.C = m.post_c; a.k = m.post_k; a.L = Lc; a.B = B;
  a.slope = 0.01f;                                            // F.leaky_relu default (models.py:553)
  c.chk(launch_conv_post(c.s, a), "dec.conv_post");
}

// Generator.forward in bf16, channels-last (kernels/gen_bf16.hip).  Same launch structure as the wide-stage fp32 path: per
// stage one ConvTranspose launch (a single conv with C_out' = u*C_out) and per dilation step two launches carrying the
// n_rbk branches; every tensor between launches is bf16 [B][L][C].  Rounding points (mirrored by oracle/bv2_oracle.py
// generator_bf16): weights, (z*mask), every stored activation, and the pre-activated conv inputs are rounded to bf16
// (RNE); accumulation, bias, residual and the branch mean are fp32; conv_post + tanh are fp32 on bf16 inputs.
static void gen_core_bf16(Ctx& c, const PlanB& P, const float* z, int z_rstride, const float* ymask, int B, int L, float* o,
                          const int64_t* lens) {
  const Model& m = c.m;
  const bv2_config& cf = m.cfg;
  const int C = cf.inter_channels, c0 = cf.upsample_initial_channel;
  auto U16 = [](float* p) { return reinterpret_cast<uint16_t*>(p); };
  auto Wb = [&](const ConvW& w) { return reinterpret_cast<const uint16_t*>(c.W(w.wb_off)); };
  auto prob = [&](const ConvW& w, const uint16_t* x, uint16_t* out, int Lc, int dil) {
    ClProb p;
    std::memset(&p, 0, sizeof(p));
    p.x[0] = x; p.nsrc = 1; p.in_scale = 1.f; p.x_bstride = (int64_t)w.cin * Lc; p.Lin = Lc;
    p.w = Wb(w); p.bias = c.W(w.b_off);
    p.out = out; p.out_bstride = (int64_t)w.cout * Lc; p.res_bstride = p.out_bstride;
    p.cin = w.cin; p.cout = w.cout; p.cout_pad = w.cout_pad; p.k = w.k; p.dil = dil; p.pad_left = ((w.k - 1) / 2) * dil;
    p.slope = 0.1f;
    return p;
  };
  auto launch = [&](ClLaunch& cl, const char* tag, double flops) {
    if (c.rc) return;
    const char* vn = "conv_cl_bf16";
    const int pi = c.prof_begin(tag);
    if (pi >= 0 && c.h->prof_mode >= 3) {
      const ClProb& q = cl.p[0];
      c.cur_shape = " n" + std::to_string(cl.nprob) + " " + std::to_string(q.cin) + ">" + std::to_string(q.cout) + " k" +
                    std::to_string(q.k) + " L" + std::to_string(cl.L) + " B" + std::to_string(cl.B);
    }
    const int r = launch_conv_cl_bf16(c.s, cl, &vn);
    c.prof_end(pi, vn, flops, conv_cl_bytes(cl));
    if (r) c.fail(tag, r);
  };
  auto flops_of = [&](const ClLaunch& cl) {
    double f = 0;
    for (int i = 0; i < cl.nprob; ++i) f += 2.0 * cl.p[i].cout * cl.p[i].cin * cl.p[i].k * (double)cl.L * cl.B;
    return f;
  };

  // debug taps (tests): widen a bf16 channels-last tensor into the caller's fp32 [B][C][L] tap buffer
  auto tap_cl = [&](const std::string& name, const uint16_t* x, int Cc, int Lc) {
    if (c.h->taps.empty() || c.rc) return;
    auto it = c.h->taps.find(name);
    if (it == c.h->taps.end() || it->second.cap < (int64_t)B * Cc * Lc) return;
    c.chk(launch_uncast_cl(c.s, x, it->second.dst, B, Cc, Lc), "tap");
  };

  uint16_t* zc = U16(P.set[1][0]);                // dead before stage 1 writes set[1][0]
  c.chk(launch_cast_cl(c.s, z, z_rstride, (int64_t)C * z_rstride, ymask, z_rstride, zc, B, C, L), "dec.cast");
  uint16_t* pre = U16(P.pre);
  {
    ClLaunch cl;
    cl.nprob = 1; cl.B = B; cl.L = L; cl.lens = lens; cl.len_mul = 1;
    cl.p[0] = prob(m.conv_pre, zc, pre, L, 1);
    cl.p[0].bias2 = P.gv; cl.p[0].bias2_bstride = P.gv_stride;
    launch(cl, "dec.conv_pre", flops_of(cl));
  }
  tap_cl("dec.pre", pre, c0, L);
  const uint16_t* src[3] = {pre, nullptr, nullptr};
  int nsrc = 1, Lc = L, up = 1;
  for (int i = 0; i < m.n_ups; ++i) {
    const UpW& U = m.ups[i];
    float* const* S = P.set[i & 1];
    uint16_t* x = U16(S[0]);
    const int Lo = Lc * U.u;
    {
      ClLaunch cl;
      cl.nprob = 1; cl.B = B; cl.L = Lc; cl.lens = lens; cl.len_mul = up; cl.ups = 1;
      ClProb p = prob(U.cl, src[0], x, Lc, 1);
      p.x[1] = src[1]; p.x[2] = src[2]; p.nsrc = nsrc; p.in_scale = 1.f / (float)nsrc;
      p.pre_lrelu = 1; p.pad_left = U.cl_pad_left;
      if (!c.h->no_ups_phase_taps && U.u <= 16) {                  // phase ph's taps inside the union window: [off, off + ntaps)
        unsigned long long offs = 0;
        bool fits = true;
        for (int ph = 0; ph < U.u; ++ph) {
          const int off = U.cl_pad_left - U.pad_left[ph];
          fits = fits && off >= 0 && off <= 15;
          offs |= (unsigned long long)(off & 15) << (4 * ph);
        }
        if (fits) { p.ph_cout = U.cout; p.ph_ntaps = U.ntaps; p.ph_offs = offs; }
      }
      cl.p[0] = p;
      // algorithmic FLOPs: the true taps of the transposed conv (the zero-padded window taps are not counted)
      launch(cl, "dec.ups", 2.0 * U.cin * U.cout * U.k * (double)Lc * B);
    }
    tap_cl("dec.ups." + std::to_string(i), x, U.cout, Lo);
    const int nb = m.n_rbk;
    const bool rb2 = m.rb_type == 2;
    bool whole = nb <= 3 && !c.h->no_fused_resblock && !rb2;
    for (int j = 0; j < nb && whole; ++j) whole = m.rbcl_w_off[i][j] >= 0;
    // C = 32: pair by pair (respair_cl_bf16.hip, one wave owns all 32 channels) — six tensor passes per ResBlock instead of two, but
    // every launch streams near the HBM rate with three workgroups per CU, where the whole-ResBlock kernel's one resident workgroup
    // runs its six GEMM / epilogue / barrier phases in lock-step: 1.58 -> 1.36 ms per step at B = 32.  (C = 16 through the same kernel
    // on a half-empty MFMA block: 1.49 -> 1.72 ms, not kept.)
    if (whole && U.cout == 32 && !c.h->no_respair_c32 && !c.h->no_fused_respair) whole = false;
    const bool narrow_layerwise = c.h->no_fused_resblock && U.cout <= 32;    // "fused_resblock" = 0: one conv per launch on the narrow stages
    // Stage hand-over (round 6): the kernel that finishes a stage's branches runs them tile by tile in one workgroup and writes ONE tensor, the
    // branch mean (cl_bf16.h stage_mean) — the next ConvTranspose1d / conv_post reads 1 tensor instead of n.  Off ("stage_sum" = 0, a branch tap
    // set, a kernel without the form): n tensors and the consumer forms the same mean with the same rounding points — bit-identical results.
    bool want_sum = !c.h->no_stage_sum && nb > 1 && nb <= 3;
    if (want_sum)
      for (const auto& kv : c.h->taps)
        if (kv.first.compare(0, 7, "dec.rb.") == 0) want_sum = false;
    bool summed = false;
    uint16_t* const sum_t = U16(S[1]);              // branch 0's final buffer: no launch of the stage's last step reads it
    if (whole) {
      // narrow stages: every branch's whole ResBlock (all dilation pairs) in ONE launch, intermediates in LDS
      RbClLaunch F;
      std::memset(&F, 0, sizeof(F));
      F.nprob = nb; F.B = B; F.C = U.cout; F.L = Lo; F.nd = m.n_rbd; F.slope = 0.1f; F.lens = lens; F.len_mul = up * U.u;
      for (int jj = 0; jj < nb; ++jj) {
        const int j = nb - 1 - jj;                                // widest kernel first
        RbClProb& p = F.p[jj];
        p.x = x; p.out = U16(S[1 + j]);
        p.w = reinterpret_cast<const uint16_t*>(c.W(m.rbcl_w_off[i][j])); p.bias = c.W(m.rbcl_b_off[i][j]);
        p.k = cf.resblock_kernel_sizes[j];
        for (int d = 0; d < m.n_rbd; ++d) p.dil[d] = cf.resblock_dilation_sizes[j][d];
      }
      // C = 16: the tap-pair form on v_mfma_f32_16x16x32_bf16 (resblock_c16_bf16.hip) when every branch has its stream
      bool c16 = U.cout == 16 && !c.h->no_resblock_c16;
      for (int j = 0; j < nb && c16; ++j) c16 = m.rb16_w_off[i][j] >= 0;
      if (c16)
        for (int jj = 0; jj < nb; ++jj) {
          const int j = nb - 1 - jj;
          F.p[jj].w = reinterpret_cast<const uint16_t*>(c.W(m.rb16_w_off[i][j])); F.p[jj].bias = c.W(m.rb16_b_off[i][j]);
        }
      if (c16 && want_sum) { F.sum_out = sum_t; summed = true; }
      if (!c.rc) {
        const int pi = c.prof_begin("dec.resblock.whole");
        if (pi >= 0 && c.h->prof_mode >= 3)
          c.cur_shape = " n" + std::to_string(nb) + " C" + std::to_string(U.cout) + " k" + std::to_string(F.p[0].k) + " L" +
                        std::to_string(Lo) + " B" + std::to_string(B);
        const int r = c16 ? launch_resblock_c16_bf16(c.s, F) : launch_resblock_cl_bf16(c.s, F);
        c.prof_end(pi, c16 ? "resblock_c16_bf16" : "resblock_cl_bf16", resblock_cl_bf16_flops(F), resblock_cl_bf16_bytes(F));
        if (r) c.fail("dec.resblock.whole", r);
      }
    }
    // wide stages: one (dilated conv, conv) pair per launch, the intermediate in LDS (respair_cl_bf16.hip).  A tile's halo rows are
    // another tile's outputs, so a pair never runs in place: branch j ping-pongs between S[1 + j] and S[1 + nb + j] and ends in S[1 + j]
    bool pairs = !rb2 && !whole && nb <= 3 && !c.h->no_fused_respair && !narrow_layerwise;
    for (int j = 0; j < nb && pairs; ++j)
      for (int d = 0; d < m.n_rbd && pairs; ++d)
        pairs = respair_cl_bf16_supported(U.cout, cf.resblock_kernel_sizes[j], cf.resblock_dilation_sizes[j][d]) &&
                m.rb[i][j][d][0].k == m.rb[i][j][d][1].k;
    for (int d = 0; d < m.n_rbd && pairs; ++d) {
      RpClLaunch F;
      std::memset(&F, 0, sizeof(F));
      F.nprob = nb; F.B = B; F.C = U.cout; F.L = Lo; F.slope = 0.1f; F.lens = lens; F.len_mul = up * U.u;
      F.mix = c.h->respair_problem_major ? 0 : 1;
      F.form = c.h->respair_form;
      const bool to_cur = ((m.n_rbd - 1 - d) & 1) == 0;
      if (want_sum && d + 1 == m.n_rbd) { F.sum_out = sum_t; summed = true; }     // the stage's last pair launch leaves the branch mean
      for (int jj = 0; jj < nb; ++jj) {
        const int j = nb - 1 - jj;                                // widest kernel first
        uint16_t* cur = U16(S[1 + j]);
        uint16_t* tmp = U16(S[1 + nb + j]);
        RpClProb& p = F.p[jj];
        p.x = d == 0 ? x : (to_cur ? tmp : cur);
        p.out = to_cur ? cur : tmp;
        p.w1 = Wb(m.rb[i][j][d][0]); p.w2 = Wb(m.rb[i][j][d][1]);
        p.b1 = c.W(m.rb[i][j][d][0].b_off); p.b2 = c.W(m.rb[i][j][d][1].b_off);
        p.k = m.rb[i][j][d][0].k; p.dil = cf.resblock_dilation_sizes[j][d];
      }
      if (!c.rc) {
        const char* vn = "respair_cl_bf16";
        const int pi = c.prof_begin("dec.resblock.pair");
        if (pi >= 0 && c.h->prof_mode >= 3)
          c.cur_shape = " n" + std::to_string(nb) + " C" + std::to_string(U.cout) + " k" + std::to_string(F.p[0].k) + " L" +
                        std::to_string(Lo) + " B" + std::to_string(B);
        const int r = launch_respair_cl_bf16(c.s, F, &vn);
        c.prof_end(pi, vn, respair_cl_bf16_flops(F), respair_cl_bf16_bytes(F));
        if (r) c.fail("dec.resblock.pair", r);
      }
    }
    for (int d = 0; d < m.n_rbd && rb2; ++d) {
      // modules.ResBlock2 in bf16: x = bf16(conv_d(bf16(lrelu(x))) + x), one conv per launch, ending in S[1 + j]
      ClLaunch c1;
      c1.nprob = nb; c1.B = B; c1.L = Lo; c1.lens = lens; c1.len_mul = up * U.u;
      const bool to_cur = ((m.n_rbd - 1 - d) & 1) == 0;
      for (int jj = 0; jj < nb; ++jj) {
        const int j = nb - 1 - jj;
        uint16_t* cur = U16(S[1 + j]);
        uint16_t* tmp = U16(S[1 + nb + j]);
        const uint16_t* xin = d == 0 ? x : (to_cur ? tmp : cur);
        ClProb p = prob(m.rb[i][j][d][0], xin, to_cur ? cur : tmp, Lo, cf.resblock_dilation_sizes[j][d]);
        p.pre_lrelu = 1; p.res = xin;
        c1.p[jj] = p;
      }
      launch(c1, "dec.resblock2.conv", flops_of(c1));
    }
    for (int d = 0; d < m.n_rbd && !rb2 && !whole && !pairs; ++d) {
      ClLaunch c1, c2;
      c1.nprob = c2.nprob = nb; c1.B = c2.B = B; c1.L = c2.L = Lo;
      c1.lens = c2.lens = lens; c1.len_mul = c2.len_mul = up * U.u;
      for (int jj = 0; jj < nb; ++jj) {
        const int j = nb - 1 - jj;                              // widest kernel first
        uint16_t* cur = U16(S[1 + j]);
        uint16_t* tmp = U16(S[1 + nb + j]);
        const uint16_t* xin = d == 0 ? x : cur;
        ClProb p = prob(m.rb[i][j][d][0], xin, tmp, Lo, cf.resblock_dilation_sizes[j][d]);
        p.pre_lrelu = 1;
        c1.p[jj] = p;
        p = prob(m.rb[i][j][d][1], tmp, cur, Lo, 1);
        p.pre_lrelu = 1; p.res = xin;
        c2.p[jj] = p;
      }
      launch(c1, "dec.resblock.convs1", flops_of(c1));
      launch(c2, "dec.resblock.convs2", flops_of(c2));
    }
    for (int j = 0; j < 3; ++j) src[j] = j < nb && (j == 0 || !summed) ? U16(S[1 + j]) : nullptr;
    if (summed) tap_cl("dec.stage." + std::to_string(i), src[0], U.cout, Lo);
    else for (int j = 0; j < nb; ++j) tap_cl("dec.rb." + std::to_string(i) + "." + std::to_string(j), src[j], U.cout, Lo);
    nsrc = summed ? 1 : nb;
    Lc = Lo;
    up *= U.u;
  }
  ConvPostClArgs a;
  std::memset(&a, 0, sizeof(a));
  a.lens = lens; a.len_mul = up;
  for (int j = 0; j < 3; ++j) a.x[j] = src[j];
  a.nsrc = nsrc; a.in_scale = 1.f / (float)nsrc;
  a.w = c.W(m.conv_post.off); a.out = o; a.C = m.post_c; a.k = m.post_k; a.L = Lc; a.B = B;
  a.slope = 0.01f;                                            // F.leaky_relu default (models.py:553)
  a.generic = c.h->no_conv_post_rows ? 1 : 0;
  c.chk(launch_conv_post_cl(c.s, a), "dec.conv_post");
}

static void phase_b_gemv(Ctx& c, const PlanB& P, const float* g, int B) {
  const Model& m = c.m;
  const bv2_config& cf = m.cfg;
  GemvLaunch G;
  std::memset(&G, 0, sizeof(G));
  int n = 0;
  auto add = [&](const GemvW& w, float* out) {
    G.p[n].w = c.W(w.w_off); G.p[n].bias = c.W(w.b_off); G.p[n].out = out; G.p[n].cout = w.cout; G.p[n].cin = w.cin;
    G.p[n].out_bstride = P.gv_stride;
    ++n;
  };
  add(m.dec_cond, P.gv);
  float* gf = P.gv + cf.upsample_initial_channel;
  for (int a = 0; a < m.n_coupling; ++a) {
    if (cf.use_transformer_flow) add(m.coupling[a].enc.spk, gf + a * cf.hidden_channels);
    else add(m.coupling[a].wn_cond, gf + (int64_t)a * 2 * cf.hidden_channels * cf.n_flow_layer);
  }
  G.nprob = n; G.B = B; G.g = g; G.g_bstride = cf.gin_channels;
  c.chk(launch_gemv(c.s, G), "gemv.B");
}

int run_decode(bv2_handle* h, hipStream_t s, const bv2_decode_in& in, const bv2_decode_out& out, void* ws, int64_t wsb) {
  const Model& m = h->model;
  const bv2_config& cf = m.cfg;
  const int B = in.B, T = in.T, Ty = in.Ty, C = cf.inter_channels;
  Arena A(ws, wsb);
  PlanB P = plan_b(A, m, B, Ty);
  if (!A.ok()) { h->err = "workspace too small for bv2_decode"; return -5; }
  Ctx c{h, s, m, h->blob};
  float* z = out.z ? out.z : P.z;
  float* ymask = out.y_mask ? out.y_mask : P.ymask;

  ExpandArgs e;
  std::memset(&e, 0, sizeof(e));
  e.w_ceil = in.w_ceil; e.x_mask = in.x_mask; e.y_lengths = in.y_lengths; e.m_p = in.m_p; e.logs_p = in.logs_p;
  e.noise = in.noise_z; e.nz_bstride = in.nz_bstride; e.nz_cstride = in.nz_cstride; e.nz_tstride = in.nz_tstride;
  e.noise_scale = in.noise_scale;
  e.frame_idx = P.fidx; e.attn = out.attn; e.y_mask = ymask; e.z_p = z; e.m_e = out.m_p; e.logs_e = out.logs_p;
  e.z_p2 = out.z_p;                                  // the flow updates z in place: z_p is kept as a second store
  e.B = B; e.C = C; e.T = T; e.Ty = Ty;
  c.chk(launch_expand(s, e), "expand");
  phase_b_gemv(c, P, in.g, B);
  flow_core(c, P, z, ymask, in.g, B, Ty);
  const int L = (in.max_len > 0 && in.max_len < Ty) ? in.max_len : Ty;
  const int64_t* lens = in.exact_lengths == 1 ? in.y_lengths : nullptr;
  if (in.exact_lengths == 2) {                       // Ty is a bucket >= max(y_lengths): cap the Generator at the longest utterance (bv2.h)
    lens = in.y_lengths;                             // B == 1: the cap is the utterance's own length
    if (B > 1) { c.chk(launch_len_cap(s, in.y_lengths, P.len_cap, B), "len_cap"); lens = P.len_cap; }
  }
  if (h->gen_dtype == BV2_BF16) gen_core_bf16(c, P, z, Ty, ymask, B, L, out.o, lens);
  else gen_core(c, P, z, Ty, ymask, B, L, out.o, lens);
  return c.rc;
}

int run_flow(bv2_handle* h, hipStream_t s, int B, int Ty, const float* z_p, const int64_t* y_lengths, const float* y_mask,
             const float* g, float* z, void* ws, int64_t wsb) {
  const Model& m = h->model;
  Arena A(ws, wsb);
  PlanB P = plan_b(A, m, B, Ty);
  if (!A.ok()) { h->err = "workspace too small for bv2_stage_flow"; return -5; }
  Ctx c{h, s, m, h->blob};
  const float* ymask = y_mask;
  if (!ymask) {
    c.chk(launch_seq_mask(s, y_lengths, P.ymask, B, Ty), "y_mask");
    ymask = P.ymask;
  }
  if (z != z_p)
    (void)hipMemcpyAsync(z, z_p, sizeof(float) * (size_t)B * m.cfg.inter_channels * Ty, hipMemcpyDeviceToDevice, s);
  phase_b_gemv(c, P, g, B);
  flow_core(c, P, z, ymask, g, B, Ty);
  return c.rc;
}

int run_generator(bv2_handle* h, hipStream_t s, int B, int Ty, int L, const float* z, const int64_t* y_lengths,
                  const float* g, float* o, void* ws, int64_t wsb) {
  const Model& m = h->model;
  if (L < 1 || L > Ty) { h->err = "bv2_stage_generator: need 1 <= L <= Ty"; return -1; }
  Arena A(ws, wsb);
  PlanB P = plan_b(A, m, B, Ty);
  if (!A.ok()) { h->err = "workspace too small for bv2_stage_generator"; return -5; }
  Ctx c{h, s, m, h->blob};
  const float* ymask = nullptr;                    // y_lengths == null: z_in is taken as it is (the exported dec graph)
  if (y_lengths) {
    c.chk(launch_seq_mask(s, y_lengths, P.ymask, B, Ty), "y_mask");
    ymask = P.ymask;
  }
  phase_b_gemv(c, P, g, B);
  if (h->gen_dtype == BV2_BF16) gen_core_bf16(c, P, z, Ty, ymask, B, L, o, nullptr);
  else gen_core(c, P, z, Ty, ymask, B, L, o, nullptr);
  return c.rc;
}

}  // namespace bv2

// Host side of libldm_hip.so: the launch sequence of ONE denoiser pass over a chunk of layouts, per numerics mode
// (CategoricalTransformer.forward, nn_lib.py:191-237 + transformer_utils.py:165-246).
#include "ldm_handle.h"
#include "ldm_pack.h"

using namespace ldm_host;

// ------------------------------------------------------------------------------------------ one pass
double ldm_host::gemm_flops(int M, int N, int K) { return 2.0 * M * N * K; }

// fast mode on the reference's backbone: the stack kernel (kernels_stack.hip) — ONE launch for all layers and the
// vocabulary head, a layout's rows in its workgroup's out-projection accumulators from the embedding output to the
// logits.  Normalisation is deferred into the kernel (no LayerNorm launch, no LN output tensor): the embedding writes raw
// rows, the kernel computes its own row statistics.  (The one-launch reverse loop, run_loop_fused, does not come here: it
// gathers the embedding itself.)
static int denoise_chunk_stack(ldm_handle* h, const int32_t* d_tokens, int t, int Bc, hipStream_t st, bool skip_embed) {
  const int M = Bc * h->S, D = h->D, F = h->F;
  if (!skip_embed) {  // x0 = emb[token] + pos -> P (raw)   (skipped when the previous step's posterior wrote P)
    LnArgs a{};
    a.tokens = d_tokens; a.emb = h->emb; a.pos = h->pos; a.y32 = h->P; a.stats_out = h->stats_a; a.raw = 1;
    a.M = M; a.D = D; a.S = h->S; a.ld16 = h->Dq;
    ldm_handle::Scope sc(h, st, "embed_stats", 0, (double)M * D * 8);
    launch_layernorm(a, st);
  }
  FusedLayerSet ls{};
  ls.n_layer = h->L;
  for (int i = 0; i < h->L; ++i) {
    const LayerW& w = h->layers[i];
    const float* ss = h->adaln + ((size_t)t * h->L + i) * 2 * D;
    ls.w[i] = FusedLayerW{h->fast[i].attn_head_img_ks, h->fast[i].b_in, ss, ss + D, h->fast[i].b_out_v,
                          h->fast[i].ffn_img_pipe, w.b1, w.b2, w.g2, w.be2};
  }
  const StackHead hd{h->head_img_ks, h->head_g, h->head_b, h->logits, h->Cp, h->Cp / 32};
  ldm_handle::Scope sc(h, st, "layers_fused",
                       h->L * (gemm_flops(M, 3 * D, D) + 4.0 * Bc * h->H * (double)h->S * h->S * h->dh +
                               gemm_flops(M, D, D) + 2 * gemm_flops(M, F, D)) + gemm_flops(M, h->C, D),
                       (double)M * (D * 4 + h->Cp * 4));
  launch_stack_stream(ls, F, h->P, D, Bc, h->S, h->H, h->dh, hd, st);
  return 0;
}

// fast mode, every other accepted geometry: fp16 LDS-DMA GEMMs + MFMA attention on the head-padded layout
static int denoise_chunk_fast(ldm_handle* h, const int32_t* d_tokens, int t, int Bc, hipStream_t st,
                              bool skip_embed = false) {
  if (h->fused_attn == 6) return denoise_chunk_stack(h, d_tokens, t, Bc, st, skip_embed);
  const int M = Bc * h->S, D = h->D, F = h->F, C = h->C, Dq = h->Dq, HD = h->HD, Fq = h->Fq;
  auto gemm = [&](const char* name, int tag, const __half* A, int lda, int K, const __half* W, int ldw, int N,
                  const float* bias, int relu, const float* res, float* C32, int ldc32, __half* C16, int ldc16,
                  double flops, double bytes) {
    GemmArgs g{};
    g.A = A; g.W = W; g.bias = bias; g.relu = relu; g.res = res; g.ldres = D;
    g.C32 = C32; g.ldc32 = ldc32; g.C16 = C16; g.ldc16 = ldc16;
    const int cfg = h->gemm_cfg[tag];
    g.M = M; g.N = N; g.K = round_up(K, gemm16_block_k(cfg)); g.lda = lda; g.ldw = ldw; g.precision = 1;
    ldm_handle::Scope sc(h, st, name, flops, bytes);
    launch_gemm16(g, cfg, tag, st);
  };
  for (int i = 0; i < h->L; ++i) {
    const LayerW& w = h->layers[i];
    const ldm_handle::FastLayer& f = h->fast[i];
    const float* ss = h->adaln + ((size_t)t * h->L + i) * 2 * D;
    {
      LnArgs a{};
      a.x = h->P; a.tokens = (i == 0) ? d_tokens : nullptr; a.emb = h->emb; a.pos = h->pos;
      a.p0 = ss; a.p1 = ss + D; a.y32 = h->P; a.y16 = h->a16;
      a.M = M; a.D = D; a.S = h->S; a.ld16 = Dq; a.ada = 1;
      ldm_handle::Scope sc(h, st, i == 0 ? "embed_adaln" : "adaln", 0, (double)M * D * 10);
      launch_layernorm(a, st);
    }
    gemm("gemm_qkv", 0, h->a16, Dq, D, f.w_in, Dq, 3 * HD, f.b_in, 0, nullptr, nullptr, 0, h->qkv16,
         3 * HD, gemm_flops(M, 3 * D, D), (double)M * (D * 2 + 3 * HD * 2));
    {
      ldm_handle::Scope sc(h, st, "attention", 4.0 * Bc * h->H * (double)h->S * h->S * h->dh, (double)M * (3 * HD + HD) * 2);
      launch_attention16(h->qkv16, h->att16, Bc, h->S, h->H, h->dh, 3 * HD, HD, st);
    }
    gemm("gemm_attn_out", 1, h->att16, HD, HD, f.w_out, HD, D, w.b_out, 0, h->P, h->Q, D, nullptr, 0,
         gemm_flops(M, D, D), (double)M * (HD * 2 + D * 8));
    {
      LnArgs a{};
      a.x = h->Q; a.p0 = w.g2; a.p1 = w.be2; a.y16 = h->h16;
      a.M = M; a.D = D; a.S = h->S; a.ld16 = Dq; a.ada = 0;
      ldm_handle::Scope sc(h, st, "layernorm2", 0, (double)M * D * 6);
      launch_layernorm(a, st);
    }
    gemm("gemm_ffn1", 2, h->h16, Dq, D, f.w1, Dq, F, w.b1, 1, nullptr, nullptr, 0, h->hid16, Fq,
         gemm_flops(M, F, D), (double)M * (D * 2 + F * 2));
    gemm("gemm_ffn2", 3, h->hid16, Fq, F, f.w2, Fq, D, w.b2, 0, h->Q, h->P, D, nullptr, 0,
         gemm_flops(M, D, F), (double)M * (F * 2 + D * 8));
  }
  {
    LnArgs a{};
    a.x = h->P; a.p0 = h->head_g; a.p1 = h->head_b; a.y16 = h->h16;
    a.M = M; a.D = D; a.S = h->S; a.ld16 = Dq; a.ada = 0;
    ldm_handle::Scope sc(h, st, "layernorm_head", 0, (double)M * D * 6);
    launch_layernorm(a, st);
  }
  gemm("gemm_head", 4, h->h16, Dq, D, h->fast_head, Dq, h->Cp, nullptr, 0, nullptr, h->logits, h->Cp,
       nullptr, 0, gemm_flops(M, C, D), (double)M * (D * 2 + C * 4));
  return 0;
}

// denoiser forward for `Bc` layouts whose tokens start at d_tokens -> h->logits [Bc*S, Cp]
// exact: fp32 MFMA tiles; split: the fp16 x 3 LDS-DMA GEMM (kernels_gemm16.hip gemm16x3_k; LDM_DEV=1 LDM_SPLIT_GEMM=old: the
// register-staged r03 kernel, kept as a cross-check)
static void launch_gemm_mode(const GemmArgs& g, int tag, hipStream_t st) {
  static const bool old_split = knob_env("LDM_SPLIT_GEMM") && std::string(knob_env("LDM_SPLIT_GEMM")) == "old";
  if (g.precision == LDM_PREC_SPLIT_F16 && !old_split && g.K % 32 == 0) launch_gemm16x3(g, tag, st);
  else launch_gemm(g, st);
}

// lngemm level 2: linear2 of layer `w` (+ bias + the residual Q) as the GEMM prologue of the launch that normalises its sum
static void ffn2_prologue(ldm_handle* h, const LayerW& w, LnGemmArgs& a) {
  a.preA = h->hid16; a.preAlo = h->hid16lo; a.pre_lda = h->Fp; a.pre_astages = h->Fp / 32; a.pre_stages = ldm_pack::x3_slab_stages(h->Fp);
  a.pre_img = (const char*)w.x3_ffn2_slab; a.pre_bias = w.b2; a.pre_scale = w.s2;
  a.pre_res = h->Q; a.pre_out = nullptr;
  a.pre_panel_stride = h->hid_panels ? h->panel_rows * 64 : 0;
  a.np_pre = h->np_ffn;
  if (h->np_ffn == 1) a.preAlo = nullptr;   // (plain-fp16 hidden activations: linear1 wrote no lo panels)
}

int ldm_host::denoise_chunk(ldm_handle* h, const int32_t* d_tokens, int t, int Bc, hipStream_t st, bool skip_embed) {
  if (h->cfg.precision == LDM_PREC_FAST_F16) return denoise_chunk_fast(h, d_tokens, t, Bc, st, skip_embed);
  const int M = Bc * h->S, D = h->D, F = h->F, C = h->C, Dp = h->Dp, Fp = h->Fp;
  const int prec = h->cfg.precision;
  const bool f16 = prec != LDM_PREC_EXACT_F32;
  const bool split = prec == LDM_PREC_SPLIT_F16;
  const size_t esz = f16 ? 2 : 4;
  for (int i = 0; i < h->L; ++i) {
    const LayerW& w = h->layers[i];
    const float* ss = h->adaln + ((size_t)t * h->L + i) * 2 * D;
    if (split && h->lngemm) {  // AdaLN + QKV projection in ONE row-resident launch (kernels_lngemm.hip); P <- normed x
      LnGemmArgs a{};
      a.x = h->P; a.ldx = D;
      a.tokens = (i == 0) ? d_tokens : nullptr;
      a.emb = h->emb; a.pos = h->pos; a.S = h->S;
      a.p0 = ss; a.p1 = ss + D; a.ada = 1;
      a.y32 = h->P;
      a.out_scale = w.s_in;
      a.M = M; a.D = D; a.np_main = h->np_w;
      if (h->attnout) {   // q / k / v as head-padded hi / lo panels for the fused attention + out_proj launch (kernels_attnout.hip)
        a.img = (const char*)w.x3_qkv_pad; a.n_tiles = 3 * h->H * 2;
        a.bias = w.b_in_pad; a.N = 3 * h->H * 64;
        a.C16 = h->qkvp_hi; a.C16lo = h->qkvp_lo; a.panel_out = 1; a.panel_stride = h->panel_rows * 64;
      } else {
        a.img = (const char*)w.x3_qkv; a.n_tiles = h->x3_qkv_tiles;
        a.bias = w.b_in; a.N = 3 * D;
        a.C32 = h->qkv32; a.ldc32 = 3 * D;
      }
      const bool pre = h->pre_ffn2 && !h->ffn_fused && i > 0;   // x = Q + hid · W2^T + b2 of the PREVIOUS layer, computed in this launch (never stored)
      if (pre) ffn2_prologue(h, h->layers[i - 1], a);
      ldm_handle::Scope sc(h, st, pre ? "gemm_ffn2_qkv_ln" : "gemm_qkv_ln", gemm_flops(M, 3 * D, D) + (pre ? gemm_flops(M, D, F) : 0.0),
                           (double)M * D * 8 + (double)M * 3 * D * 4 + (pre ? (double)M * F * 4 : 0.0));
      if (launch_lngemm16x3(a, st)) return h->fail(-4, "row-resident LayerNorm + GEMM: geometry not supported");
    } else {
      {  // AdaLN (layer 0: fused with the embedding gather); P <- normed x (the residual base)
        LnArgs a{};
        a.x = h->P;
        a.tokens = (i == 0) ? d_tokens : nullptr;
        a.emb = h->emb;
        a.pos = h->pos;
        a.p0 = ss;
        a.p1 = ss + D;
        a.y32 = h->P;
        a.y16 = f16 ? h->a16 : nullptr;
        a.y16lo = split ? h->a16lo : nullptr;
        a.M = M; a.D = D; a.S = h->S; a.ld16 = Dp; a.ada = 1;
        ldm_handle::Scope sc(h, st, i == 0 ? "embed_adaln" : "adaln", 0, (double)M * D * (4 + 4 + (f16 ? 2 : 0)));
        launch_layernorm(a, st);
      }
      {  // QKV projection
        GemmArgs g{};
        g.A = f16 ? (const void*)h->a16 : (const void*)h->P;
        g.Alo = h->a16lo;
        g.W = f16 ? (const void*)w.w_in16 : (const void*)w.w_in;
        g.Wlo = w.w_in16lo; g.out_scale = w.s_in;
        g.bias = w.b_in;
        g.C32 = (prec == LDM_PREC_FAST_F16) ? nullptr : h->qkv32;
        g.C16 = (prec == LDM_PREC_FAST_F16) ? h->qkv16 : nullptr;
        g.M = M; g.N = 3 * D; g.K = f16 ? Dp : D; g.lda = f16 ? Dp : D; g.ldw = f16 ? Dp : D;
        g.ldc32 = 3 * D; g.ldc16 = 3 * D; g.precision = prec;
        ldm_handle::Scope sc(h, st, "gemm_qkv", gemm_flops(M, 3 * D, D), (double)M * D * esz + (double)M * 3 * D * (prec == 1 ? 2 : 4));
        launch_gemm_mode(g, 0, st);
      }
    }
    if (split && h->attnout) {  // attention + out-proj + residual in ONE layout-resident launch:  Q = P + softmax(q k^T) v · Wo^T + bo
      AttnOutArgs a{};
      a.qkv_hi = (const char*)h->qkvp_hi; a.qkv_lo = (const char*)h->qkvp_lo; a.panel_stride = h->panel_rows * 64;
      a.w_img = (const char*)w.x3_out_kstep;
      a.res = h->P; a.bias = w.b_out; a.out = h->Q;
      a.S = h->S; a.D = D; a.scale = 1.0f / sqrtf((float)h->dh); a.out_scale = w.s_out; a.w2 = h->w2p;
      if (h->attn_ffn_fused) {   // hybrid: the block's plain-fp16 FFN behind the attention in the same launch; P receives x + attention + FFN
        a.ffn_img = (const char*)w.ffn16_img; a.ffn_gamma = w.g2; a.ffn_beta = w.be2; a.ffn_b1 = w.b1; a.ffn_b2 = w.b2;
        a.ffn_out = h->P; a.F = F; a.n_chunks = F / 32;
      }
      ldm_handle::Scope sc(h, st, h->attn_ffn_fused ? "attn_out_ffn_fused" : "attn_out_fused",
                           4.0 * Bc * h->H * (double)h->S * h->S * h->dh + gemm_flops(M, D, D) + (h->attn_ffn_fused ? gemm_flops(M, F, D) + gemm_flops(M, D, F) : 0.0),
                           (double)M * 3 * h->H * 64 * 4 + (double)M * D * 8);
      if (launch_attnout16x3(a, Bc, st)) return h->fail(-4, "fused attention + out_proj: geometry not supported");
    } else {
    {  // attention
      AttnArgs a{};
      a.in_f16 = (prec == LDM_PREC_FAST_F16);
      a.qkv = a.in_f16 ? (const void*)h->qkv16 : (const void*)h->qkv32;
      a.out32 = f16 ? nullptr : h->att32;
      a.out16 = f16 ? h->att16 : nullptr;
      a.out16lo = split ? h->att16lo : nullptr;
      a.B = Bc; a.S = h->S; a.H = h->H; a.dh = h->dh; a.D = D; a.ld = 3 * D; a.ldo32 = D; a.ldo16 = Dp;
      ldm_handle::Scope sc(h, st, "attention", 4.0 * Bc * h->H * (double)h->S * h->S * h->dh,
                           (double)M * 3 * D * (a.in_f16 ? 2 : 4) + (double)M * D * esz);
      launch_attention(a, st);
    }
    if (!(split && h->pre_out)) {  // out-proj + residual onto the normed x:  Q = P + att·Wo^T + bo
      GemmArgs g{};
      g.A = f16 ? (const void*)h->att16 : (const void*)h->att32;
      g.Alo = h->att16lo;
      g.W = f16 ? (const void*)w.w_out16 : (const void*)w.w_out;
      g.Wlo = w.w_out16lo; g.out_scale = w.s_out;
      g.bias = w.b_out;
      g.res = h->P; g.ldres = D;
      g.C32 = h->Q; g.ldc32 = D;
      g.M = M; g.N = D; g.K = f16 ? Dp : D; g.lda = f16 ? Dp : D; g.ldw = f16 ? Dp : D; g.precision = prec;
      ldm_handle::Scope sc(h, st, "gemm_attn_out", gemm_flops(M, D, D), (double)M * D * (esz + 8));
      launch_gemm_mode(g, 1, st);
    }
    }
    if (split && h->attn_ffn_fused) {
      // (the FFN ran behind the attention: kernels_attnout.hip FFN)
    } else if (split && h->ffn_fused) {   // hybrid: P = Q + b2 + W2 relu(W1 LN2(Q) + b1) in ONE plain-fp16 launch (kernels_ffn16.hip); the hidden activations stay on chip
      FfnRowsArgs a{};
      a.x = h->Q; a.out = h->P;
      a.gamma = w.g2; a.beta = w.be2; a.b1 = w.b1; a.b2 = w.b2;
      a.img = (const char*)w.ffn16_img;
      a.M = M; a.D = D; a.F = F; a.n_chunks = F / 32;
      ldm_handle::Scope sc(h, st, "ffn_fused16", gemm_flops(M, F, D) + gemm_flops(M, D, F), (double)M * D * 8);
      if (launch_ffn16_rows(a, st)) return h->fail(-4, "fused fp16 FFN: geometry not supported");
    } else if (split && h->lngemm) {  // LayerNorm 2 + FFN1 + ReLU in ONE row-resident launch: hi / lo hidden activations out
      LnGemmArgs a{};
      a.x = h->Q; a.ldx = D;
      a.p0 = w.g2; a.p1 = w.be2; a.ada = 0;
      a.img = (const char*)w.x3_ffn1; a.n_tiles = h->x3_ffn1_tiles;
      a.bias = w.b1; a.out_scale = w.s1; a.relu = 1;
      a.C16 = h->hid16; a.C16lo = h->hid16lo; a.ldc16 = Fp;
      if (h->hid_panels) { a.panel_out = 1; a.panel_stride = h->panel_rows * 64; }   // (read back by ffn2_prologue in the same form)
      a.M = M; a.N = F; a.D = D; a.S = h->S; a.np_main = h->np_ffn;
      if (h->np_ffn == 1) a.C16lo = nullptr;   // hybrid: ReLU output rounded once (panel-major: ldm_create requires hid_panels for it)
      if (h->pre_out) {   // Q = P + att · Wo^T + bo computed in this launch, written once (linear2's residual base)
        a.preA = h->att16; a.preAlo = h->att16lo; a.pre_lda = Dp; a.pre_astages = Dp / 32; a.pre_stages = ldm_pack::x3_slab_stages(Dp);
        a.pre_img = (const char*)w.x3_out_slab; a.pre_bias = w.b_out; a.pre_scale = w.s_out;
        a.pre_res = h->P; a.pre_out = h->Q;
      }
      ldm_handle::Scope sc(h, st, h->pre_out ? "gemm_out_ffn1_ln" : "gemm_ffn1_ln", gemm_flops(M, F, D) + (h->pre_out ? gemm_flops(M, D, D) : 0.0),
                           (double)M * D * 4 + (double)M * F * 4 + (h->pre_out ? (double)M * D * 12 : 0.0));
      if (launch_lngemm16x3(a, st)) return h->fail(-4, "row-resident LayerNorm + GEMM: geometry not supported");
    } else {
      {  // LayerNorm 2
        LnArgs a{};
        a.x = h->Q; a.p0 = w.g2; a.p1 = w.be2;
        a.y32 = f16 ? nullptr : h->h32;
        a.y16 = f16 ? h->h16 : nullptr;
        a.y16lo = split ? h->h16lo : nullptr;
        a.M = M; a.D = D; a.S = h->S; a.ld16 = Dp; a.ada = 0;
        ldm_handle::Scope sc(h, st, "layernorm2", 0, (double)M * D * (4 + esz));
        launch_layernorm(a, st);
      }
      {  // FFN1 + ReLU
        GemmArgs g{};
        g.A = f16 ? (const void*)h->h16 : (const void*)h->h32;
        g.Alo = h->h16lo;
        g.W = f16 ? (const void*)w.w1_16 : (const void*)w.w1;
        g.Wlo = w.w1_16lo; g.out_scale = w.s1;
        g.bias = w.b1; g.relu = 1;
        g.C32 = f16 ? nullptr : h->hid32; g.ldc32 = F;
        g.C16 = f16 ? h->hid16 : nullptr; g.C16lo = split ? h->hid16lo : nullptr; g.ldc16 = Fp;
        g.M = M; g.N = F; g.K = f16 ? Dp : D; g.lda = f16 ? Dp : D; g.ldw = f16 ? Dp : D; g.precision = prec;
        ldm_handle::Scope sc(h, st, "gemm_ffn1", gemm_flops(M, F, D), (double)M * D * esz + (double)M * F * esz);
        launch_gemm_mode(g, 2, st);
      }
    }
    if (!(split && (h->pre_ffn2 || h->ffn_fused))) {  // FFN2 + residual:  P = Q + hid·W2^T + b2   (level 2: the prologue of the next AdaLN + in_proj launch / of the head)
      GemmArgs g{};
      g.A = f16 ? (const void*)h->hid16 : (const void*)h->hid32;
      g.Alo = h->hid16lo;
      g.W = f16 ? (const void*)w.w2_16 : (const void*)w.w2;
      g.Wlo = w.w2_16lo; g.out_scale = w.s2;
      g.bias = w.b2;
      g.res = h->Q; g.ldres = D;
      g.C32 = h->P; g.ldc32 = D;
      g.M = M; g.N = D; g.K = f16 ? Fp : F; g.lda = f16 ? Fp : F; g.ldw = f16 ? Fp : F; g.precision = prec;
      ldm_handle::Scope sc(h, st, "gemm_ffn2", gemm_flops(M, D, F), (double)M * F * esz + (double)M * D * 8);
      launch_gemm_mode(g, 3, st);
    }
  }
  if (split && h->lngemm) {  // head: LayerNorm + vocabulary projection (no bias) in ONE row-resident launch
    LnGemmArgs a{};
    a.x = h->P; a.ldx = D;
    a.p0 = h->head_g; a.p1 = h->head_b; a.ada = 0;
    a.img = (const char*)h->x3_head; a.n_tiles = h->x3_head_tiles;
    a.out_scale = h->head_s;
    a.C32 = h->logits; a.ldc32 = h->Cp;
    a.np_main = h->np_ffn;
    a.M = M; a.N = h->Cp; a.D = D; a.S = h->S;   // (columns C .. Cp of the image are zero rows: exact zeros in the padding)
    const bool hp = h->pre_ffn2 && !h->ffn_fused;
    if (hp) ffn2_prologue(h, h->layers[h->L - 1], a);
    ldm_handle::Scope sc(h, st, hp ? "gemm_ffn2_head_ln" : "gemm_head_ln", gemm_flops(M, C, D) + (hp ? gemm_flops(M, D, F) : 0.0),
                         (double)M * D * 4 + (double)M * C * 4 + (hp ? (double)M * F * 4 : 0.0));
    if (launch_lngemm16x3(a, st)) return h->fail(-4, "row-resident LayerNorm + GEMM: geometry not supported");
    return 0;
  }
  {  // head: LayerNorm + vocab projection (no bias)
    LnArgs a{};
    a.x = h->P; a.p0 = h->head_g; a.p1 = h->head_b;
    a.y32 = f16 ? nullptr : h->h32;
    a.y16 = f16 ? h->h16 : nullptr;
    a.y16lo = split ? h->h16lo : nullptr;
    a.M = M; a.D = D; a.S = h->S; a.ld16 = Dp; a.ada = 0;
    {
      ldm_handle::Scope sc(h, st, "layernorm_head", 0, (double)M * D * (4 + esz));
      launch_layernorm(a, st);
    }
    GemmArgs g{};
    g.A = f16 ? (const void*)h->h16 : (const void*)h->h32;
    g.Alo = h->h16lo;
    g.W = f16 ? (const void*)h->head_w16 : (const void*)h->head_w;
    g.Wlo = h->head_w16lo; g.out_scale = h->head_s;
    g.C32 = h->logits; g.ldc32 = h->Cp;
    g.M = M; g.N = C; g.K = f16 ? Dp : D; g.lda = f16 ? Dp : D; g.ldw = f16 ? Dp : D; g.precision = prec;
    ldm_handle::Scope sc(h, st, "gemm_head", gemm_flops(M, C, D), (double)M * D * esz + (double)M * C * 4);
    launch_gemm_mode(g, 4, st);
  }
  return 0;
}


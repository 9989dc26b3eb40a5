// Host side of libldm_hip.so: the reference checkpoint -> device weights (ldm_load_weight / ldm_finalize_weights): fp32 views,
// fp16 / split copies, the LDS weight images and parameter tables of the layout-resident kernels (ldm_pack.h).
#include "ldm_handle.h"

#include <functional>

using namespace ldm_host;

// ------------------------------------------------------------------------------------------ weights
static std::string strip_prefix(const char* key) {
  std::string k(key);
  for (const char* p : {"model.module.", "module.", "model."}) {
    const size_t n = strlen(p);
    if (k.compare(0, n, p) == 0) {
      k = k.substr(n);
      break;
    }
  }
  return k;
}

extern "C" int ldm_load_weight(ldm_handle* h, const char* key, const float* h_data, const int64_t* shape, int ndim) {
  if (!h || !key || !h_data || (ndim > 0 && !shape)) return h ? h->fail(-1, "null argument") : -1;
  ON_DEVICE(h);
  const std::string k = strip_prefix(key);
  Raw r;
  r.shape.assign(shape, shape + ndim);
  const int64_t n = r.numel();
  if (n <= 0) return h->fail(-1, "empty tensor for key %s", key);
  auto it = h->raw.find(k);
  if (it != h->raw.end()) {
    if (it->second.numel() != n) return h->fail(-1, "key %s reloaded with a different size", key);
    r.d = it->second.d;
  } else {
    HIP_OK(h, hipMalloc((void**)&r.d, n * sizeof(float)));
  }
  HIP_OK(h, hipMemcpy(r.d, h_data, n * sizeof(float), hipMemcpyHostToDevice));
  h->raw[k] = r;
  h->finalized = false;
  return 0;
}

static int need(ldm_handle* h, const std::string& key, std::initializer_list<int64_t> shape, const float** out) {
  auto it = h->raw.find(key);
  if (it == h->raw.end()) return h->fail(-4, "missing checkpoint key: %s", key.c_str());
  std::vector<int64_t> want(shape);
  if (it->second.shape != want) {
    std::string got;
    for (auto s : it->second.shape) got += std::to_string(s) + ",";
    return h->fail(-4, "checkpoint key %s has shape (%s) — does not match the configured geometry", key.c_str(),
                   got.c_str());
  }
  *out = it->second.d;
  return 0;
}

// fp16 (and split-lo) copy of a [N,K] weight with the K axis zero-padded to Kp
// out_scale (split mode): the weights are multiplied by the exact power of two 2^k that brings max |w| into [1, 2) before the hi / lo
// split — lo is stored unscaled (kSplitLoScale = 1) and fp16's denormal spacing 2^-24 must be small against the weights — and
// *out_scale = 2^-k goes to the GEMM's epilogue.
static int make_w16(ldm_handle* h, const float* w, int N, int K, int Kp, __half** hi, __half** lo, float* out_scale = nullptr) {
  const bool split = h->cfg.precision == LDM_PREC_SPLIT_F16;
  float scale = 1.0f;
  if (split) {
    std::vector<float> host((size_t)N * K);
    HIP_OK(h, hipMemcpy(host.data(), w, host.size() * sizeof(float), hipMemcpyDeviceToHost));
    float mx = 0.f;
    for (float x : host) mx = std::max(mx, std::fabs(x));
    if (mx > 0.f && std::isfinite(mx)) scale = std::ldexp(1.0f, -(int)std::floor(std::log2(mx)));
  }
  if (out_scale) *out_scale = 1.0f / scale;
  // (rows padded with zeros to whole 128-row tiles: the LDS-DMA GEMMs load W without bounds checks)
  const size_t Nt = (size_t)round_up(N, 256);
  int rc = h->dalloc(hi, Nt * Kp);
  if (rc) return rc;
  if (split && (rc = h->dalloc(lo, Nt * Kp))) return rc;
  if (K == Kp) {
    launch_f32_to_f16(w, *hi, split ? *lo : nullptr, (int64_t)N * K, 0, scale);
  } else {
    __half *thi = nullptr, *tlo = nullptr;
    if ((rc = h->dalloc(&thi, (size_t)N * K))) return rc;
    if (split && (rc = h->dalloc(&tlo, (size_t)N * K))) return rc;
    launch_f32_to_f16(w, thi, tlo, (int64_t)N * K, 0, scale);
    HIP_OK(h, hipMemcpy2DAsync(*hi, (size_t)Kp * 2, thi, (size_t)K * 2, (size_t)K * 2, N, hipMemcpyDeviceToDevice, 0));
    if (split)
      HIP_OK(h, hipMemcpy2DAsync(*lo, (size_t)Kp * 2, tlo, (size_t)K * 2, (size_t)K * 2, N, hipMemcpyDeviceToDevice, 0));
    // (the unpadded temporaries are not kept until the next finalize)
    HIP_OK(h, hipStreamSynchronize(0));
    for (__half* t : {thi, tlo}) {
      if (!t) continue;
      auto& v = h->to_derived ? h->derived : h->owned;
      v.erase(std::remove(v.begin(), v.end(), (void*)t), v.end());
      (void)hipFree(t);
    }
  }
  return 0;
}

// hi | lo tile image of a split-mode weight for the row-resident x3 GEMM (kernels_lngemm.hip): the [>= 32 n_tiles][Kp = 512] hi / lo
// copies make_w16 built (natural K order) -> K axis in k-slot order -> ldm_pack::pack_x3_tile_image
// rowmap (optional): image row rowmap(r) <- weight row r for the n_src first rows, every other image row zero (the head-padded in_proj
// in front of kernels_attnout.hip: ldm_pack::qkv_row)
static int make_x3_image(ldm_handle* h, const __half* hi, const __half* lo, int n_tiles, void** out, int n_src = 0,
                         const std::function<int(int)>& rowmap = nullptr) {
  const size_t n = (size_t)n_tiles * 32 * 512;
  const int rows = rowmap ? n_src : n_tiles * 32;
  std::vector<uint16_t> a((size_t)rows * 512), b((size_t)rows * 512), ak(n, 0), bk(n, 0);
  HIP_OK(h, hipDeviceSynchronize());  // (the cast kernels of make_w16)
  HIP_OK(h, hipMemcpy(a.data(), hi, a.size() * 2, hipMemcpyDeviceToHost));
  HIP_OK(h, hipMemcpy(b.data(), lo, b.size() * 2, hipMemcpyDeviceToHost));
  for (int r = 0; r < rows; ++r) {
    const size_t d = (size_t)(rowmap ? rowmap(r) : r) * 512;
    for (int k = 0; k < 512; ++k) {
      ak[d + ldm_pack::kslot(k)] = a[(size_t)r * 512 + k];
      bk[d + ldm_pack::kslot(k)] = b[(size_t)r * 512 + k];
    }
  }
  const std::vector<uint16_t> img = ldm_pack::pack_x3_tile_image(ak.data(), bk.data(), n_tiles);
  __half* d = nullptr;
  int rc = h->dalloc(&d, img.size(), false);
  if (rc) return rc;
  HIP_OK(h, hipMemcpy(d, img.data(), img.size() * 2, hipMemcpyHostToDevice));
  *out = d;
  return 0;
}

// K-slab image (ldm_pack::pack_x3_slab_image) of a split-mode weight [N][ld] hi / lo for the GEMM prologue of kernels_lngemm.hip
static int make_x3_slab(ldm_handle* h, const __half* hi, const __half* lo, int N, int ld, int K, void** out) {
  const size_t n = (size_t)N * ld;
  std::vector<uint16_t> a(n), b(n);
  HIP_OK(h, hipDeviceSynchronize());  // (the cast kernels of make_w16)
  HIP_OK(h, hipMemcpy(a.data(), hi, n * 2, hipMemcpyDeviceToHost));
  HIP_OK(h, hipMemcpy(b.data(), lo, n * 2, hipMemcpyDeviceToHost));
  const std::vector<uint16_t> img = ldm_pack::pack_x3_slab_image(a.data(), b.data(), N, ld, K);
  __half* d = nullptr;
  int rc = h->dalloc(&d, img.size(), false);
  if (rc) return rc;
  HIP_OK(h, hipMemcpy(d, img.data(), img.size() * 2, hipMemcpyHostToDevice));
  *out = d;
  return 0;
}

// k-step image (ldm_pack::pack_x3_kstep_image) of out_proj's split copies [D][ld] for kernels_attnout.hip
static int make_x3_kstep(ldm_handle* h, const __half* hi, const __half* lo, int N, int ld, void** out) {
  const size_t n = (size_t)N * ld;
  std::vector<uint16_t> a(n), b(n);
  HIP_OK(h, hipDeviceSynchronize());  // (the cast kernels of make_w16)
  HIP_OK(h, hipMemcpy(a.data(), hi, n * 2, hipMemcpyDeviceToHost));
  HIP_OK(h, hipMemcpy(b.data(), lo, n * 2, hipMemcpyDeviceToHost));
  const std::vector<uint16_t> img = ldm_pack::pack_x3_kstep_image(a.data(), b.data(), N, ld, h->H, h->dh);
  __half* d = nullptr;
  int rc = h->dalloc(&d, img.size(), false);
  if (rc) return rc;
  HIP_OK(h, hipMemcpy(d, img.data(), img.size() * 2, hipMemcpyHostToDevice));
  *out = d;
  return 0;
}

// ---- fast-mode weight images (built on the host once; tiny compared with one sampling call)
static uint16_t f2h_bits(float x) {
  const __half hh = __float2half(x);
  uint16_t u;
  memcpy(&u, &hh, 2);
  return u;
}

// dst[Np][Kp] fp16 (zero filled) with dst[rmap(n)][cmap(k)] = src[n][k]
template <typename RM, typename CM>
static int pack_w16(ldm_handle* h, const float* d_src, int N, int K, int Np, int Kp, RM rmap, CM cmap, __half** out) {
  std::vector<float> src((size_t)N * K);
  HIP_OK(h, hipMemcpy(src.data(), d_src, src.size() * sizeof(float), hipMemcpyDeviceToHost));
  std::vector<uint16_t> dst((size_t)Np * Kp, 0);
  for (int n = 0; n < N; ++n) {
    const size_t ro = (size_t)rmap(n) * Kp;
    for (int k = 0; k < K; ++k) dst[ro + cmap(k)] = f2h_bits(src[(size_t)n * K + k]);
  }
  int rc = h->dalloc(out, dst.size(), false);
  if (rc) return rc;
  HIP_OK(h, hipMemcpy(*out, dst.data(), dst.size() * 2, hipMemcpyHostToDevice));
  return 0;
}

// ---- LDS-image weight streams: the stack kernel
// copies their weights global -> LDS with linear 1-KiB DMA instructions, so the global copy is stored in
// consumption order with the LDS bank swizzle already applied.
static std::vector<uint16_t> download16(ldm_handle* h, const __half* d, size_t n, int* rc) {
  std::vector<uint16_t> v(n);
  *rc = 0;
  if (hipMemcpy(v.data(), d, n * 2, hipMemcpyDeviceToHost) != hipSuccess) {
    h->err = "hipMemcpy (weight image) failed";
    *rc = -2;
  }
  return v;
}
static int upload_image(ldm_handle* h, const std::vector<uint16_t>& img, void** out) {
  __half* d = nullptr;
  int rc = h->dalloc(&d, img.size(), false);
  if (rc) return rc;
  HIP_OK(h, hipMemcpy(d, img.data(), img.size() * 2, hipMemcpyHostToDevice));
  *out = d;
  return 0;
}
// (index maps and image packers: ldm_pack.h — pure C++, unit-tested on the CPU by tests/cpu_pack_check.cpp)

static int build_fast_weights(ldm_handle* h) {
  const int D = h->D, F = h->F, C = h->C, H = h->H, dh = h->dh, HD = h->HD, Dq = h->Dq, Fq = h->Fq;
  auto id = [](int x) { return x; };
  // in_proj row n = which*D + head*dh + d  ->  (which*H + head)*64 + d   (head slices padded to 64)
  auto qkv_row = [=](int n) { return ldm_pack::qkv_row(n, D, H, dh); };
  // out_proj column k = head*dh + d -> head*64 + d (matches the attention kernel's output layout)
  auto head_col = [=](int k) { return ldm_pack::head_col(k, dh); };
  auto kslot = [](int k) { return ldm_pack::kslot(k); };
  const bool stack = h->fused_attn == 6;  // (geometry checked in ldm_create)
  h->fast.assign(h->L, ldm_handle::FastLayer{});
  int rc;
  for (int i = 0; i < h->L; ++i) {
    const LayerW& w = h->layers[i];
    ldm_handle::FastLayer& f = h->fast[i];
    if (!stack) {  // head-padded fp16 copies for the generic tiled GEMMs + attention16
      if ((rc = pack_w16(h, w.w_in, 3 * D, D, round_up(3 * HD, 256), Dq, qkv_row, id, &f.w_in))) return rc;
      if ((rc = pack_w16(h, w.w_out, D, D, round_up(D, 256), HD, id, head_col, &f.w_out))) return rc;
      if ((rc = pack_w16(h, w.w1, F, D, round_up(F, 256), Dq, id, id, &f.w1))) return rc;
      if ((rc = pack_w16(h, w.w2, D, F, round_up(D, 256), Fq, id, id, &f.w2))) return rc;
    } else {
      // LDS images of the stack kernel.  K axes in MFMA k-slot order (position 16s + 8g + e <- index 16s + 8(e>>2) + 4g +
      // (e&3)): a lane's accumulator-layout registers of column groups 2ks, 2ks+1 ARE its fragment of k16-step ks
      __half *w1p = nullptr, *w2p = nullptr, *w_in_ks = nullptr, *w_out_ks = nullptr;
      auto head_kslot = [=](int k) { return kslot(head_col(k)); };
      if ((rc = pack_w16(h, w.w1, F, D, round_up(F, 256), Dq, id, kslot, &w1p))) return rc;
      if ((rc = pack_w16(h, w.w2, D, F, round_up(D, 256), Fq, id, kslot, &w2p))) return rc;
      if ((rc = pack_w16(h, w.w_in, 3 * D, D, round_up(3 * HD, 256), Dq, qkv_row, kslot, &w_in_ks))) return rc;
      if ((rc = pack_w16(h, w.w_out, D, D, round_up(D, 256), HD, id, head_kslot, &w_out_ks))) return rc;
      const std::vector<uint16_t> h1p = download16(h, w1p, (size_t)F * Dq, &rc);
      if (rc) return rc;
      const std::vector<uint16_t> h2 = download16(h, w2p, (size_t)round_up(D, 256) * Fq, &rc);
      if (rc) return rc;
      const std::vector<uint16_t> ffn = ldm_pack::pack_ffn_image(h1p.data(), h2.data(), Fq, F, 480);
      if ((rc = upload_image(h, ldm_pack::pack_ffn_image_pipelined(ffn, F / 32), &f.ffn_img_pipe))) return rc;
      const std::vector<uint16_t> hin_ks = download16(h, w_in_ks, (size_t)3 * HD * Dq, &rc);
      if (rc) return rc;
      const std::vector<uint16_t> hout = download16(h, w_out_ks, (size_t)round_up(D, 256) * HD, &rc);
      if (rc) return rc;
      const std::vector<uint16_t> slab_ks = ldm_pack::pack_attn_slab_image(hin_ks.data(), hout.data(), H);
      if ((rc = upload_image(h, ldm_pack::pack_attn_head_image(slab_ks, H), &f.attn_head_img_ks))) return rc;
    }
    std::vector<float> b(3 * D), bp((size_t)3 * HD, 0.f);
    HIP_OK(h, hipMemcpy(b.data(), w.b_in, b.size() * 4, hipMemcpyDeviceToHost));
    for (int n = 0; n < 3 * D; ++n) bp[qkv_row(n)] = b[n];
    if ((rc = h->dalloc(&f.b_in, bp.size(), false))) return rc;
    HIP_OK(h, hipMemcpy(f.b_in, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
    if (stack) {
      // softmax rows sum to 1, so P (V + 1 b_v^T) = P V + 1 b_v^T and the V bias reaches the block output as the
      // constant W_out b_v: folded into the out-projection bias once, here (fp64 accumulate)
      std::vector<float> wo((size_t)D * D), bo(D), bov(D);
      HIP_OK(h, hipMemcpy(wo.data(), w.w_out, wo.size() * 4, hipMemcpyDeviceToHost));
      HIP_OK(h, hipMemcpy(bo.data(), w.b_out, bo.size() * 4, hipMemcpyDeviceToHost));
      for (int n = 0; n < D; ++n) {
        double acc = bo[n];
        for (int k = 0; k < D; ++k) acc += (double)wo[(size_t)n * D + k] * (double)b[2 * D + k];
        bov[n] = (float)acc;
      }
      if ((rc = h->dalloc(&f.b_out_v, bov.size(), false))) return rc;
      HIP_OK(h, hipMemcpy(f.b_out_v, bov.data(), bov.size() * 4, hipMemcpyHostToDevice));
    }
  }
  if (stack) {
    __half* hk = nullptr;
    if ((rc = pack_w16(h, h->head_w, C, D, round_up(C, 256), Dq, id, kslot, &hk))) return rc;
    const std::vector<uint16_t> hh = download16(h, hk, (size_t)round_up(C, 256) * Dq, &rc);
    if (rc) return rc;
    return upload_image(h, ldm_pack::pack_head_image(hh.data(), h->Cp / 32), &h->head_img_ks);
  }
  return pack_w16(h, h->head_w, C, D, round_up(C, 256), Dq, id, id, &h->fast_head);
}

// Parameter tables of the loop kernel as LDS images (kernels_stack.hip HEAD == 2 copies them global -> LDS with the DMA,
// one phase ahead of their use, instead of 26 loads per thread behind a barrier at every layer entry):
//   att_static[l]      kStackTblAttStatic floats   head-padded in_proj bias
//   att_dyn[t][l]      kStackTblAttDyn floats      1 + AdaLN scale | AdaLN shift | b_out + W_out b_v + shift   (0 beyond d_model)
//   ffn[l]             kStackTblFfn floats         linear1 bias (0-padded to 2048) | norm2 gamma | norm2 beta | linear2 bias
//   head               kStackTblAttDyn floats      head LayerNorm gamma | beta | 0      (takes the att_dyn slot behind the last layer)
static int build_loop_tables(ldm_handle* h) {
  const int D = h->D, F = h->F, L = h->L, T = h->T;
  if (h->fused_attn != 6 || h->H * 64 * 3 != kStackTblAttStatic || D > 512 || F > 2048) return 0;  // not on the stack kernel
  std::vector<float> ada((size_t)T * L * 2 * D);
  HIP_OK(h, hipDeviceSynchronize());  // (the AdaLN table kernels)
  HIP_OK(h, hipMemcpy(ada.data(), h->adaln, ada.size() * 4, hipMemcpyDeviceToHost));
  auto pull = [&](const float* d, size_t n, std::vector<float>& out) -> int {
    out.resize(n);
    HIP_OK(h, hipMemcpy(out.data(), d, n * 4, hipMemcpyDeviceToHost));
    return 0;
  };
  std::vector<float> att_static((size_t)L * kStackTblAttStatic, 0.f), att_dyn((size_t)T * L * kStackTblAttDyn, 0.f),
      ffn((size_t)L * kStackTblFfn, 0.f), head(kStackTblAttDyn, 0.f), v;
  int rc;
  for (int l = 0; l < L; ++l) {
    if ((rc = pull(h->fast[l].b_in, kStackTblAttStatic, v))) return rc;
    std::copy(v.begin(), v.end(), att_static.begin() + (size_t)l * kStackTblAttStatic);
    std::vector<float> bov;
    if ((rc = pull(h->fast[l].b_out_v, D, bov))) return rc;
    for (int t = 0; t < T; ++t) {
      const float* ss = &ada[((size_t)t * L + l) * 2 * D];
      float* o = &att_dyn[((size_t)t * L + l) * kStackTblAttDyn];
      for (int i = 0; i < D; ++i) {
        o[i] = 1.0f + ss[i];           // multiplier (0 beyond d_model: padded columns come out as exact zeros)
        o[512 + i] = ss[D + i];        // shift
        o[1024 + i] = bov[i] + ss[D + i];
      }
    }
    float* f = &ffn[(size_t)l * kStackTblFfn];
    if ((rc = pull(h->layers[l].b1, F, v))) return rc;
    std::copy(v.begin(), v.end(), f);
    if ((rc = pull(h->layers[l].g2, D, v))) return rc;
    std::copy(v.begin(), v.end(), f + 2048);
    if ((rc = pull(h->layers[l].be2, D, v))) return rc;
    std::copy(v.begin(), v.end(), f + 2048 + 512);
    if ((rc = pull(h->layers[l].b2, D, v))) return rc;
    std::copy(v.begin(), v.end(), f + 2048 + 1024);
  }
  if ((rc = pull(h->head_g, D, v))) return rc;
  std::copy(v.begin(), v.end(), head.begin());
  if ((rc = pull(h->head_b, D, v))) return rc;
  std::copy(v.begin(), v.end(), head.begin() + 512);
  auto push = [&](const std::vector<float>& src, float** dst) -> int {
    if (!*dst && (rc = h->dalloc(dst, src.size(), false))) return rc;
    HIP_OK(h, hipMemcpy(*dst, src.data(), src.size() * 4, hipMemcpyHostToDevice));
    return 0;
  };
  if ((rc = push(att_static, &h->tbl_att_static))) return rc;
  if ((rc = push(att_dyn, &h->tbl_att_dyn))) return rc;
  if ((rc = push(ffn, &h->tbl_ffn))) return rc;
  return push(head, &h->tbl_head);
}

extern "C" int ldm_finalize_weights(ldm_handle* h) {
  if (!h) return -1;
  ON_DEVICE(h);
  const int D = h->D, F = h->F, C = h->C, T = h->T, L = h->L;
  const std::string tr = "transformer.";
  int rc;
  const float *elem = nullptr, *attr = nullptr;
  if ((rc = need(h, tr + "cat_emb.weight", {C, D}, &h->emb))) return rc;
  if ((rc = need(h, tr + "pos_emb.elem_emb", {h->cfg.max_elem, D}, &elem))) return rc;
  if ((rc = need(h, tr + "pos_emb.attr_emb", {h->cfg.n_attr, D}, &attr))) return rc;
  if ((rc = need(h, tr + "head.0.weight", {D}, &h->head_g))) return rc;
  if ((rc = need(h, tr + "head.0.bias", {D}, &h->head_b))) return rc;
  if ((rc = need(h, tr + "head.1.weight", {C, D}, &h->head_w))) return rc;
  if (!h->pos && (rc = h->dalloc(&h->pos, (size_t)h->S * D))) return rc;
  if (!h->adaln && (rc = h->dalloc(&h->adaln, (size_t)T * L * 2 * D))) return rc;
  // everything below is derived from the checkpoint: drop what an earlier finalize built (nothing may still be running on it),
  // then collect the new allocations in h->derived
  HIP_OK(h, hipDeviceSynchronize());
  h->finalized = false;   // (a finalize that fails part-way must not leave an earlier success standing over freed images)
  for (auto& g : h->graphs) g.destroy();
  h->graphs.clear();
  for (void* p : h->derived) (void)hipFree(p);
  h->derived.clear();
  h->fast.clear();
  h->fast_head = nullptr;
  h->head_img_ks = nullptr;
  h->head_w16 = h->head_w16lo = nullptr;
  h->x3_head = nullptr;
  h->tbl_att_static = h->tbl_att_dyn = h->tbl_ffn = h->tbl_head = nullptr;
  struct Sink {
    ldm_handle* h;
    explicit Sink(ldm_handle* h_) : h(h_) { h->to_derived = true; }
    ~Sink() { h->to_derived = false; }
  } sink(h);
  launch_pos_table(elem, attr, h->pos, h->cfg.max_elem, h->cfg.n_attr, D, 0);
  const bool f16 = h->cfg.precision != LDM_PREC_EXACT_F32;
  h->layers.assign(L, LayerW{});
  for (int i = 0; i < L; ++i) {
    const std::string b = tr + "backbone.layers." + std::to_string(i) + ".";
    LayerW& w = h->layers[i];
    const float *emb_t = nullptr, *lin_w = nullptr, *lin_b = nullptr;
    if ((rc = need(h, b + "self_attn.in_proj_weight", {3 * D, D}, &w.w_in))) return rc;
    if ((rc = need(h, b + "self_attn.in_proj_bias", {3 * D}, &w.b_in))) return rc;
    if ((rc = need(h, b + "self_attn.out_proj.weight", {D, D}, &w.w_out))) return rc;
    if ((rc = need(h, b + "self_attn.out_proj.bias", {D}, &w.b_out))) return rc;
    if ((rc = need(h, b + "linear1.weight", {F, D}, &w.w1))) return rc;
    if ((rc = need(h, b + "linear1.bias", {F}, &w.b1))) return rc;
    if ((rc = need(h, b + "linear2.weight", {D, F}, &w.w2))) return rc;
    if ((rc = need(h, b + "linear2.bias", {D}, &w.b2))) return rc;
    if ((rc = need(h, b + "norm1.emb.weight", {T, D}, &emb_t))) return rc;
    if ((rc = need(h, b + "norm1.linear.weight", {2 * D, D}, &lin_w))) return rc;
    if ((rc = need(h, b + "norm1.linear.bias", {2 * D}, &lin_b))) return rc;
    if ((rc = need(h, b + "norm2.weight", {D}, &w.g2))) return rc;
    if ((rc = need(h, b + "norm2.bias", {D}, &w.be2))) return rc;
    launch_adaln_table(emb_t, lin_w, lin_b, h->adaln, T, D, L, i, 0);
    if (f16 && h->cfg.precision != LDM_PREC_FAST_F16) {
      if ((rc = make_w16(h, w.w_in, 3 * D, D, h->Dp, &w.w_in16, &w.w_in16lo, &w.s_in))) return rc;
      if ((rc = make_w16(h, w.w_out, D, D, h->Dp, &w.w_out16, &w.w_out16lo, &w.s_out))) return rc;
      if ((rc = make_w16(h, w.w1, F, D, h->Dp, &w.w1_16, &w.w1_16lo, &w.s1))) return rc;
      if ((rc = make_w16(h, w.w2, D, F, h->Fp, &w.w2_16, &w.w2_16lo, &w.s2))) return rc;
      if (h->lngemm) {
        if ((rc = make_x3_image(h, w.w_in16, w.w_in16lo, h->x3_qkv_tiles, &w.x3_qkv))) return rc;
        if ((rc = make_x3_image(h, w.w1_16, w.w1_16lo, h->x3_ffn1_tiles, &w.x3_ffn1))) return rc;
      }
      if (h->attnout) {   // in_proj with head-padded output columns (+ bias), out_proj as per-(head, k16-step) stages
        const int H = h->H, dh = h->dh;
        if ((rc = make_x3_image(h, w.w_in16, w.w_in16lo, 3 * H * 2, &w.x3_qkv_pad, 3 * D, [=](int n) { return ldm_pack::qkv_row(n, D, H, dh); }))) return rc;
        std::vector<float> b(3 * D), bp((size_t)3 * H * 64, 0.f);
        HIP_OK(h, hipMemcpy(b.data(), w.b_in, b.size() * 4, hipMemcpyDeviceToHost));
        for (int n = 0; n < 3 * D; ++n) bp[ldm_pack::qkv_row(n, D, H, dh)] = b[n];
        if ((rc = h->dalloc(&w.b_in_pad, bp.size(), false))) return rc;
        HIP_OK(h, hipMemcpy(w.b_in_pad, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
        if ((rc = make_x3_kstep(h, w.w_out16, w.w_out16lo, D, h->Dp, &w.x3_out_kstep))) return rc;
      }
      if (h->pre_out && (rc = make_x3_slab(h, w.w_out16, w.w_out16lo, D, h->Dp, h->Dp, &w.x3_out_slab))) return rc;
      if (h->pre_ffn2 && (rc = make_x3_slab(h, w.w2_16, w.w2_16lo, D, h->Fp, h->Fp, &w.x3_ffn2_slab))) return rc;
      if (h->ffn_fused) {   // hybrid: the FFN as ONE plain-fp16 launch (kernels_ffn16.hip) on the fast mode's chunk image — unscaled fp16 weights,
                            // K axes in MFMA k-slot order, stage i = W1 tile i | W2 slab i - 1 (ldm_pack::pack_ffn_image_pipelined)
        auto id = [](int x) { return x; };
        auto kslot = [](int k) { return ldm_pack::kslot(k); };
        const int Fq = round_up(F, 64);
        __half *w1p = nullptr, *w2p = nullptr;
        if ((rc = pack_w16(h, w.w1, F, D, round_up(F, 256), 512, id, kslot, &w1p))) return rc;
        if ((rc = pack_w16(h, w.w2, D, F, round_up(D, 256), Fq, id, kslot, &w2p))) return rc;
        const std::vector<uint16_t> h1p = download16(h, w1p, (size_t)F * 512, &rc);
        if (rc) return rc;
        const std::vector<uint16_t> h2 = download16(h, w2p, (size_t)round_up(D, 256) * Fq, &rc);
        if (rc) return rc;
        const std::vector<uint16_t> ffn = ldm_pack::pack_ffn_image(h1p.data(), h2.data(), Fq, F, 480);
        if ((rc = upload_image(h, ldm_pack::pack_ffn_image_pipelined(ffn, F / 32), &w.ffn16_img))) return rc;
      }
    }
  }
  if (h->cfg.precision == LDM_PREC_FAST_F16) {
    if ((rc = build_fast_weights(h))) return rc;
    if ((rc = build_loop_tables(h))) return rc;
  } else if (f16) {
    if ((rc = make_w16(h, h->head_w, C, D, h->Dp, &h->head_w16, &h->head_w16lo, &h->head_s))) return rc;
    if (h->lngemm && (rc = make_x3_image(h, h->head_w16, h->head_w16lo, h->x3_head_tiles, &h->x3_head))) return rc;
  }
  // schedule buffers are taken from the checkpoint, not recomputed
  static const char* names[kNumSched] = {"log_at",         "log_bt",         "log_ct",       "log_cumprod_at",
                                         "log_cumprod_bt", "log_cumprod_ct", "log_1_min_ct", "log_1_min_cumprod_ct"};
  static const char* keys[5] = {"c", "x", "y", "w", "h"};
  std::vector<float> host((size_t)kNumSched * h->cfg.n_attr * (T + 1), 0.f);
  for (int k = 0; k < kNumSched; ++k) {
    const bool cum = (k == kLogCumAt || k == kLogCumBt || k == kLogCumCt || k == kLog1mCumCt);
    for (int a = 0; a < h->cfg.n_attr; ++a) {
      // vanilla.py:66-73 registers ONE un-prefixed set; it is replicated into every attribute's row
      const std::string key = h->cfg.q_type == LDM_Q_VANILLA ? std::string(names[k]) : std::string(keys[a]) + "_" + names[k];
      const float* d = nullptr;
      if ((rc = need(h, key, {cum ? T + 1 : T}, &d))) return rc;
      HIP_OK(h, hipMemcpy(&host[((size_t)k * h->cfg.n_attr + a) * (T + 1)], d, (cum ? T + 1 : T) * sizeof(float),
                          hipMemcpyDeviceToHost));
    }
  }
  HIP_OK(h, hipMemcpy(h->sched, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_OK(h, hipDeviceSynchronize());
  HIP_OK(h, hipGetLastError());
  // graphs captured against older weights stay valid (pointers unchanged) but drop them anyway
  for (auto& g : h->graphs) g.destroy();
  h->graphs.clear();
  h->finalized = true;
  return 0;
}


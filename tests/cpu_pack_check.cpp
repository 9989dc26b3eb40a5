// CPU unit test of the weight repacking (layout_dm_amd/csrc/ldm_pack.h) against the read formulas of the fused
// kernels (csrc/ldm_pipes.h: the pipelines of kernels_stack.hip).  Compiled with plain g++ by
// tests/test_pack_images.py.  Weights are filled with unique 16-bit ids so every fetched element can be identified.
//
// What the kernels do (restated here):
//   * a tile / chunk image is copied linearly into an LDS stage (global_load_lds, lane-linear);
//   * MFMA A-operand read of k16-step ks by lane (r = lane&31, hi = lane>>5) from a 1-KiB-row tile:
//       byte  r*1024 + 256*(ks>>3) + ((((ks&7)<<1 | hi) ^ (r&15)) << 4)        -> 8 halfs  W[row r][16*ks + 8*hi + e]
//   * FFN GEMM2 A-operand read (slab at +32 KiB, output tile t, k-step sx):
//       byte  32768 + t*2048 + r*64 + (((2*sx+hi) ^ ((r>>2)&3)) << 4)          -> 8 halfs  W2[32t+r][chunk*32 + f(sx,hi,e)]
//     with f(s,g,e) = 16s + 8(e>>2) + 4g + (e&3): the accumulator layout of the GEMM1 MFMA that produced the B operand;
//   * out-projection B operand of k-step ks = 4h + 2dt + s: attention output d = 32dt + f(s,hi,e) - 32dt... of head h.
#include <cstdio>
#include <cstdlib>

#include "../layout_dm_amd/csrc/ldm_pack.h"

static int fails = 0;
#define CHECK(cond, ...)                  \
  do {                                    \
    if (!(cond)) {                        \
      if (fails < 10) { printf(__VA_ARGS__); printf("\n"); } \
      ++fails;                            \
    }                                     \
  } while (0)

static inline int f_slot(int s, int g, int e) { return 16 * s + 8 * (e >> 2) + 4 * g + (e & 3); }

int main() {
  const int D = 464, F = 1856, H = 8, dh = 58, HD = 512, Dq = 512, Fq = 1856;
  // ---------------- logical weights with identifiable values (never 0: 0 marks padding)
  auto id16 = [](int a, int b) { return (uint16_t)(1 + ((a * 2654435761u + b * 40503u) % 65535u)); };
  std::vector<uint16_t> w_in((size_t)3 * D * D), w_out((size_t)D * D), w1((size_t)F * D), w2((size_t)D * F);
  for (int n = 0; n < 3 * D; ++n) for (int k = 0; k < D; ++k) w_in[(size_t)n * D + k] = id16(n, k);
  for (int n = 0; n < D; ++n) for (int k = 0; k < D; ++k) w_out[(size_t)n * D + k] = id16(n + 5000, k);
  for (int n = 0; n < F; ++n) for (int k = 0; k < D; ++k) w1[(size_t)n * D + k] = id16(n + 9000, k);
  for (int n = 0; n < D; ++n) for (int k = 0; k < F; ++k) w2[(size_t)n * F + k] = id16(n + 20000, k);
  // ---------------- pack_w16 equivalents (dst[rmap(n)][cmap(k)] = src[n][k], zero padded)
  std::vector<uint16_t> p_in((size_t)1536 * Dq, 0), p_out_ks((size_t)512 * HD, 0), p_w1((size_t)2048 * Dq, 0),
      p_w2p((size_t)512 * Fq, 0);
  for (int n = 0; n < 3 * D; ++n) for (int k = 0; k < D; ++k)
    p_in[(size_t)ldm_pack::qkv_row(n, D, H, dh) * Dq + k] = w_in[(size_t)n * D + k];
  for (int n = 0; n < D; ++n) for (int k = 0; k < D; ++k)
    p_out_ks[(size_t)n * HD + ldm_pack::kslot(ldm_pack::head_col(k, dh))] = w_out[(size_t)n * D + k];
  for (int n = 0; n < F; ++n) for (int k = 0; k < D; ++k) p_w1[(size_t)n * Dq + k] = w1[(size_t)n * D + k];
  for (int n = 0; n < D; ++n) for (int k = 0; k < F; ++k) p_w2p[(size_t)n * Fq + ldm_pack::kslot(k)] = w2[(size_t)n * F + k];
  const std::vector<uint16_t> ffn = ldm_pack::pack_ffn_image(p_w1.data(), p_w2p.data(), Fq, F, 480);
  const std::vector<uint16_t> att = ldm_pack::pack_attn_image(p_in.data(), p_out_ks.data(), H, 15);
  CHECK(ffn.size() == (size_t)58 * 32768, "ffn image size");
  CHECK(att.size() == (size_t)64 * 16384, "attention image size");

  auto tile_read = [](const uint16_t* stage, int r, int hi, int ks) {  // 1-KiB-row tile, returns pointer to 8 halfs
    const int byte = r * 1024 + 256 * (ks >> 3) + (((((ks & 7) << 1) | hi) ^ (r & 15)) << 4);
    return stage + byte / 2;
  };
  // ---------------- fused FFN
  for (int c = 0; c < F / 32; ++c) {
    const uint16_t* stage = ffn.data() + (size_t)c * 32768;  // linear DMA: LDS stage == image chunk
    for (int r = 0; r < 32; ++r) for (int hi = 0; hi < 2; ++hi) {
      for (int ks = 0; ks < 29; ++ks) {  // GEMM1: hidden unit c*32+r, K slice 16ks+8hi..
        const uint16_t* p = tile_read(stage, r, hi, ks);
        for (int e = 0; e < 8; ++e)
          CHECK(p[e] == w1[(size_t)(c * 32 + r) * D + ks * 16 + hi * 8 + e], "W1 c=%d r=%d hi=%d ks=%d e=%d", c, r, hi, ks, e);
      }
      for (int t = 0; t < 15; ++t) for (int sx = 0; sx < 2; ++sx) {  // GEMM2: output feature 32t+r, hidden in k-slot order
        const int byte = 32768 + t * 2048 + r * 64 + (((2 * sx + hi) ^ ((r >> 2) & 3)) << 4);
        const uint16_t* p = stage + byte / 2;
        const int n = t * 32 + r;
        for (int e = 0; e < 8; ++e) {
          const uint16_t want = n < D ? w2[(size_t)n * F + c * 32 + f_slot(sx, hi, e)] : 0;
          CHECK(p[e] == want, "W2 c=%d t=%d r=%d hi=%d sx=%d e=%d", c, t, r, hi, sx, e);
        }
      }
    }
  }
  // ---------------- fused FFN version 2: W1 with the K axis in k-slot order.  The prologue loads the token row in
  // ACCUMULATOR layout — lane (row, hi) holds columns 8g + 4hi + i of 8-column group g — and uses groups 2ks, 2ks+1
  // as the B-operand fragment of k16-step ks: fragment element e <-> column 16ks + 8(e>>2) + 4hi + (e&3).  The A
  // operand fetched for (ks, hi, e) must therefore be W1[hidden][that column].
  {
    std::vector<uint16_t> p_w1ks((size_t)2048 * Dq, 0);
    for (int n = 0; n < F; ++n) for (int k = 0; k < D; ++k) p_w1ks[(size_t)n * Dq + ldm_pack::kslot(k)] = w1[(size_t)n * D + k];
    const std::vector<uint16_t> ffn_ks = ldm_pack::pack_ffn_image(p_w1ks.data(), p_w2p.data(), Fq, F, 480);
    {
      // the re-timed image of the software-pipelined chunk stream (FfnStream PIPE): stage i = W1 tile of chunk i (what
      // iteration i's GEMM1 reads: the first 32 KiB of the stage) | W2 slab of chunk i - 1 (what its GEMM2 reads: the
      // second 32 KiB); stage 0 has a zero W2 half, stage nc a zero W1 half
      const int nc = F / 32;
      const std::vector<uint16_t> pipe = ldm_pack::pack_ffn_image_pipelined(ffn_ks, nc);
      CHECK(pipe.size() == (size_t)(nc + 1) * 32768, "pipelined ffn image size");
      for (int i = 0; i <= nc; ++i)
        for (int e = 0; e < 16384; ++e) {
          const uint16_t w1_want = i < nc ? ffn_ks[(size_t)i * 32768 + e] : 0;
          const uint16_t w2_want = i > 0 ? ffn_ks[(size_t)(i - 1) * 32768 + 16384 + e] : 0;
          CHECK(pipe[(size_t)i * 32768 + e] == w1_want, "pipelined W1 half: stage %d half-word %d", i, e);
          CHECK(pipe[(size_t)i * 32768 + 16384 + e] == w2_want, "pipelined W2 half: stage %d half-word %d", i, e);
        }
    }
    for (int c = 0; c < F / 32; c += 7) {
      const uint16_t* stage = ffn_ks.data() + (size_t)c * 32768;
      for (int r = 0; r < 32; ++r) for (int hi = 0; hi < 2; ++hi) for (int ks = 0; ks < 29; ++ks) {
        const uint16_t* p = tile_read(stage, r, hi, ks);
        for (int e = 0; e < 8; ++e) {
          const int g = 2 * ks + (e >> 2), col = 8 * g + 4 * hi + (e & 3);  // accumulator-layout column of (group, hi, i)
          CHECK(col == 16 * ks + f_slot(0, hi, e), "column identity");
          CHECK(p[e] == w1[(size_t)(c * 32 + r) * D + col], "W1(k-slot) c=%d r=%d hi=%d ks=%d e=%d", c, r, hi, ks, e);
        }
      }
    }
  }
  // ---------------- attention block
  for (int h = 0; h < H; ++h) for (int j = 0; j < 6; ++j) {
    const uint16_t* stage = att.data() + (size_t)(h * 6 + j) * 16384;
    const int which = (j < 2) ? 1 : (j < 4 ? 2 : 0);  // k0 k1 v0 v1 q0 q1
    for (int r = 0; r < 32; ++r) for (int hi = 0; hi < 2; ++hi) for (int ks = 0; ks < 29; ++ks) {
      const uint16_t* p = tile_read(stage, r, hi, ks);
      const int d = (j & 1) * 32 + r;
      for (int e = 0; e < 8; ++e) {
        const uint16_t want = d < dh ? w_in[(size_t)(which * D + h * dh + d) * D + ks * 16 + hi * 8 + e] : 0;
        CHECK(p[e] == want, "Win h=%d j=%d r=%d hi=%d ks=%d e=%d", h, j, r, hi, ks, e);
      }
    }
  }
  for (int ot = 0; ot < 15; ++ot) {
    const uint16_t* stage = att.data() + (size_t)(H * 6 + ot) * 16384;
    for (int r = 0; r < 32; ++r) for (int hi = 0; hi < 2; ++hi) for (int ks = 0; ks < 32; ++ks) {
      const uint16_t* p = tile_read(stage, r, hi, ks);
      const int n = ot * 32 + r, h = ks >> 2, dt = (ks >> 1) & 1, s = ks & 1;
      for (int e = 0; e < 8; ++e) {
        const int d = dt * 32 + f_slot(s, hi, e);  // attention-output feature this k-slot carries (B operand layout)
        const uint16_t want = (n < D && d < dh) ? w_out[(size_t)n * D + h * dh + d] : 0;
        CHECK(p[e] == want, "Wout ot=%d r=%d hi=%d ks=%d e=%d", ot, r, hi, ks, e);
      }
    }
  }
  for (size_t i = (size_t)63 * 16384; i < att.size(); ++i) CHECK(att[i] == 0, "padding tile not zero");
  // ---------------- fused layer kernel: in_proj tiles + out-projection K-slabs
  {
    const std::vector<uint16_t> sl = ldm_pack::pack_attn_slab_image(p_in.data(), p_out_ks.data(), H);
    CHECK(sl.size() == (size_t)(48 + 16 + 1) * 16384, "slab image size");
    for (size_t i = 0; i < (size_t)48 * 16384; ++i) CHECK(sl[i] == att[i], "in_proj tiles differ from pack_attn_image");
    for (int c = 0; c < 16; ++c) {  // k chunk c: head c/2, d-half c%2; k16-steps ks = 2c + sx
      const uint16_t* stage = sl.data() + (size_t)(48 + c) * 16384;
      const int h = c >> 1, dt = c & 1;
      for (int t = 0; t < 15; ++t) for (int r = 0; r < 32; ++r) for (int hi = 0; hi < 2; ++hi) for (int sx = 0; sx < 2; ++sx) {
        const int byte = t * 2048 + r * 64 + (((2 * sx + hi) ^ ((r >> 2) & 3)) << 4);  // SlabPair slab read (ldm_pipes.h)
        const uint16_t* p = stage + byte / 2;
        const int n = t * 32 + r;
        for (int e = 0; e < 8; ++e) {
          const int d = dt * 32 + f_slot(sx, hi, e);
          const uint16_t want = (n < D && d < dh) ? w_out[(size_t)n * D + h * dh + d] : 0;
          CHECK(p[e] == want, "Wout slab c=%d t=%d r=%d hi=%d sx=%d e=%d", c, t, r, hi, sx, e);
        }
      }
    }
    for (size_t i = (size_t)64 * 16384; i < sl.size(); ++i) CHECK(sl[i] == 0, "slab image padding stage not zero");
    // multi-layer kernel: in_proj with the K axis in k-slot order (fragments come from accumulator-layout registers:
    // element e of k16-step ks <-> input feature 16ks + f_slot(0, hi, e))
    std::vector<uint16_t> p_in_ks((size_t)1536 * Dq, 0);
    for (int n = 0; n < 3 * D; ++n) for (int k = 0; k < D; ++k)
      p_in_ks[(size_t)ldm_pack::qkv_row(n, D, H, dh) * Dq + ldm_pack::kslot(k)] = w_in[(size_t)n * D + k];
    const std::vector<uint16_t> slk = ldm_pack::pack_attn_slab_image(p_in_ks.data(), p_out_ks.data(), H);
    for (int h = 0; h < H; ++h) for (int j = 0; j < 6; ++j) {
      const uint16_t* stage = slk.data() + (size_t)(h * 6 + j) * 16384;
      const int which = (j < 2) ? 1 : (j < 4 ? 2 : 0);
      for (int r = 0; r < 32; ++r) for (int hi = 0; hi < 2; ++hi) for (int ks = 0; ks < 29; ++ks) {
        const uint16_t* p = tile_read(stage, r, hi, ks);
        const int d = (j & 1) * 32 + r;
        for (int e = 0; e < 8; ++e) {
          const uint16_t want = d < dh ? w_in[(size_t)(which * D + h * dh + d) * D + 16 * ks + f_slot(0, hi, e)] : 0;
          CHECK(p[e] == want, "Win(k-slot) h=%d j=%d r=%d hi=%d ks=%d e=%d", h, j, r, hi, ks, e);
        }
      }
    }
    for (size_t i = (size_t)48 * 16384; i < slk.size(); ++i) CHECK(slk[i] == sl[i], "k-slot image: out-proj slabs must not change");
    // stack kernel: per head 6 in_proj tiles (k-slot K) + its two out-proj slabs, then two zero stages
    const std::vector<uint16_t> hd = ldm_pack::pack_attn_head_image(slk, H);
    CHECK(hd.size() == (size_t)(H * 8 + 2) * 16384, "head image size");
    for (int hh = 0; hh < H; ++hh) {
      for (int j6 = 0; j6 < 6; ++j6)
        CHECK(memcmp(hd.data() + (size_t)(hh * 8 + j6) * 16384, slk.data() + (size_t)(hh * 6 + j6) * 16384, 32768) == 0,
              "head image: tile %d of head %d", j6, hh);
      for (int d2 = 0; d2 < 2; ++d2)
        CHECK(memcmp(hd.data() + (size_t)(hh * 8 + 6 + d2) * 16384, slk.data() + (size_t)(48 + 2 * hh + d2) * 16384, 32768) == 0,
              "head image: slab %d of head %d", d2, hh);
    }
    for (size_t i = (size_t)H * 8 * 16384; i < hd.size(); ++i) CHECK(hd[i] == 0, "head image padding stages not zero");
    // stack kernel's vocabulary head: 32-class tiles with the 1-KiB-row tile swizzle, K axis in k-slot order
    {
      const int C = 154, Cp = 160;
      std::vector<uint16_t> hw((size_t)C * D), hp((size_t)256 * Dq, 0);
      for (int n = 0; n < C; ++n) for (int k = 0; k < D; ++k) hw[(size_t)n * D + k] = id16(n + 40000, k);
      for (int n = 0; n < C; ++n) for (int k = 0; k < D; ++k) hp[(size_t)n * Dq + ldm_pack::kslot(k)] = hw[(size_t)n * D + k];
      const std::vector<uint16_t> hi_img = ldm_pack::pack_head_image(hp.data(), Cp / 32);
      CHECK(hi_img.size() == (size_t)(Cp / 32) * 16384, "head tile image size");
      for (int t = 0; t < Cp / 32; ++t)
        for (int r = 0; r < 32; ++r) for (int hi = 0; hi < 2; ++hi) for (int ks = 0; ks < 29; ++ks) {
          const uint16_t* p = tile_read(hi_img.data() + (size_t)t * 16384, r, hi, ks);
          for (int e = 0; e < 8; ++e) {
            // fragment element e of k16-step ks, lane half hi <-> accumulator-layout column 16 ks + 8 (e >> 2) + 4 hi + (e & 3)
            const int k = 16 * ks + 8 * (e >> 2) + 4 * hi + (e & 3);
            const int n = t * 32 + r;
            const uint16_t want = (n < C && k < D) ? hw[(size_t)n * D + k] : 0;
            CHECK(p[e] == want, "head tile t=%d r=%d hi=%d ks=%d e=%d", t, r, hi, ks, e);
          }
        }
    }
  }
  {
    // row-resident x3 GEMM (kernels_lngemm.hip): stage t = hi tile | lo tile of output columns 32 t .. 32 t + 31, each read
    // with the tile formula (lg_read: hi at +0, lo at +32 KiB of the stage), K axis in k-slot order
    const int N = 1392, n_tiles = 44;
    std::vector<uint16_t> hi((size_t)n_tiles * 32 * Dq, 0), lo((size_t)n_tiles * 32 * Dq, 0);
    for (int n = 0; n < N; ++n) for (int k = 0; k < D; ++k) {
      hi[(size_t)n * Dq + ldm_pack::kslot(k)] = id16(n + 100, k);
      lo[(size_t)n * Dq + ldm_pack::kslot(k)] = id16(n + 30000, k);
    }
    const std::vector<uint16_t> img = ldm_pack::pack_x3_tile_image(hi.data(), lo.data(), n_tiles);
    CHECK(img.size() == (size_t)n_tiles * 32768, "x3 tile image size");
    for (int t = 0; t < n_tiles; ++t)
      for (int part = 0; part < 2; ++part)
        for (int r = 0; r < 32; ++r) for (int hh = 0; hh < 2; ++hh) for (int ks = 0; ks < 29; ++ks) {
          const uint16_t* p = tile_read(img.data() + (size_t)t * 32768 + part * 16384, r, hh, ks);
          for (int e = 0; e < 8; ++e) {
            const int k = 16 * ks + 8 * (e >> 2) + 4 * hh + (e & 3);
            const int n = t * 32 + r;
            const uint16_t want = (n < N && k < D) ? id16(n + (part ? 30000 : 100), k) : 0;
            CHECK(p[e] == want, "x3 tile t=%d part=%d r=%d hi=%d ks=%d e=%d", t, part, r, hh, ks, e);
          }
        }
  }
  {  // K-slab image of the row-resident kernel's GEMM prologue (kernels_lngemm.hip lp_read: lane (column j, k half hh) of tile t reads
     // 16 bytes at stage + t * 2 KiB + j * 64 + (((2 s + hh) ^ ((j >> 2) & 3)) << 4) for k16-step s of the stage; lo slab: + 32 KiB)
    const int N = 464, ld = 1856, K = 1856;
    std::vector<uint16_t> hi((size_t)N * ld), lo((size_t)N * ld);
    for (int n = 0; n < N; ++n)
      for (int k = 0; k < ld; ++k) {
        hi[(size_t)n * ld + k] = id16(n + 7, k);
        lo[(size_t)n * ld + k] = id16(n + 20000, k);
      }
    const std::vector<uint16_t> img = ldm_pack::pack_x3_slab_image(hi.data(), lo.data(), N, ld, K);
    CHECK(img.size() == (size_t)ldm_pack::x3_slab_stages(K) * 32768 && ldm_pack::x3_slab_stages(K) == 60 && ldm_pack::x3_slab_stages(512) == 18, "x3 slab image size");
    for (size_t i = (size_t)(K / 32) * 32768; i < img.size(); ++i) CHECK(img[i] == 0, "x3 slab padding stage not zero at %zu", i);
    for (int st = 0; st < K / 32; ++st)
      for (int part = 0; part < 2; ++part)
        for (int t = 0; t < 15; ++t) for (int j = 0; j < 32; ++j) for (int hh = 0; hh < 2; ++hh) for (int sx = 0; sx < 2; ++sx) {
          const size_t byte = (size_t)st * 65536 + part * 32768 + t * 2048 + j * 64 + ((((sx << 1) | hh) ^ ((j >> 2) & 3)) << 4);
          const uint16_t* p = img.data() + byte / 2;
          for (int e = 0; e < 8; ++e) {
            const int n = t * 32 + j, k = st * 32 + sx * 16 + hh * 8 + e;   // natural MFMA k order: lane half hh holds k = 8 hh .. 8 hh + 7
            const uint16_t want = n < N ? id16(n + (part ? 20000 : 7), k) : 0;
            CHECK(p[e] == want, "x3 slab st=%d part=%d t=%d j=%d hh=%d s=%d e=%d", st, part, t, j, hh, sx, e);
          }
        }
  }
  {  // k-step image of out_proj for the fused attention + out_proj kernel (kernels_attnout.hip: lane (column m, k half g) of tile t reads 16
     // bytes at stage + t * 1 KiB + m * 32 + ((g ^ ((m >> 3) & 1)) << 4), lo half + 16 KiB; stage = head * 4 + s; element e = the attention
     // output d = f_slot(s & 1, g, e) of d tile s >> 1, i.e. accumulator register 8 (s & 1) + e of the lane's O^T tile)
    const int N = 464, ld = 512;
    std::vector<uint16_t> hi((size_t)N * ld), lo((size_t)N * ld);
    for (int n = 0; n < N; ++n)
      for (int k = 0; k < ld; ++k) {
        hi[(size_t)n * ld + k] = id16(n + 11, k);
        lo[(size_t)n * ld + k] = id16(n + 21000, k);
      }
    const std::vector<uint16_t> img = ldm_pack::pack_x3_kstep_image(hi.data(), lo.data(), N, ld, H, dh);
    CHECK(img.size() == (size_t)H * 4 * 16384, "x3 k-step image size");
    for (int hh = 0; hh < H; ++hh) for (int s = 0; s < 4; ++s) for (int part = 0; part < 2; ++part)
      for (int t = 0; t < 15; ++t) for (int m = 0; m < 32; ++m) for (int g = 0; g < 2; ++g) {
        const size_t byte = (size_t)(hh * 4 + s) * 32768 + part * 16384 + t * 1024 + m * 32 + ((g ^ ((m >> 3) & 1)) << 4);
        const uint16_t* p = img.data() + byte / 2;
        for (int e = 0; e < 8; ++e) {
          const int n = t * 32 + m, d = 32 * (s >> 1) + f_slot(s & 1, g, e);
          const uint16_t want = (n < N && d < dh) ? id16(n + (part ? 21000 : 11), hh * dh + d) : 0;
          CHECK(p[e] == want, "x3 k-step head=%d s=%d part=%d t=%d m=%d g=%d e=%d", hh, s, part, t, m, g, e);
        }
      }
    // the 1 KiB behind each 15-KiB half (rows 480 .. 511) is never read and stays zero
    for (int st = 0; st < H * 4; ++st) for (int part = 0; part < 2; ++part)
      for (int i = 480 * 16; i < 8192; ++i) CHECK(img[(size_t)st * 16384 + part * 8192 + i] == 0, "x3 k-step padding not zero");
  }
  if (fails) { printf("FAILED: %d mismatches\n", fails); return 1; }
  printf("OK: FFN image (58 chunks) and attention image (63 tiles + pad) match the kernels' read formulas\n");
  return 0;
}

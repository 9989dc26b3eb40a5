// Host-side weight repacking for the fused fast-mode kernels — pure C++ (no HIP), so the index maps and the
// LDS-image packers can be unit-tested on a CPU (tests/cpu_pack_check.cpp) against the kernels' read formulas.
//
// Index maps (destination position <- logical weight index):
//   qkv_row    in_proj row  n = which*D + head*dh + d  ->  (which*H + head)*64 + d      (head slices padded to 64)
//   head_col   out_proj col k = head*dh + d            ->  head*64 + d
//   kslot      MFMA k-slot order inside every 32-wide K chunk: position 16s + 8g + e holds logical 16s + 8(e>>2) + 4g + (e&3),
//              i.e. the accumulator layout of v_mfma_f32_32x32x16_f16 (lane half g, register 8s + e) IS the B operand
//              of the next MFMA (FFN GEMM2 and the out-projection slabs of the stack kernel)
// LDS images: global memory holds exactly what the kernels want in LDS (tiles in consumption order, bank swizzle
// applied), so the weight stream is a linear copy.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace ldm_pack {

inline int qkv_row(int n, int D, int H, int dh) { return ((n / D) * H + (n % D) / dh) * 64 + (n % D) % dh; }
inline int head_col(int k, int dh) { return (k / dh) * 64 + k % dh; }
inline int kslot(int k) {
  const int c = k & ~31, f = k & 31;
  const int s2 = f >> 4, e_hi = (f >> 3) & 1, g = (f >> 2) & 1, e_lo = f & 3;
  return c + 16 * s2 + 8 * g + 4 * e_hi + e_lo;
}

// one 32-row x 512-half tile: row i = src row (or zeros), 16-B chunk L -> physical chunk L ^ (i & 15)
inline void put_tile_1k(uint16_t* dst, const uint16_t* const src_rows[32]) {
  for (int i = 0; i < 32; ++i) {
    if (!src_rows[i]) continue;
    for (int L = 0; L < 64; ++L) memcpy(dst + i * 512 + ((L ^ (i & 15)) << 3), src_rows[i] + L * 8, 16);
  }
}

// w1: [>=F rows][512] fp16 (row = hidden unit), w2p: [>=n_rows2 rows][ldw2] fp16 with the k-slot ordered K axis.
// Per 32-wide hidden chunk: 64 KiB = 32 KiB W1 tile | 30 KiB W2 slab (480 rows x 64 B, chunk L of row n at
// L ^ ((n >> 2) & 3)) | 2 KiB padding.
inline std::vector<uint16_t> pack_ffn_image(const uint16_t* w1, const uint16_t* w2p, int ldw2, int F, int n_rows2) {
  const int nc = F / 32;
  std::vector<uint16_t> img((size_t)nc * 32768, 0);
  for (int c = 0; c < nc; ++c) {
    uint16_t* t = img.data() + (size_t)c * 32768;
    const uint16_t* rows[32];
    for (int i = 0; i < 32; ++i) rows[i] = w1 + (size_t)(c * 32 + i) * 512;
    put_tile_1k(t, rows);
    uint16_t* t2 = t + 16384;
    for (int n = 0; n < n_rows2; ++n)
      for (int L = 0; L < 4; ++L)
        memcpy(t2 + n * 32 + ((L ^ ((n >> 2) & 3)) << 3), w2p + (size_t)n * ldw2 + c * 32 + L * 8, 16);
  }
  return img;
}

// The same weights RE-TIMED for the software-pipelined chunk stream (FfnStream<..., PIPE = true>): iteration i of the
// stream runs GEMM1 of chunk i and GEMM2 of chunk i - 1 — the ReLU / cast of chunk i sits in the MFMA shadows of that
// GEMM2 instead of idling the matrix pipe between the two GEMMs of one chunk — so stage i holds W1 tile i | W2 slab i - 1.
// nc + 1 stages: stage 0 has a zero W2 half (GEMM2 of iteration 0 multiplies zero fragments), stage nc a zero W1 half
// (its GEMM1 result is never used).
inline std::vector<uint16_t> pack_ffn_image_pipelined(const std::vector<uint16_t>& img, int nc) {
  std::vector<uint16_t> out((size_t)(nc + 1) * 32768, 0);
  for (int i = 0; i <= nc; ++i) {
    if (i < nc) memcpy(out.data() + (size_t)i * 32768, img.data() + (size_t)i * 32768, 16384 * 2);
    if (i > 0) memcpy(out.data() + (size_t)i * 32768 + 16384, img.data() + (size_t)(i - 1) * 32768 + 16384, 16384 * 2);
  }
  return out;
}

// w_in: [3*H*64][512] head-padded in_proj (q | k | v), w_out_ks: [>=32*n_out_tiles][512] out_proj with head-padded
// k-slot K.  Tiles: h*6 + {k0 k1 v0 v1 q0 q1}, then the out_proj tiles, then one zero tile.
inline std::vector<uint16_t> pack_attn_image(const uint16_t* w_in, const uint16_t* w_out_ks, int H, int n_out_tiles) {
  const int nt = H * 6 + n_out_tiles + 1;
  std::vector<uint16_t> img((size_t)nt * 16384, 0);
  const uint16_t* rows[32];
  for (int hh = 0; hh < H; ++hh)
    for (int j = 0; j < 6; ++j) {
      const int which = (j < 2) ? 1 : (j < 4 ? 2 : 0);
      const int row0 = (which * H + hh) * 64 + (j & 1) * 32;
      for (int i = 0; i < 32; ++i) rows[i] = w_in + (size_t)(row0 + i) * 512;
      put_tile_1k(img.data() + (size_t)(hh * 6 + j) * 16384, rows);
    }
  for (int t = 0; t < n_out_tiles; ++t) {
    for (int i = 0; i < 32; ++i) rows[i] = w_out_ks + (size_t)(t * 32 + i) * 512;
    put_tile_1k(img.data() + (size_t)(H * 6 + t) * 16384, rows);
  }
  return img;
}

// Attention image in slab form (the building block of pack_attn_head_image): the 6H in_proj tiles as in pack_attn_image, then the
// out-projection as 16 K-SLABS (one per 32-wide k chunk c = (head, d-half)): 480 output rows x 64 B, 16-B chunk L of
// row n at physical chunk L ^ ((n >> 2) & 3) — the W2-slab format of the fused FFN, so the out-projection runs as
// 30 independent-accumulator MFMAs per slab over 15 persistent output tiles — then one zero stage (last prefetch).
inline std::vector<uint16_t> pack_attn_slab_image(const uint16_t* w_in, const uint16_t* w_out_ks, int H) {
  const int n_slab = H * 2;
  const int nt = H * 6 + n_slab + 1;
  std::vector<uint16_t> img((size_t)nt * 16384, 0);
  const uint16_t* rows[32];
  for (int hh = 0; hh < H; ++hh)
    for (int j = 0; j < 6; ++j) {
      const int which = (j < 2) ? 1 : (j < 4 ? 2 : 0);
      const int row0 = (which * H + hh) * 64 + (j & 1) * 32;
      for (int i = 0; i < 32; ++i) rows[i] = w_in + (size_t)(row0 + i) * 512;
      put_tile_1k(img.data() + (size_t)(hh * 6 + j) * 16384, rows);
    }
  for (int c = 0; c < n_slab; ++c) {
    uint16_t* t2 = img.data() + (size_t)(H * 6 + c) * 16384;
    for (int n = 0; n < 480; ++n)
      for (int L = 0; L < 4; ++L)
        memcpy(t2 + n * 32 + ((L ^ ((n >> 2) & 3)) << 3), w_out_ks + (size_t)n * 512 + c * 32 + L * 8, 16);
  }
  return img;
}

// Attention image of the stack kernel (kernels_stack.hip): per head EIGHT consecutive 32-KiB stages — the six in_proj
// tiles k0 k1 v0 v1 q0 q1 (K axis in k-slot order: the kernel builds its activation fragments from accumulator-layout
// registers) followed by the head's two out-projection K-slabs (k chunks 2h, 2h + 1 of pack_attn_slab_image) — so that
// the out-projection of a head runs right behind its attention core; then two zero stages (the last head's "next head"
// prefetch needs no branch).  Built from a pack_attn_slab_image image.
inline std::vector<uint16_t> pack_attn_head_image(const std::vector<uint16_t>& slab_img, int H) {
  const size_t st = 16384;  // halfs per 32-KiB stage
  std::vector<uint16_t> img((size_t)(H * 8 + 2) * st, 0);
  for (int hh = 0; hh < H; ++hh) {
    for (int j = 0; j < 6; ++j)
      memcpy(img.data() + (size_t)(hh * 8 + j) * st, slab_img.data() + (size_t)(hh * 6 + j) * st, st * 2);
    for (int d = 0; d < 2; ++d)
      memcpy(img.data() + (size_t)(hh * 8 + 6 + d) * st, slab_img.data() + (size_t)(H * 6 + 2 * hh + d) * st, st * 2);
  }
  return img;
}

// Vocabulary-head image of the stack kernel: n_tiles x 32 KiB, tile t = classes 32t .. 32t + 31 (rows of w: [>= 32 n_tiles
// rows][512] fp16, K axis already in k-slot order, zero rows beyond the vocabulary), 1-KiB rows with the tile swizzle.
inline std::vector<uint16_t> pack_head_image(const uint16_t* w, int n_tiles) {
  std::vector<uint16_t> img((size_t)n_tiles * 16384, 0);
  const uint16_t* rows[32];
  for (int t = 0; t < n_tiles; ++t) {
    for (int i = 0; i < 32; ++i) rows[i] = w + (size_t)(t * 32 + i) * 512;
    put_tile_1k(img.data() + (size_t)t * 16384, rows);
  }
  return img;
}

// hi | lo tile images of the row-resident fp16 x 3 GEMM (kernels_lngemm.hip): per 32-row tile ONE 64-KiB stage = the hi tile
// (1-KiB rows, tile swizzle: pack_head_image's format) followed by the lo tile.  hi / lo: [>= 32 n_tiles rows][512] fp16 with
// the K axis already in k-slot order, zero rows beyond the GEMM's N.
inline std::vector<uint16_t> pack_x3_tile_image(const uint16_t* hi, const uint16_t* lo, int n_tiles) {
  std::vector<uint16_t> img((size_t)n_tiles * 32768, 0);
  const uint16_t* rows[32];
  for (int t = 0; t < n_tiles; ++t) {
    for (int i = 0; i < 32; ++i) rows[i] = hi + (size_t)(t * 32 + i) * 512;
    put_tile_1k(img.data() + (size_t)t * 32768, rows);
    for (int i = 0; i < 32; ++i) rows[i] = lo + (size_t)(t * 32 + i) * 512;
    put_tile_1k(img.data() + (size_t)t * 32768 + 16384, rows);
  }
  return img;
}

// K-slab image of a split-mode weight for the GEMM PROLOGUE of the row-resident kernel (kernels_lngemm.hip, PRE = true: out_proj /
// linear2 run in front of the LayerNorm that consumes their sum).  hi / lo: [N rows][ld] fp16, natural K order, K % 32 == 0.
// One 64-KiB stage per 32 k (two k16-steps): hi slab at byte 0, lo slab at byte 32 768; a slab is 480 rows (output columns,
// zero beyond N) x 64 B, 16-byte chunk L (k = 32 stage + 8 L ..) of row n at physical chunk L ^ ((n >> 2) & 3) — the W2-slab
// format of the fused FFN (pack_ffn_image), conflict-free for ds_read_b128 by (column, k half) lanes.
inline int x3_slab_stages(int K) { return (K / 32 + 2) / 3 * 3; }   // the kernel walks the slabs three at a time: zero slabs fill up
inline std::vector<uint16_t> pack_x3_slab_image(const uint16_t* hi, const uint16_t* lo, int N, int ld, int K) {
  const int n_stage = K / 32;
  std::vector<uint16_t> img((size_t)x3_slab_stages(K) * 32768, 0);
  for (int st = 0; st < n_stage; ++st)
    for (int part = 0; part < 2; ++part) {
      const uint16_t* src = part ? lo : hi;
      uint16_t* dst = img.data() + (size_t)st * 32768 + part * 16384;
      for (int n = 0; n < N && n < 480; ++n)
        for (int L = 0; L < 4; ++L) memcpy(dst + n * 32 + ((L ^ ((n >> 2) & 3)) << 3), src + (size_t)n * ld + st * 32 + L * 8, 16);
    }
  return img;
}

// k-step image of out_proj for the fused attention + out_proj kernel of the split mode (kernels_attnout.hip).  hi / lo: [N rows][ld]
// fp16, natural K order (k = head * dh + d).  One 32-KiB stage per (head, k16-step s of the head's 64 padded d): hi half at byte 0, lo
// half at byte 16 384; a half is 480 rows (output columns, zero beyond N) x 32 B = two 16-byte chunks, chunk g of row n at physical
// chunk g ^ ((n >> 3) & 1) (conflict-free ds_read_b128 by (column, k half) lanes).  Chunk g, element e holds d = 16 s + 8 (e >> 2) + 4 g +
// (e & 3) — MFMA k-slot order (kslot above): the attention output's accumulator registers 8 s' .. 8 s' + 7 of d tile s / 2 (s' = s & 1)
// ARE the B operand of the stage; zero for d >= dh.
inline std::vector<uint16_t> pack_x3_kstep_image(const uint16_t* hi, const uint16_t* lo, int N, int ld, int H, int dh) {
  std::vector<uint16_t> img((size_t)H * 4 * 16384, 0);
  for (int hh = 0; hh < H; ++hh)
    for (int s = 0; s < 4; ++s)
      for (int part = 0; part < 2; ++part) {
        const uint16_t* src = part ? lo : hi;
        uint16_t* dst = img.data() + (size_t)(hh * 4 + s) * 16384 + part * 8192;
        for (int n = 0; n < N && n < 480; ++n)
          for (int g = 0; g < 2; ++g)
            for (int e = 0; e < 8; ++e) {
              const int d = 16 * s + 8 * (e >> 2) + 4 * g + (e & 3);
              if (d < dh) dst[n * 16 + ((g ^ ((n >> 3) & 1)) << 3) + e] = src[(size_t)n * ld + hh * dh + d];
            }
      }
  return img;
}

}  // namespace ldm_pack

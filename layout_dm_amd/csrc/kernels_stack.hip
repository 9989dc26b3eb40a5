// The whole denoiser stack in ONE launch per step, one workgroup per LAYOUT, with the layout's rows RESIDENT in the
// out-projection accumulators (15 tiles = 240 AGPRs per wave) from the embedding output to the input of the head.
//
// A per-layer kernel (r02's layer_stream_k, removed in r03) spends a quarter of its cycles in three all-CU memory bursts per layer —
// operand rows in, residual rows in, result rows out (3 x 59 MB per 256-layout launch, matrix pipes idle) — because the
// attention heads park their output fragments in the AGPRs the out-projection accumulators need.  Here the
// out-projection of a head runs RIGHT BEHIND its attention core (SlabPair: the head's two K-slabs), so nothing is
// parked, the accumulators are alive through the whole layer and carry the residual stream:
//
//   prologue      one read of the rows in accumulator layout -> AdaLN fragments (k-slot K order) AND residual seed
//   per layer     per head: HeadStream (k0 k1 v0 v1 q0 q1) -> AttnCoreV (all-VGPR core) -> SlabPair (acc += o_h·Wo_h^T)
//                 LN2 + FFN chunk stream on the same accumulators (FfnStream)
//                 layer boundary, in registers: statistics of x2, next layer's AdaLN fragments, next residual seed
//   epilogue      statistics + one write of the rows (HEAD 0) | vocabulary head, logits out (HEAD 1)
//
// HEAD 2 — the whole REVERSE LOOP of a layout in its workgroup (BaseMaskAndReplaceDiffusion.sample, base.py:293-371):
// layouts never exchange data, so the workgroup that owns a layout runs all its T steps — tokens in LDS, the embedding
// gathered straight into the accumulators, the stack, the vocabulary head, and behind it the step's tail (log-softmax,
// constrained posterior, cond overrides, draw: ldm_post_token.h on 16-lane groups, four tokens per wavefront) on the
// layout's logits in LDS.  One launch per sampling call; per step nothing touches memory but the weight stream (L2 /
// Infinity Cache resident), the 125 cond tokens and, on request, the intermediate tokens.  No hipGraph, no chunk
// pipelines, no logits / row / embedding round trips.
//
// Between layers nothing touches memory but the weight stream.  Weight image per layer: ldm_pack::pack_attn_head_image
// (per head 6 in_proj tiles + its 2 out-proj slabs, 9-slot ring cycle: see SlabPair) and the k-slot FFN image.
// Reference semantics: TransformerEncoder / Block.forward, trainer/models/transformer_utils.py:165-246.
#include <cstdlib>

#include "ldm_kernels.h"
#include "ldm_dma.h"
#include "ldm_pipes.h"
#include "ldm_post_dpp.h"
#include "ldm_relation_core.h"

namespace ldm {

struct StackArgs {
  FusedLayerSet ls;     // per layer: head image, in_proj bias, AdaLN scale / shift, b_out + W_out b_v, FFN image, b1 b2 g2 be2
  float* x;             // [M, ldx] rows in (HEAD 1)
  int ldx, N, S, H, n_chunks;
  float scale_log2e;
  // fused vocabulary head (nn_lib.py:186-189: LayerNorm + Linear without bias), or head_img == nullptr
  const char* head_img; // n_head_tiles x 32 KiB LDS images of 32 classes each (K axis in k-slot order, zero rows beyond C)
  const float *head_g, *head_b;
  float* logits;        // [M, ldl]
  int ldl, n_head_tiles;
  // HEAD == 2: the reverse loop (see the header).  post carries the tail's parameters (schedule, cond, sampler, RNG,
  // vocabulary, emb / pos, tokens in = post.tokens, tokens out = post.tokens_out, tie flags of step i at tie_flags + i * tie_ld)
  PostArgs post;
  const float* adaln;        // [T][L][2 N] AdaLN (scale | shift) table (transformer_utils.py:79-81)
  StackTables tbl;           // the same parameters as LDS images (ldm_kernels.h): DMA'd one phase ahead of their use
  int32_t* inter;            // [n_steps][inter_ld][S] tokens after every step, or nullptr (get_intermediate_results)
  int n_steps, inter_ld, tie_ld;
  int16_t t_model[kStackLoopMaxSteps], t_post[kStackLoopMaxSteps];  // the denoiser's timestep and q_posterior's (base.py:218-240)
  // REL: cond=relation (base.py:261-269) — the graph of the launch's layouts (edge_off + workgroup index) and its n_bin
  RelGraph rel;
  int rel_n_bin;
};
// HEAD == 2 reads its loop parameters from the kernel-argument segment AT THE POINT OF USE, through a pointer hipcc
// cannot see through: hoisted out of the step loop they would occupy ~60 SGPRs for the whole kernel, which already
// fills the register file (measured: 224 SGPR spills and, through the VGPRs those need, 116 bytes of scratch per lane).
typedef const __attribute__((address_space(4))) StackArgs* stack_kargs_ptr;
__device__ __forceinline__ stack_kargs_ptr stack_kargs() {
  stack_kargs_ptr q = (stack_kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(q));
  return q;
}
constexpr int kPostLd = 161;   // floats per token row of the logits in LDS (odd: the 16-lane groups of a wavefront hit distinct banks)
constexpr int kPostRows = 128 * kPostLd * 4;  // bytes of the logits rows; behind them one float4 (max, lse, max |x|, -) per row
constexpr int kStackLoopB1 = 2048;            // HEAD == 2: the linear1 bias table is padded to whole 1-KiB DMA pieces
constexpr int kStackLoopRelLds = REL_MAX_EDGE * 4 + 4 * 32 * 4 + 16;  // REL: the layout's edges, packed | cluster centres [4][32] | edge count —
                                                                     // staged once per launch (RelPersist), behind kStackLoopLds
constexpr int kStackLoopLds = 1024 + 128 + 2304;  // HEAD == 2: tokens [128] | cond token + strong bit [128] | REL: element -> graph
                                                  // node [32] | incidence lists of the graph's nodes (ldm_relation_core.h kRelIncBytes)
static_assert(kRelIncBytes <= 2304, "incidence lists outgrow their LDS slot");
static_assert(kRelScratchFloats * 4 <= 2 * KV_BYTES, "the SGD's scratch lives in the K / V buffers");

__device__ unsigned long long g_stack_phase[16];

__device__ __forceinline__ int stack_lane_id() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// the other half of a row's lane pair (lanes r and r + 32) through v_permlane32_swap: __shfl_xor takes the lane id as a
// bpermute index, which hipcc hoists out of the step loop of HEAD == 2 into one more long-lived VGPR
__device__ __forceinline__ float stack_xor32_swap(float v) {
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return (stack_lane_id() & 32) ? __uint_as_float(sw[0]) : __uint_as_float(sw[1]);
}
#define LDM_XOR32(v) (HEAD == 2 ? stack_xor32_swap(v) : __shfl_xor(v, 32, 64))

// HEAD: 1 = one denoiser pass: rows in, logits out (parity hook + per-step path); 2 = the whole reverse loop: tokens in, tokens out
// REL (HEAD == 2 only): cond=relation — steps with t >= 10 run posterior (+ strong mask) -> SGD on the layout's log-probabilities
// in LDS (ldm_relation_core.h; the 25 elements of the layout couple, so all four wavefronts cooperate between two barriers)
// -> [PAD] disable -> draw, the reference's order (base.py:243-291).  Its own instantiation: the plain loop kernel is untouched.
template <bool TM, int HEAD, bool REL = false>
__global__ __launch_bounds__(256, 1) void stack_stream_k(StackArgs a) {
  static_assert(!REL || HEAD == 2, "the relation tail belongs to the loop kernel");
  constexpr int KS = 29, STAGE = TILE_STAGE, NT2 = 15, NGV = 58;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid_o = threadIdx.x;
  const int wave_o = __builtin_amdgcn_readfirstlane(tid_o >> 6);
  const int b_o = blockIdx.x;

  unsigned long long t0 = 0, t_real0 = 0, s_pro = 0, s_stream = 0, s_core = 0, s_slab = 0, s_ln2 = 0, s_ffn = 0, s_bnd = 0, s_bnd_a = 0, s_bnd_b = 0, s_hsync = 0, s_ssync = 0;
  if constexpr (TM) {
    t0 = __builtin_amdgcn_s_memtime();
    t_real0 = __builtin_amdgcn_s_memrealtime();
  }
  f32x16 acc[NT2];     // the residual stream: x (raw rows) -> seed -> x1 -> x2 of the current layer -> ...
  if constexpr (HEAD == 2) {
    // x_t of the layout's tokens and its cond token | strong << 30 (or -1): LDS behind the tables, for the whole loop
    const stack_kargs_ptr kp = stack_kargs();
    const int S = kp->S;
    int* toks = reinterpret_cast<int*>(smem + 3 * STAGE + 2 * KV_BYTES + (3 * kp->H * 64 + 2 * LN_DP + 512 + kStackLoopB1 + 2 * LN_DP + 512) * 4);
    if (tid_o < 128) {
      const int s = tid_o < S ? tid_o : S - 1;
      const size_t row = (size_t)b_o * S + s;
      toks[tid_o] = kp->post.tokens[row];
      int cc = -1;
      if (kp->post.cond_seq) cc = kp->post.cond_seq[row] | ((kp->post.strong && kp->post.strong[row]) ? (1 << 30) : 0);
      toks[128 + tid_o] = cc;
    }
    if constexpr (REL) {  // element -> node of the layout's relation graph (node 0 = canvas), -1 = [PAD] element
      if (tid_o == 0) {
        int k = 1;
        const int A5 = kp->post.v.n_attr;
        for (int e = 0; e < S / A5; ++e)
          toks[256 + e] = ((kp->post.cond_seq[(size_t)b_o * S + e * A5]) != kp->post.v.pad_id) ? k++ : -1;
      }
      // which edges touch which node, in edge order: once per launch (the K / V buffers serve as scratch)
      const int e0 = kp->rel.edge_off[b_o], ne = kp->rel.edge_off[b_o + 1] - e0;
      relation_incidence(kp->rel, e0, ne, tid_o, S / kp->post.v.n_attr, reinterpret_cast<float*>(smem + 3 * STAGE), toks + 288,
                         reinterpret_cast<unsigned short*>(toks + 288 + kRelIncOffInts), [] { __syncthreads(); });
      // ... and the graph itself: r04's first loop form re-read the edges, the centres, the canvas box and the edge offsets from
      // global memory in every adjusted step (three dependent round trips in front of the SGD)
      unsigned* pke = reinterpret_cast<unsigned*>(toks + kStackLoopLds / 4);
      float* pcen = reinterpret_cast<float*>(pke + REL_MAX_EDGE);
      int* pmeta = reinterpret_cast<int*>(pcen + 128);  // [0] edge count, [1] != 0: an edge does not fit the packed form
      if (tid_o == 0) {
        pmeta[0] = ne;
        pmeta[1] = 0;
      }
      __syncthreads();
      for (int k = tid_o; k < ne && k < REL_MAX_EDGE; k += 256) {
        const unsigned es = (unsigned)kp->rel.edge_src[e0 + k], ed = (unsigned)kp->rel.edge_dst[e0 + k], ea = (unsigned)kp->rel.edge_attr[e0 + k];
        if ((es | ed) >= 64u || ea >= (1u << 20)) atomicOr(&pmeta[1], 1);  // (the general form of the SGD takes such a graph)
        pke[k] = es | (ed << 6) | (ea << 12);
      }
      if (tid_o < 128) pcen[tid_o] = (tid_o & 31) < kp->rel_n_bin ? kp->rel.centres[(tid_o >> 5) * kp->rel_n_bin + (tid_o & 31)] : 0.f;
    }
    __syncthreads();
  }
  unsigned long long t_epi = 0;
  int n_iter = 1;
  if constexpr (HEAD == 2) n_iter = stack_kargs()->n_steps;
  int it = 0;
  do {  // (HEAD < 2: the condition is a constant false — no loop exists in those instantiations)
  // HEAD == 2: everything a step derives from the thread / workgroup coordinates and from the kernel arguments is
  // re-derived per step from values hipcc cannot see through — hoisted out of the step loop, those ~100 uniform
  // addresses and parameters would stay live across a body that already fills the register file (170 SGPR spills and
  // 100 bytes of scratch per lane without this)
  int wave = wave_o, b = b_o;
  if constexpr (HEAD == 2) asm volatile("" : "+s"(wave), "+s"(b));
  // the thread index: HEAD == 2 re-derives it from the hardware lane id at every use (no VGPR carried across the
  // streams, which fill the register file)
  [[maybe_unused]] const int tid = tid_o;
  auto tid_now = [&]() {
    if constexpr (HEAD == 2) return wave * 64 + stack_lane_id();
    else return tid;
  };
  auto&& A = [&]() -> decltype(auto) {
    if constexpr (HEAD == 2) return (*stack_kargs());
    else return (a);
  }();
  char* kvbuf = smem + 3 * STAGE;          // Ks 16 KiB | Vs 16 KiB behind the 3-stage weight ring
  float* sbias = reinterpret_cast<float*>(kvbuf + 2 * KV_BYTES);  // [3*H*64]
  float* sp = sbias + 3 * A.H * 64;        // AdaLN multiplier / shift (2 x LN_DP)
  float* sbo = sp + 2 * LN_DP;             // b_out + W_out b_v + AdaLN shift [512]
  float* sb1 = sbo + 512;                  // linear1 bias [n_chunks*32]
  float* sp2 = sb1 + (HEAD == 2 ? kStackLoopB1 : A.n_chunks * 32);  // norm2 gamma | beta (2 x LN_DP)
  float* sb2 = sp2 + 2 * LN_DP;            // linear2 bias [512], zero beyond N
  [[maybe_unused]] int* toks = reinterpret_cast<int*>(sb2 + 512);   // HEAD == 2: x_t of the layout's tokens
  [[maybe_unused]] int* condc = toks + 128;                         //            cond token | strong << 30 (or -1)
  const int S = A.S, H = A.H, L = A.ls.n_layer;
  unsigned lds0 = (unsigned)(size_t)(lds_char_ptr)smem;
  if constexpr (HEAD == 2) asm volatile("" : "+s"(lds0));
  const unsigned voff = HEAD == 2 ? (unsigned)stack_lane_id() * 16 : (tid & 63) * 16;
  // parameter tables of one layer, global -> LDS: every load is issued before the first ds_write (a loop of dependent
  // load / store pairs pays one L2 round trip per iteration: 15 of them made the layer boundary 17k cycles longer,
  // profiles/r02_call23_*).  Every table is zero beyond N so that padded columns come out as exact zeros without masks.
  auto stage_tables = [&](const auto& w, const float* ada_scale, const float* ada_shift, int tid) {
    float vb[6], v1[8], vs[2], vh[2], vo[2], vg[2], ve[2], v2[2];
#pragma unroll
    for (int k = 0; k < 6; ++k) vb[k] = w.bias_in[tid + 256 * k];  // 3 * 8 heads * 64 = 1536 entries (launcher: H == 8)
#pragma unroll
    for (int k = 0; k < 8; ++k) v1[k] = tid + 256 * k < A.n_chunks * 32 ? w.b1[tid + 256 * k] : 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + 256 * k;
      const bool in = i < A.N;
      vs[k] = in ? ada_scale[i] : -1.f;  // multiplier 1 + scale = 0 beyond N
      vh[k] = in ? ada_shift[i] : 0.f;
      vo[k] = in ? w.b_out[i] : 0.f;
      vg[k] = in ? w.g2[i] : 0.f;
      ve[k] = in ? w.be2[i] : 0.f;
      v2[k] = in ? w.b2[i] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 6; ++k) sbias[tid + 256 * k] = vb[k];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (tid + 256 * k < A.n_chunks * 32) sb1[tid + 256 * k] = v1[k];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + 256 * k;
      sp[i] = 1.0f + vs[k];
      sp[LN_DP + i] = vh[k];
      sbo[i] = vo[k] + vh[k];
      sp2[i] = vg[k];
      sp2[LN_DP + i] = ve[k];
      sb2[i] = v2[k];
    }
  };
  // HEAD == 2: a parameter table image, n_kib 1-KiB pieces dealt round-robin to the four wavefronts (landing is covered by
  // the next s_waitcnt vmcnt(0) of every wavefront + the barrier behind it, which every consumer below already has)
  auto dma_table = [&](const float* g, const float* lds_dst, int n_kib) {
    const unsigned dst = lds0 + (unsigned)(reinterpret_cast<const char*>(lds_dst) - smem);
    for (int k = wave; k < n_kib; k += 4) {
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(dst + k * 1024) : "memory");
      dma_lin<0>(voff, reinterpret_cast<const char*>(g) + (size_t)k * 1024);
    }
  };
  auto dma_first_tiles = [&](const char* img) {  // tiles 0 / 1 of a layer's first head -> stages 0 / 1
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_lin4(voff, img + wave * 8192 + (k >> 1) * STAGE + (k & 1) * 4096,
                                         lds0 + wave * 8192 + (k >> 1) * STAGE + (k & 1) * 4096);
  };
  int t_model = 0;
  if constexpr (HEAD == 2) t_model = stack_kargs()->t_model[it];
#pragma unroll
  for (int i = 8; i < 16; ++i) acc[NT2 - 1][i] = to_agpr(0.f);  // columns 464..479: padding
  if constexpr (HEAD == 2) {
    // ------------------------------------------------------------------ prologue: x = cat_emb[token] + pos[s]
    // (nn_lib.py:204,220) gathered straight into the accumulator layout; the tables (155 x 464 and 125 x 464 floats)
    // are L2 resident
    const int lane = stack_lane_id();
    const int r = lane & 31, hi = lane >> 5;
    const int row = wave * 32 + r;
    const int srow = row < S ? row : S - 1;
    const stack_kargs_ptr kp = stack_kargs();
    // attention-phase tables of layer 0 at this step's timestep (every wavefront is past the previous step's head phase,
    // the last reader of these LDS regions): they land while the embedding is gathered
    dma_table(kp->tbl.att_static, sbias, kStackTblAttStatic / 256);
    dma_table(kp->tbl.att_dyn + (size_t)t_model * L * kStackTblAttDyn, sp, kStackTblAttDyn / 256);
    const float* erow = kp->post.emb + (size_t)toks[srow] * kp->post.D + hi * 4;
    const float* prow = kp->post.pos + (size_t)srow * kp->post.D + hi * 4;
    constexpr int GB = 12;
#pragma unroll
    for (int g0 = 0; g0 < NGV; g0 += GB) {
      float4 re[GB], rp[GB];
#pragma unroll
      for (int i = 0; i < GB; ++i)
        if (g0 + i < NGV) {
          re[i] = *reinterpret_cast<const float4*>(erow + (g0 + i) * 8);
          rp[i] = *reinterpret_cast<const float4*>(prow + (g0 + i) * 8);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < GB; ++i) {
        const int gg = g0 + i;
        if (gg < NGV) {
          const int t = gg >> 2, q0 = (gg & 3) * 4;
          acc[t][q0 + 0] = to_agpr(re[i].x + rp[i].x);
          acc[t][q0 + 1] = to_agpr(re[i].y + rp[i].y);
          acc[t][q0 + 2] = to_agpr(re[i].z + rp[i].z);
          acc[t][q0 + 3] = to_agpr(re[i].w + rp[i].w);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    // ------------------------------------------------------------------ prologue: the rows, raw, in accumulator layout
    // (lane (row, hi) owns columns 8g + 4hi .. +3 of every 8-column group g); every layer, the first included, then
    // starts from the accumulators
    const int lane = stack_lane_id();
    const int r = lane & 31, hi = lane >> 5;
    const int row = wave * 32 + r;
    const size_t m = (size_t)b * S + (row < S ? row : S - 1);
    constexpr int GB = 20;
    const float* rrow = A.x + m * A.ldx + hi * 4;
#pragma unroll
    for (int g0 = 0; g0 < NGV; g0 += GB) {
      float4 raw[GB];
#pragma unroll
      for (int i = 0; i < GB; ++i)
        if (g0 + i < NGV) raw[i] = *reinterpret_cast<const float4*>(rrow + (g0 + i) * 8);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < GB; ++i) {
        const int gg = g0 + i;
        if (gg < NGV) {
          const int t = gg >> 2, q0 = (gg & 3) * 4;
          acc[t][q0 + 0] = to_agpr(raw[i].x);
          acc[t][q0 + 1] = to_agpr(raw[i].y);
          acc[t][q0 + 2] = to_agpr(raw[i].z);
          acc[t][q0 + 3] = to_agpr(raw[i].w);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (TM) s_pro = __builtin_amdgcn_s_memtime() - t0;

  for (int l = 0; l < L; ++l) {
    const auto& w = A.ls.w[l];
    unsigned long long tL = 0;
    if constexpr (TM) tL = __builtin_amdgcn_s_memtime();
    f16x8 xf[KS];        // AdaLN(x) of this layer, fp16 MFMA B fragments (k-slot K order)
    {
      // ---------------------------------------------------------------- layer entry, in registers: acc = x (layer
      // input).  Every wave is past the previous layer's FFN LDS reads and table reads: the ring takes this layer's first
      // tiles, the tables its parameters; then row statistics, AdaLN fragments and the residual seed
      // AdaLN(x) + b_out + W_out b_v from the same registers.
      if constexpr (HEAD == 2) {
        // this layer's attention-phase tables were issued a phase ago (step prologue / previous layer's LN2): own pieces
        // landed, then everybody's; the FFN-phase tables follow the first tiles and land under the attention block
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        dma_first_tiles((const char*)w.img);
        dma_table(stack_kargs()->tbl.ffn + (size_t)l * kStackTblFfn, sb1, kStackTblFfn / 256);
      } else {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        dma_first_tiles((const char*)w.img);
        stage_tables(w, w.ada_scale, w.ada_shift, tid);
        __syncthreads();
      }
      unsigned long long tE1 = 0, tE2 = 0;
      if constexpr (TM) tE1 = __builtin_amdgcn_s_memtime();
      const int lane3 = stack_lane_id();
      const int hi3 = lane3 >> 5;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int t = 0; t < NT2; ++t) {
        const f32x16 tile = acc[t];
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (t * 4 + (i >> 2) < NGV) {
            s1 += tile[i];
            s2 += tile[i] * tile[i];
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      s1 += LDM_XOR32(s1);
      s2 += LDM_XOR32(s2);
      if constexpr (TM) tE2 = __builtin_amdgcn_s_memtime();
      constexpr float kInvN3 = 1.0f / 464.0f;
      const float mean = s1 * kInvN3;
      const float rstd = 1.0f / sqrtf(fmaxf(s2 * kInvN3 - mean * mean, 0.f) + 1e-5f);
      const float ra = rstd, rb = -mean * rstd;
      const float* gmp = sp + hi3 * 4;
      const float* tbp = sbo + hi3 * 4;
#pragma unroll
      for (int t = 0; t < NT2; ++t) {
        f32x16 tile = acc[t];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int gg = t * 4 + g;
          if (gg < NGV) {
            const int ks = gg >> 1, e0 = (gg & 1) * 4;
            const float4 gm = *reinterpret_cast<const float4*>(gmp + gg * 8);
            const float4 sh = *reinterpret_cast<const float4*>(gmp + LN_DP + gg * 8);
            const float4 tb = *reinterpret_cast<const float4*>(tbp + gg * 8);
            const float n0 = fmaf(tile[g * 4 + 0], ra, rb), n1 = fmaf(tile[g * 4 + 1], ra, rb);
            const float n2 = fmaf(tile[g * 4 + 2], ra, rb), n3 = fmaf(tile[g * 4 + 3], ra, rb);
            xf[ks][e0 + 0] = (_Float16)fmaf(n0, gm.x, sh.x);
            xf[ks][e0 + 1] = (_Float16)fmaf(n1, gm.y, sh.y);
            xf[ks][e0 + 2] = (_Float16)fmaf(n2, gm.z, sh.z);
            xf[ks][e0 + 3] = (_Float16)fmaf(n3, gm.w, sh.w);
            tile[g * 4 + 0] = fmaf(n0, gm.x, tb.x);
            tile[g * 4 + 1] = fmaf(n1, gm.y, tb.y);
            tile[g * 4 + 2] = fmaf(n2, gm.z, tb.z);
            tile[g * 4 + 3] = fmaf(n3, gm.w, tb.w);
            if (gg & 1) asm volatile("" : "+v"(xf[ks]));
          }
        }
        asm volatile("" : "+a"(tile));
        acc[t] = tile;
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (TM) {
        s_bnd += __builtin_amdgcn_s_memtime() - tL;
        s_bnd_a += tE1 - tL;
        s_bnd_b += tE2 - tE1;
      }
    }
    {
      // ---------------------------------------------------------------- attention block, head by head.  Every phase of
      // a head re-derives its lane coordinates and LDS addresses from the hardware lane id (an asm hipcc cannot hoist):
      // the stream's 21 address registers are then dead during the attention core — the register peak of the kernel,
      // with the 240 accumulator AGPRs and the 116 fragment VGPRs alive around it — and the core's and the slab pair's
      // during the stream.
      // Whatever hipcc parked outside the register file around the layer entry comes back HERE: a use of every fragment
      // and a builtin vmcnt(0) (which its scoreboard sees), so that it places no s_waitcnt vmcnt(n) of its own inside the
      // streams, where such a wait would also wait for the weight DMA it cannot see (2 300 cycles per head:
      // profiles/r02_call23_*)
#pragma unroll
      for (int k = 0; k < KS; ++k) asm volatile("" : "+v"(xf[k]));
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __builtin_amdgcn_sched_barrier(0);
      // tiles 0 and 1 of the first head have landed (own pieces, then everybody's)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const unsigned kv0 = lds0 + 3 * STAGE;
      for (int h = 0; h < H; ++h) {
        unsigned long long tA = 0, tB = 0, tC = 0;
        if constexpr (TM) tA = __builtin_amdgcn_s_memtime();
        const char* gimg = (const char*)w.img + (size_t)h * 8 * STAGE + wave * 8192;
        f16x8 qf[4];
        {
          const int lane = stack_lane_id();
          const int r = lane & 31, hi = lane >> 5;
          const int row_in = wave * 32 + r;
          HeadStream<TM, true> HS;
          HS.xf = xf;
          HS.qf = qf;
          HS.voff = voff;
          HS.lds_w = lds0 + wave * 8192;
          HS.H = H;
          HS.h = h;
          HS.gimg = gimg;
          HS.a_bias = lds0 + (unsigned)(reinterpret_cast<char*>(sbias) - smem) + hi * 16;
#pragma unroll
          for (int k = 0; k < 8; ++k) HS.aW[k] = lds0 + r * RKB + ((((k << 1) | hi) ^ (r & 15)) << 4);
          const int sw = (row_in >> 1) & 7;
          HS.aK[0] = kv0 + row_in * 128 + ((hi ^ sw) << 4);
          HS.aK[1] = kv0 + row_in * 128 + (((2 + hi) ^ sw) << 4);
          HS.aV[0] = kv0 + KV_BYTES + r * 256 + (((wave * 4 + hi) ^ (r & 15)) << 4);
          HS.aV[1] = kv0 + KV_BYTES + r * 256 + (((wave * 4 + 2 + hi) ^ (r & 15)) << 4);
          HS.run();
          if constexpr (TM) s_hsync += HS.t_sync;
        }
        if constexpr (TM) tB = __builtin_amdgcn_s_memtime();
        f16x8 nf[4];
        {
          const int lane = stack_lane_id();
          const int r = lane & 31, hi = lane >> 5;
          AttnCoreV AC;
          AC.qf = qf;
          AC.aKr = kv0 + r * 128 + ((hi ^ ((r >> 1) & 7)) << 4);
          AC.aVr = kv0 + KV_BYTES + r * 256 + ((hi ^ (r & 15)) << 4);
          AC.scale_log2e = A.scale_log2e;
          AC.S = S;
          AC.hi = hi;
          AC.run(nf);
        }
        if constexpr (TM) tC = __builtin_amdgcn_s_memtime();
        {
          const int lane = stack_lane_id();
          const int r = lane & 31, hi = lane >> 5;
          SlabPair<NT2, TM> SP;
          SP.acc = acc;
          SP.voff = voff;
          SP.lds_w = lds0 + wave * 8192;
          SP.aS[0] = lds0 + r * 64 + (((0 + hi) ^ ((r >> 2) & 3)) << 4);
          SP.aS[1] = lds0 + r * 64 + (((2 + hi) ^ ((r >> 2) & 3)) << 4);
          SP.nf = nf;
          SP.gnext = gimg + (size_t)8 * STAGE;
          SP.run();
          if constexpr (TM) s_ssync += SP.t_sync;
        }
        if constexpr (TM) {
          s_stream += tB - tA;
          s_core += tC - tB;
          s_slab += __builtin_amdgcn_s_memtime() - tC;
        }
      }
    }
    unsigned long long t_att = 0, t_ln2 = 0;
    if constexpr (TM) t_att = __builtin_amdgcn_s_memtime();
    // acc = x1.  Everybody is done with the attention ring / K,V buffers after this barrier (and the last head's
    // padding prefetch has landed): the FFN ring (2 x 64 KiB at LDS 0) takes their place.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {  // FFN chunk 0 -> stage 0 (this wave's 16 KiB); lands while LN2 runs
      const char* g0 = (const char*)w.ffn_img + wave * 16384;
#pragma unroll
      for (int k = 0; k < 4; ++k) dma_lin4(voff, g0 + k * 4096, lds0 + wave * 16384 + k * 4096);
    }
    if constexpr (HEAD == 2) {
      // the NEXT consumer's attention-phase tables (every wavefront is past this layer's heads and its layer entry, the
      // readers of these regions): the next layer's, or behind the last layer the vocabulary head's LayerNorm affine
      const stack_kargs_ptr kp = stack_kargs();
      if (l + 1 < L) {
        dma_table(kp->tbl.att_static + (size_t)(l + 1) * kStackTblAttStatic, sbias, kStackTblAttStatic / 256);
        dma_table(kp->tbl.att_dyn + ((size_t)t_model * L + l + 1) * kStackTblAttDyn, sp, kStackTblAttDyn / 256);
      } else {
        dma_table(kp->tbl.head, sp, kStackTblAttDyn / 256);
      }
    }
    const int lane2 = stack_lane_id();
    const int r2 = lane2 & 31, hi2 = lane2 >> 5;
    f16x8 xf2[KS];
    {
      // LN2 statistics of the row (this lane's half + lane^32), normalised fp16 fragments in k-slot order (groups
      // 2ks, 2ks+1 of the accumulator layout ARE fragment ks), GEMM2 seed acc = x1 + b2
      // (tile by tile: whole-tuple copies between the accumulator AGPRs and arch VGPRs, no sub-register asm operands —
      //  with element-wise accumulator updates hipcc split the live ranges of half the tiles through scratch)
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int t = 0; t < NT2; ++t) {
        const f32x16 tile = acc[t];
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (t * 4 + (i >> 2) < NGV) {
            s1 += tile[i];
            s2 += tile[i] * tile[i];
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      s1 += LDM_XOR32(s1);
      s2 += LDM_XOR32(s2);
      constexpr float kInvN = 1.0f / 464.0f;  // N = 464 (launcher)
      const float mean = s1 * kInvN;
      const float rstd = 1.0f / sqrtf(fmaxf(s2 * kInvN - mean * mean, 0.f) + 1e-5f);
      const float* gp = sp2 + hi2 * 4;
      const float* bp = sb2 + hi2 * 4;
#pragma unroll
      for (int t = 0; t < NT2; ++t) {
        f32x16 tile = acc[t];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int gg = t * 4 + g;
          if (gg < NGV) {
            const int ks = gg >> 1, e0 = (gg & 1) * 4;
            const float4 ga = *reinterpret_cast<const float4*>(gp + gg * 8);
            const float4 be = *reinterpret_cast<const float4*>(gp + LN_DP + gg * 8);
            const float4 bb = *reinterpret_cast<const float4*>(bp + gg * 8);
            const float v0 = tile[g * 4 + 0], v1 = tile[g * 4 + 1], v2 = tile[g * 4 + 2], v3 = tile[g * 4 + 3];
            xf2[ks][e0 + 0] = (_Float16)fmaf((v0 - mean) * rstd, ga.x, be.x);
            xf2[ks][e0 + 1] = (_Float16)fmaf((v1 - mean) * rstd, ga.y, be.y);
            xf2[ks][e0 + 2] = (_Float16)fmaf((v2 - mean) * rstd, ga.z, be.z);
            xf2[ks][e0 + 3] = (_Float16)fmaf((v3 - mean) * rstd, ga.w, be.w);
            tile[g * 4 + 0] = v0 + bb.x;
            tile[g * 4 + 1] = v1 + bb.y;
            tile[g * 4 + 2] = v2 + bb.z;
            tile[g * 4 + 3] = v3 + bb.w;
            if (gg & 1) asm volatile("" : "+v"(xf2[ks]));
          }
        }
        asm volatile("" : "+a"(tile));
        acc[t] = tile;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (TM) t_ln2 = __builtin_amdgcn_s_memtime();
    {
      // ---- FFN chunk loop: one continuous LDS-read / MFMA pipeline (ldm_pipes.h FfnStream)
      unsigned relW1[8], relW2[2];
#pragma unroll
      for (int k = 0; k < 8; ++k) relW1[k] = r2 * RKB + ((((k << 1) | hi2) ^ (r2 & 15)) << 4);
#pragma unroll
      for (int sx = 0; sx < 2; ++sx) relW2[sx] = r2 * 64 + (((2 * sx + hi2) ^ ((r2 >> 2) & 3)) << 4);
      const unsigned relB = lds0 + (unsigned)(reinterpret_cast<char*>(sb1) - smem) + hi2 * 16;
      FfnStream<KS, NT2, 2, false, 6, true> F;
      F.xf = xf2;
      F.acc = acc;
      F.voff = voff;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // chunk 0 (own pieces), then everybody's
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int k = 0; k < 8; ++k) F.aW1[k] = lds0 + relW1[k];
#pragma unroll
      for (int sx = 0; sx < 2; ++sx) F.aW2[sx] = lds0 + relW2[sx];
      F.ab_next = relB;
      // (same fence as in front of the head loop: every fragment back in its registers, hipcc's scoreboard drained)
#pragma unroll
      for (int k = 0; k < KS; ++k) asm volatile("" : "+v"(xf2[k]));
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __builtin_amdgcn_sched_barrier(0);
      F.read_bias();
      F.template prologue<0>();
      {
        const f16x8 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f,
                         (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        F.pf[0] = F.pf[1] = F.pfn[0] = F.pfn[1] = z;  // iteration 0's GEMM2 multiplies zero fragments (and a zero W2 half)
      }
      // software-pipelined chunk stream (ldm_pipes.h FfnStream PIPE): iteration c = GEMM1 of chunk c + GEMM2 of chunk
      // c - 1 on stage c = W1 tile c | W2 slab c - 1 (ldm_pack::pack_ffn_image_pipelined); n_chunks + 1 iterations
      const char* fimg = (const char*)w.ffn_img;
      for (int c = 0; c <= A.n_chunks; ++c) {
#if !defined(LDM_ABL_FFN_WINDOW)
        F.gnext = fimg + (size_t)(c == A.n_chunks ? 0 : c + 1) * FFN_STAGE + wave * 16384;
#elif LDM_ABL_FFN_WINDOW == 1
        // MEASUREMENT builds only (tools/build_measurement_variants.py, profiles/r04_call26_*; wrong numbers by design, never the
        // shipped library): the FFN weight stream — 65 % of the 23.4 MB a workgroup-step pulls — re-reads stage 0 of the layer's
        // image: a 64-KiB window that the XCD's L2 serves (no fabric / Infinity-Cache traffic) ...
        F.gnext = fimg + wave * 16384;
#else
        // ... or one 16-KiB piece for all four waves, which the CU's vector L1 serves (no L2 -> CU traffic either)
        F.gnext = fimg;
#endif
        F.mnext = lds0 + ((c + 1) & 1) * FFN_STAGE + wave * 16384;
        F.ab_next = relB + (c + 1 >= A.n_chunks ? 0 : c + 1) * 128;
        F.template step<0, true>();
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    unsigned long long t_ffn = 0;
    if constexpr (TM) {
      t_ffn = __builtin_amdgcn_s_memtime();
      s_ln2 += t_ln2 - t_att;
      s_ffn += t_ffn - t_ln2;
    }
  }
  if constexpr (TM) t_epi = __builtin_amdgcn_s_memtime();
  {
    // ---- vocabulary head in the same workgroup: logits = LN_head(x_out) · Wh^T.  The rows never leave the registers:
    // statistics and the normalised fp16 fragments (k-slot K order) from the accumulators, then n_head_tiles 32-class
    // weight tiles through a 4-stage ring in the LDS the FFN ring used (TilePipe: 29 MFMAs per tile), 16-byte stores of
    // the logits.  Replaces the row / statistics store and the separate head GEMM launch (which cannot overlap the
    // other lane's stack kernel: its workgroups own whole CUs).
    __builtin_amdgcn_s_barrier();  // every wave is past its FFN LDS reads and the last layer's table reads
    asm volatile("" ::: "memory");
    auto dma_head_tile = [&](int ht) {
      const char* g = A.head_img + (size_t)ht * STAGE + wave * 8192;
      const unsigned l = lds0 + (unsigned)(ht & 3) * STAGE + wave * 8192;
      dma_lin4(voff, g, l);
      dma_lin4(voff, g + 4096, l + 4096);
    };
    for (int ht = 0; ht < 3 && ht < A.n_head_tiles; ++ht) dma_head_tile(ht);
    if constexpr (HEAD != 2) {  // (HEAD == 2: the image was issued at the last layer's LN2; the FFN stream ends on vmcnt(0))
      for (int i = tid_now(); i < LN_DP; i += 256) {
        sp[i] = i < A.N ? A.head_g[i] : 0.f;
        sp[LN_DP + i] = i < A.N ? A.head_b[i] : 0.f;
      }
    }
    __syncthreads();
    const int lane3 = stack_lane_id();
    const int r3 = lane3 & 31, hi3 = lane3 >> 5;
    const int row3 = wave * 32 + r3;
    f16x8 xf3[KS];
    {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int t = 0; t < NT2; ++t) {
        const f32x16 tile = acc[t];
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (t * 4 + (i >> 2) < NGV) {
            s1 += tile[i];
            s2 += tile[i] * tile[i];
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      s1 += LDM_XOR32(s1);
      s2 += LDM_XOR32(s2);
      constexpr float kInvN3 = 1.0f / 464.0f;
      const float mean = s1 * kInvN3;
      const float rstd = 1.0f / sqrtf(fmaxf(s2 * kInvN3 - mean * mean, 0.f) + 1e-5f);
      const float ra = rstd, rb = -mean * rstd;
      const float* gmp = sp + hi3 * 4;
#pragma unroll
      for (int t = 0; t < NT2; ++t) {
        const f32x16 tile = acc[t];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int gg = t * 4 + g;
          if (gg < NGV) {
            const int ks = gg >> 1, e0 = (gg & 1) * 4;
            const float4 gm = *reinterpret_cast<const float4*>(gmp + gg * 8);
            const float4 be = *reinterpret_cast<const float4*>(gmp + LN_DP + gg * 8);
            xf3[ks][e0 + 0] = (_Float16)fmaf(fmaf(tile[g * 4 + 0], ra, rb), gm.x, be.x);
            xf3[ks][e0 + 1] = (_Float16)fmaf(fmaf(tile[g * 4 + 1], ra, rb), gm.y, be.y);
            xf3[ks][e0 + 2] = (_Float16)fmaf(fmaf(tile[g * 4 + 2], ra, rb), gm.z, be.z);
            xf3[ks][e0 + 3] = (_Float16)fmaf(fmaf(tile[g * 4 + 3], ra, rb), gm.w, be.w);
            if (gg & 1) asm volatile("" : "+v"(xf3[ks]));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) asm volatile("" : "+v"(xf3[k]));
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_sched_barrier(0);
    const bool valid3 = row3 < S;
    float* lrow = HEAD == 2 ? nullptr : A.logits + ((size_t)b * S + (valid3 ? row3 : S - 1)) * A.ldl + hi3 * 4;
    unsigned relW[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) relW[k] = lds0 + r3 * RKB + ((((k << 1) | hi3) ^ (r3 & 15)) << 4);
    if constexpr (HEAD == 2) {
      // ---- the step's tail in the same workgroup.  The five 32-class tiles stay in the (dead) residual accumulators;
      // when the last one is done the weight ring is free and takes the layout's logits as [token][kPostLd] floats.
      constexpr int NHT = 5;  // launcher: n_head_tiles == 5
#pragma unroll
      for (int ht = 0; ht < NHT; ++ht) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (ht + 3 < NHT) dma_head_tile(ht + 3);
        TilePipe<KS, 8> TP;
        TP.xf = xf3;
        TP.voff = voff;
#pragma unroll
        for (int k = 0; k < 8; ++k) TP.aW[k] = relW[k] + (unsigned)(ht & 3) * STAGE;
        TP.template run<false, false>();
        f32x16 lg = TP.acc;
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(lg));
        asm volatile("" : "+a"(lg));
        acc[ht] = lg;
      }
      __builtin_amdgcn_s_barrier();  // nobody reads the ring any more
      asm volatile("" ::: "memory");
      const stack_kargs_ptr kp = stack_kargs();
      const auto& p = kp->post;
      float* lgs = reinterpret_cast<float*>(smem);
      float4* rstat = reinterpret_cast<float4*>(smem + kPostRows);          // [128] (max, lse, max |x|, -)
      float* ssch = reinterpret_cast<float*>(smem + kPostRows + 2048) + wave * 128;  // this wave's copy: [n_attr][10] | [n_attr][2][5] q terms
      const int Cm1 = p.v.n_class - 1;
      const int t_post = kp->t_post[it];
      {
        // predict_start's log-softmax over the classes [0, C - 1) (base.py:131-144), fp32 (fast numerics mode): row
        // maximum and log-sum-exp from the registers of the row's two lanes (D[i = class][j = row]: lane (row, hi) holds
        // classes 32 ht + 8 rq + 4 hi + i), then the raw logits go to LDS
        float mx = -INFINITY, am = 0.f;
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht) {
          const f32x16 tile = acc[ht];
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (ht * 32 + (i >> 2) * 8 + hi3 * 4 + (i & 3) < Cm1) {
              mx = fmaxf(mx, tile[i]);
              am = fmaxf(am, fabsf(tile[i]));
            }
        }
        mx = fmaxf(mx, LDM_XOR32(mx));
        am = fmaxf(am, LDM_XOR32(am));
        float se = 0.f;
        float* mine = lgs + row3 * kPostLd + hi3 * 4;
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht) {
          const f32x16 tile = acc[ht];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if (ht * 32 + (i >> 2) * 8 + hi3 * 4 + (i & 3) < Cm1) se += __expf(tile[i] - mx);
            mine[ht * 32 + (i >> 2) * 8 + (i & 3)] = tile[i];
          }
        }
        se += LDM_XOR32(se);
        if (hi3 == 0) rstat[row3] = make_float4(mx, __logf(se), am, 0.f);
        // the step's schedule scalars of every attribute (constrained.py:81-90,114)
        const int T1 = p.T + 1, u = (t_post - 1 + T1) % T1;
        if (lane3 < p.v.n_attr * 10) {
          const int at = lane3 / 10, k = lane3 - at * 10;
          const int kind = k == 0 ? kLogAt : k == 1 ? kLogBt : k == 2 ? kLogCt : k == 3 ? kLogCumAt : k == 4 ? kLogCumBt
                         : k == 5 ? kLogCumCt : k == 6 ? kLogCumAt : k == 7 ? kLogCumBt : k == 8 ? kLogCumCt : kLog1mCumCt;
          ssch[lane3] = p.sched[((size_t)kind * p.v.n_attr + at) * T1 + (k < 6 ? t_post : u)];
        }
        // ... and the body of every attribute's sub-vocabulary (indexed per lane below: from LDS, not from a private copy)
        if (lane3 < p.v.n_attr) {
          reinterpret_cast<int*>(ssch)[100 + 2 * lane3] = p.v.start[lane3];
          reinterpret_cast<int*>(ssch)[101 + 2 * lane3] = p.v.count[lane3];
        }
      }
      // rows 32 wave .. 32 wave + 31 were written by THIS wavefront and are read by it alone: LDS operations of a
      // wavefront complete in order
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      {
        // q(x_t | x_0), q(x_t | x_{t-1}) per (attribute, x_t is / is not [MASK]): ten variants per step instead of one
        // evaluation (12 transcendentals) per token
        const int lane5 = stack_lane_id();
        if (lane5 < 2 * p.v.n_attr) {
          const float* sc10 = ssch + (lane5 >> 1) * 10;
          const ldm_post::StepSchedule sc{sc10[0], sc10[1], sc10[2], sc10[3], sc10[4],
                                          sc10[5], sc10[6], sc10[7], sc10[8], sc10[9]};
          const ldm_post::QTerms k = ldm_post::q_terms(ldm_post::DppGroup<16, true>{0}, (lane5 & 1) != 0, sc);
          float* o = ssch + 50 + lane5 * 5;  // (launcher: n_attr == 5)
          o[0] = k.qt_same; o[1] = k.qt_other; o[2] = k.q1_same; o[3] = k.q1_other; o[4] = k.q1_mask;
        }
      }
      // rows 32 wave .. 32 wave + 31 were written by THIS wavefront and are read by it alone: LDS operations of a
      // wavefront complete in order
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      {
        const int lane4 = stack_lane_id();
        const int grp = lane4 >> 4;
        const ldm_post::DppGroup<16, true> g{lane4 & 15};
        const ldm_post::SlotMap<16, 3, true> m{lane4 & 15};
        const bool last = it + 1 == n_iter;
        // REL: this step adjusts the log-probabilities (t >= 10, logit_adjustment.py:107)?  If so the loop below stops at the
        // posterior (log-probabilities -> the token's own LDS row, slot order) and the draw follows the SGD further down
        bool adjust = false;
        if constexpr (REL) adjust = kp->t_model[it] >= 10 && kp->rel.num_update > 0;
        [[maybe_unused]] const int* node_of = condc + 128;
#pragma unroll 2
        for (int rd = 0; rd < 8; ++rd) {
          const int s = wave * 32 + rd * 4 + grp;
          if (s < S) {
            const int attr = s % p.v.n_attr;
            const float* sc10 = ssch + attr * 10;
            const ldm_post::StepSchedule sc{sc10[0], sc10[1], sc10[2], sc10[3], sc10[4],
                                            sc10[5], sc10[6], sc10[7], sc10[8], sc10[9]};
            const int cc = condc[s];
            ldm_post::TokenArgs ta{};
            ta.tok = toks[s];
            ta.start = reinterpret_cast<const int*>(ssch)[100 + 2 * attr];
            ta.count = reinterpret_cast<const int*>(ssch)[101 + 2 * attr];
            ta.pad_id = p.v.pad_id;
            ta.mask_id = p.v.mask_id;
            ta.n_class = p.v.n_class;
            ta.cond_tok = cc < 0 ? -1 : (cc & 0x3fffffff);
            ta.strong = cc >= 0 && (cc >> 30) != 0;
            ta.weak = p.weak ? p.weak + (size_t)b * p.v.n_class * S + s : nullptr;  // (B, C, S)
            ta.weak_stride = S;
            ta.pad_disable = !adjust && p.pad_disable && cc >= 0 && attr != 0 && ta.cond_tok != p.v.pad_id;  // base.py:272-284
            ta.kind = p.kind;
            ta.temperature = p.temperature;
            ta.top_p = p.top_p;
            ta.top_k = p.top_k;
            ta.pos = (uint32_t)s;
            ta.step = (uint32_t)(p.step + it);
            ta.layout = p.rng[1] + (uint64_t)p.layout_off + (uint64_t)b;
            ta.seed = p.rng[0];
            bool shortcut = ldm_post::strong_shortcut(ta);  // conditioned token: every sampler returns it (ldm_post_token.h)
            // (... unless the SGD moves its row: a strong-masked bbox token of a graph node, kernels_relation.hip)
            if constexpr (REL) shortcut = shortcut && !(adjust && attr != 0 && node_of[s / p.v.n_attr] > 0);
            if (shortcut) {
              if (g.lane() == 0) {
                toks[s] = ta.cond_tok;
                if (kp->inter) kp->inter[((size_t)it * kp->inter_ld + b) * S + s] = ta.cond_tok;
                if (last) p.tokens_out[(size_t)b * S + s] = ta.cond_tok;
              }
              continue;
            }
            float* lrow_s = lgs + s * kPostLd;
            const float4 rs = rstat[s];
            float l0[3], lp[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              const int c = m.cls(ta, j);
              l0[j] = (m.valid(ta, j) && c < Cm1) ? ldm_post::l0_f32(lrow_s[c], rs.x, rs.y) : -70.0f;
            }
            const float* qk = ssch + 50 + (attr * 2 + (ta.tok == ta.mask_id ? 1 : 0)) * 5;
            float k0 = qk[0], k1 = qk[1], k2 = qk[2], k3 = qk[3], k4 = qk[4];
            asm volatile("" : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3), "+v"(k4));  // (registers, not a private array hipcc indexes)
            const ldm_post::QTerms k{k0, k1, k2, k3, k4};
            ldm_post::token_log_probs(g, m, ta, sc, k, l0, lp);
            if constexpr (REL) {
              if (adjust) {  // (the row's logits are in registers by now: every lane of the group has read its l0)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                  if (m.valid(ta, j)) lrow_s[m.sidx(j)] = lp[j];
                continue;
              }
            }
            // (scratch of top-k / top-p: the token's own row — its logits are in registers by now)
            const ldm_post::Draw d = ldm_post::draw_token(g, m, ta, lp, lrow_s, lrow_s + 48, false);
            if (g.lane() == 0) {
              toks[s] = d.token;
              if (kp->inter) kp->inter[((size_t)it * kp->inter_ld + b) * S + s] = d.token;
              if (last) p.tokens_out[(size_t)b * S + s] = d.token;
              if (p.tie_flags && d.gap < fmaxf(p.tie_rel * rs.z, p.tie_abs)) p.tie_flags[(size_t)it * kp->tie_ld + b] = 1;
            }
          }
        }
        if constexpr (REL) {
          if (adjust) {
            // ---- the SGD of logit_adjustment.update on the layout's bbox tokens: element e, coordinate x = token e A + 1 + x,
            // whose 32 body bins are the first 32 floats of its row (slot order); softmax scratch at + 96 (behind the
            // samplers' 2 x 48); the SGD's own scratch in the K / V buffers, dead until the next step's first head
            const int A5 = p.v.n_attr;
            float* rscr = reinterpret_cast<float*>(smem + 3 * STAGE);
            const int tid5 = wave * 64 + stack_lane_id();
            const unsigned* pke = reinterpret_cast<const unsigned*>(toks + kStackLoopLds / 4);
            const float* pcen = reinterpret_cast<const float*>(pke + REL_MAX_EDGE);
            if (tid5 < 4) rscr[kRelBboxOff + tid5] = pcen[tid5 * 32 + kp->rel.canvas_bins[tid5]];  // canvas: one-hot expectation
            if (tid5 < 32) reinterpret_cast<int*>(rscr + kRelNodeOff)[tid5] = node_of[tid5];
            __syncthreads();  // every wavefront's posterior rows are in LDS
            const int* pmeta = reinterpret_cast<const int*>(pcen + 128);
            const int ne = pmeta[0];
            auto lg_at = [&](int e, int x) { return lgs + (e * A5 + 1 + x) * kPostLd; };
            auto pr_at = [&](int e, int x) { return lgs + (e * A5 + 1 + x) * kPostLd + 96; };
            const int* inc_off = node_of + 32;
            const unsigned short* inc = reinterpret_cast<const unsigned short*>(node_of + 32 + kRelIncOffInts);
            if (ne <= REL_MAX_EDGE && inc_off[0] >= 0 && pmeta[1] == 0)
              relation_sgd<true>(kp->rel, 0, ne, tid5, S / A5, kp->rel_n_bin, lg_at, pr_at, rscr, inc_off, inc, RelPersist{pke, pcen},
                                 [] { __syncthreads(); });
            else  // (more edges than the staged form holds: the general form, edges re-staged block by block)
              relation_sgd<false>(kp->rel, kp->rel.edge_off[b], ne, tid5, S / A5, kp->rel_n_bin, lg_at, pr_at, rscr, inc_off, inc,
                                  RelPersist{nullptr, nullptr}, [] { __syncthreads(); });
            // ---- [PAD] disable + draw from the adjusted rows (base.py:272-291)
#pragma unroll 1
            for (int rd = 0; rd < 8; ++rd) {
              const int s = wave * 32 + rd * 4 + grp;
              if (s >= S) continue;
              const int attr = s % A5;
              const int cc = condc[s];
              ldm_post::TokenArgs ta{};
              ta.tok = toks[s];
              ta.start = reinterpret_cast<const int*>(ssch)[100 + 2 * attr];
              ta.count = reinterpret_cast<const int*>(ssch)[101 + 2 * attr];
              ta.pad_id = p.v.pad_id;
              ta.mask_id = p.v.mask_id;
              ta.n_class = p.v.n_class;
              ta.cond_tok = cc < 0 ? -1 : (cc & 0x3fffffff);
              ta.strong = cc >= 0 && (cc >> 30) != 0;
              ta.pad_disable = p.pad_disable && cc >= 0 && attr != 0 && ta.cond_tok != p.v.pad_id;
              ta.kind = p.kind;
              ta.temperature = p.temperature;
              ta.top_p = p.top_p;
              ta.top_k = p.top_k;
              ta.pos = (uint32_t)s;
              ta.step = (uint32_t)(p.step + it);
              ta.layout = p.rng[1] + (uint64_t)p.layout_off + (uint64_t)b;
              ta.seed = p.rng[0];
              if (ldm_post::strong_shortcut(ta) && !(attr != 0 && node_of[s / A5] > 0)) continue;  // stored above
              float* lrow_s = lgs + s * kPostLd;
              float lp[3];
#pragma unroll
              for (int j = 0; j < 3; ++j) lp[j] = m.valid(ta, j) ? lrow_s[m.sidx(j)] : -INFINITY;
              ldm_post::pad_disable_only(m, ta, lp);
              g.sync();  // (the row is about to become the samplers' scratch)
              const ldm_post::Draw d = ldm_post::draw_token(g, m, ta, lp, lrow_s, lrow_s + 48, false);
              if (g.lane() == 0) {
                toks[s] = d.token;
                if (kp->inter) kp->inter[((size_t)it * kp->inter_ld + b) * S + s] = d.token;
                if (last) p.tokens_out[(size_t)b * S + s] = d.token;
              }
            }
            __syncthreads();  // the K / V buffers and the rows go back to the next step
          }
        }
      }
    } else
    for (int ht = 0; ht < A.n_head_tiles; ++ht) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // tile ht (and every earlier piece) has landed
      __builtin_amdgcn_s_barrier();                                // ... everybody's; tile ht - 1 is read by nobody any more
      asm volatile("" ::: "memory");
      if (ht + 3 < A.n_head_tiles) dma_head_tile(ht + 3);          // -> stage of tile ht - 1
      TilePipe<KS, 8> TP;
      TP.xf = xf3;
      TP.voff = voff;
#pragma unroll
      for (int k = 0; k < 8; ++k) TP.aW[k] = relW[k] + (unsigned)(ht & 3) * STAGE;
      TP.template run<false, false>();
      // D[i = class][j = row]: lane (row, hi) holds classes 32 ht + 8 rq + 4 hi + i
      f32x16 lg = TP.acc;
      asm volatile("s_nop 7\n\ts_nop 7" : "+v"(lg));
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
        if (valid3)
          *reinterpret_cast<float4*>(lrow + ht * 32 + rq * 8) = make_float4(lg[rq * 4 + 0], lg[rq * 4 + 1], lg[rq * 4 + 2], lg[rq * 4 + 3]);
    }
  }
  } while (HEAD == 2 && ++it < n_iter);  // step loop
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last FFN prefetch must land before the LDS is released
  if constexpr (TM) {
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    const unsigned long long t_real1 = __builtin_amdgcn_s_memrealtime();
    if (tid_o == 0) {
      atomicAdd(&g_stack_phase[0], 1ull);
      atomicAdd(&g_stack_phase[1], t_end - t0);
      atomicAdd(&g_stack_phase[2], t_real1 - t_real0);
      atomicAdd(&g_stack_phase[3], s_pro);
      atomicAdd(&g_stack_phase[4], s_stream);
      atomicAdd(&g_stack_phase[5], s_core);
      atomicAdd(&g_stack_phase[6], s_slab);
      atomicAdd(&g_stack_phase[7], s_ln2);
      atomicAdd(&g_stack_phase[8], s_ffn);
      atomicAdd(&g_stack_phase[9], s_bnd);
      atomicAdd(&g_stack_phase[10], t_end - t_epi);
      atomicAdd(&g_stack_phase[11], s_bnd_a);
      atomicAdd(&g_stack_phase[12], s_bnd_b);
      atomicAdd(&g_stack_phase[13], s_hsync);
      atomicAdd(&g_stack_phase[14], s_ssync);
    }
  }
}

void launch_stack_stream(const FusedLayerSet& ls, int F, float* x, int ldx, int B, int S, int H, int dh,
                         const StackHead& head, hipStream_t st) {
  const int lds = 3 * TILE_STAGE + 2 * KV_BYTES + (3 * H * 64 + 2 * LN_DP + 512 + F + 2 * LN_DP + 512) * 4;
  static const bool tm = knob_int("LDM_ATTN_TM", 0) != 0;
  auto kern = tm ? stack_stream_k<true, 1> : stack_stream_k<false, 1>;
  allow_big_lds((const void*)kern);
  StackArgs a{};
  a.ls = ls; a.x = x; a.ldx = ldx; a.N = ldx; a.S = S; a.H = H; a.n_chunks = F / 32;
  a.scale_log2e = 1.4426950408889634f / sqrtf((float)dh);
  a.head_img = (const char*)head.img; a.head_g = head.g; a.head_b = head.b;
  a.logits = head.logits; a.ldl = head.ldl; a.n_head_tiles = head.n_tiles;
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, st, a);
}

// HEAD == 2: the whole reverse loop, one workgroup per layout (no probe variant: it would be the one instantiation with
// spills).  The caller has checked: 5 head tiles, live sub-vocabularies <= 48 classes, S <= 128.
void launch_stack_loop(const FusedLayerSet& ls, int F, int N, int B, int S, int H, int dh, const StackHead& head,
                       const StackLoop& lp, hipStream_t st) {
  const int lds = 3 * TILE_STAGE + 2 * KV_BYTES + (3 * H * 64 + 2 * LN_DP + 512 + kStackLoopB1 + 2 * LN_DP + 512) * 4 + kStackLoopLds +
                  (lp.rel ? kStackLoopRelLds : 0);
  auto kern = lp.rel ? stack_stream_k<false, 2, true> : stack_stream_k<false, 2, false>;
  allow_big_lds((const void*)kern);
  StackArgs a{};
  if (lp.rel) {
    a.rel.edge_off = lp.rel->edge_off; a.rel.edge_src = lp.rel->edge_src; a.rel.edge_dst = lp.rel->edge_dst;
    a.rel.edge_attr = lp.rel->edge_attr; a.rel.centres = lp.rel->centres;
    for (int x = 0; x < 4; ++x) a.rel.canvas_bins[x] = lp.rel->canvas_bins[x];
    a.rel.step = lp.rel->step; a.rel.num_update = lp.rel->num_update;
    a.rel_n_bin = lp.rel->n_bin;
  }
  a.ls = ls; a.ldx = N; a.N = N; a.S = S; a.H = H; a.n_chunks = F / 32;
  a.scale_log2e = 1.4426950408889634f / sqrtf((float)dh);
  a.head_img = (const char*)head.img; a.head_g = head.g; a.head_b = head.b; a.n_head_tiles = head.n_tiles;
  a.post = *lp.post; a.adaln = lp.adaln; a.tbl = lp.tables; a.inter = lp.inter;
  a.n_steps = lp.n_steps; a.inter_ld = lp.inter_ld; a.tie_ld = lp.tie_ld;
  for (int i = 0; i < lp.n_steps && i < kStackLoopMaxSteps; ++i) {
    a.t_model[i] = (int16_t)lp.t_model[i];
    a.t_post[i] = (int16_t)lp.t_post[i];
  }
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, st, a);
}

void stack_phase_read(unsigned long long* out16) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stack_phase), 16 * sizeof(unsigned long long));
  unsigned long long z[16] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stack_phase), z, sizeof(z));
}

}  // namespace ldm

// cond=relation: the SGD of the reference's logit adjustment on ONE layout whose log-probabilities live in LDS — the single
// source of this arithmetic for every kernel that runs it:
//
//   relation_update_k   (kernels_relation.hip)  the split-step hook ldm_relation_update: (B,C,S) tensor in / out
//   relation_step_k     (kernels_relation.hip)  the per-step path's fused tail: posterior -> SGD -> [PAD] disable -> draw in
//                                               one launch per chunk-step (exact / split numerics, other geometries)
//   stack_stream_k<., 2, true> (kernels_stack.hip)  the one-launch reverse loop: the same three stages behind the vocabulary
//                                               head, on the logits the workgroup already holds in LDS
//
// Reference: update() in trainer/models/categorical_diffusion/logit_adjustment.py:88-126 runs `relation_num_update`
// plain-SGD steps (lr = relation_lambda, t >= 10 only) on  mean_{graph, f} cost_f  w.r.t. the (B,C,S) log-probability
// tensor, where the 14 costs f (trainer/models/clg/const.py:221-236) are hinge losses on the EXPECTED boxes
//     bbox[node, x] = sum_n softmax_n(logp[node, bins of x]) * centre_x[n]        (_stochastic_convert, l.16-85,
//                                                                                  mode = "average")
// of the canvas node (fixed) and of every element whose conditioned category is not PAD.  Autograd there; here
//     d mean / d logit[node,n,x] = 1/(14 B) * p_n (c_n - bbox_x) * G[node,x],   G = sum over the node's edges of the
// hinge sub-gradients (relu'(z) = [z > 0]) of the area / centre-y / left-top-right-bottom terms.
// Only the body bins of the bbox positions of graph nodes ever change (the softmax runs over the N bins of the
// coordinate's slice: logit_adjustment.py:54-66); [PAD] / [MASK] and the category positions are untouched.
//
// Work split (256 threads): one 16-lane DPP row per (element, coordinate) softmax, two bins per lane (no LDS round trip, no
// ds_bpermute: ldm_post_dpp.h); one thread per edge for the hinge sub-gradients; node gradients summed in EDGE ORDER
// (deterministic); the SGD step on all E x 4 x N logits.  Results agree with the autograd reference to fp32 rounding.
#pragma once
#include "ldm_kernels.h"
#include "ldm_post_dpp.h"

namespace ldm {

constexpr int REL_MAX_ELEM = 32;
constexpr int REL_MAX_EDGE = 512;

// LDS scratch of one layout (floats): bbox | grad | eg | node_of | the layout's edges (src | dst | attr), staged ONCE — the
// r03 kernel re-read them from global memory in every iteration, inside the serial per-node sums: ~60 of its 100 us
constexpr int kRelBboxOff = 0;
constexpr int kRelGradOff = kRelBboxOff + (REL_MAX_ELEM + 1) * 4;
constexpr int kRelEgOff = kRelGradOff + (REL_MAX_ELEM + 1) * 4;
constexpr int kRelNodeOff = kRelEgOff + REL_MAX_EDGE * 8;
constexpr int kRelEdgeOff = kRelNodeOff + REL_MAX_ELEM;
constexpr int kRelCentreOff = kRelEdgeOff + 3 * REL_MAX_EDGE;   // the cluster centres, [4][32] (r04: were re-read from global
                                                                // memory by every softmax round: ~11 of the SGD's 44 us)
constexpr int kRelDummyOff = kRelCentreOff + 4 * 32;            // 32 logits | 32 softmax values | 1 box entry that nobody reads back
constexpr int kRelScratchFloats = kRelDummyOff + 68;
// incidence lists of the layout's graph nodes (which edges touch a node, in EDGE ORDER: the order the node gradients are
// summed in), built once per launch: offsets [REL_MAX_ELEM + 2] ints, entries [2 REL_MAX_EDGE] = edge << 1 | (node is dst).
// r03 scanned all edges per node, per iteration (a chain of dependent LDS reads: ~12 us per step).
constexpr int kRelIncOffInts = REL_MAX_ELEM + 2;
constexpr int kRelIncBytes = kRelIncOffInts * 4 + 2 * REL_MAX_EDGE * 2;

// the graph and the hyper-parameters of one call (device pointers; edge offsets are absolute positions in the edge arrays)
struct RelGraph {
  const int32_t* edge_off;                         // + global layout index
  const int32_t *edge_src, *edge_dst, *edge_attr;
  const float* centres;                            // (4, n_bin)
  int canvas_bins[4];
  float step;                                      // relation_lambda / (14 * number of graphs of the call)
  int num_update;
};

// element -> node index (1..; 0 = canvas), -1 = not in the graph: tid 0 fills it; canvas box by tid < 4.  cond_tok(e) =
// conditioned category token of element e.  The caller synchronises afterwards.
template <class CondTok>
__device__ __forceinline__ void relation_nodes(float* scratch, int tid, int E, int pad_id, int n_bin, const float* centres,
                                               const int* canvas_bins, CondTok cond_tok) {
  int* node_of = reinterpret_cast<int*>(scratch + kRelNodeOff);
  if (tid == 0) {
    int k = 1;
    for (int e = 0; e < E; ++e) node_of[e] = (cond_tok(e) != pad_id) ? k++ : -1;
  }
  if (tid < 4) scratch[kRelBboxOff + tid] = centres[tid * n_bin + canvas_bins[tid]];  // canvas: one-hot expectation
}

// Fills inc_off / inc for the layout's graph (inc_off[0] = -1: more than REL_MAX_EDGE edges, relation_sgd then scans).
// Uses the edge and gradient areas of `scratch`; ends with a barrier.
template <class Graph, class Barrier>
__device__ __forceinline__ void relation_incidence(const Graph& a, int e0, int ne, int tid, int E, float* scratch, int* inc_off,
                                                   unsigned short* inc, Barrier barrier) {
  if (ne > REL_MAX_EDGE) {
    if (tid == 0) inc_off[0] = -1;
    barrier();
    return;
  }
  int* es = reinterpret_cast<int*>(scratch + kRelEdgeOff);
  int* ed = es + REL_MAX_EDGE;
  int* deg = reinterpret_cast<int*>(scratch + kRelGradOff);
  for (int k = tid; k < ne; k += 256) {
    es[k] = a.edge_src[e0 + k];
    ed[k] = a.edge_dst[e0 + k];
  }
  barrier();
  if (tid <= E) {
    int c = 0;
    for (int k = 0; k < ne; ++k) c += (es[k] == tid) + (ed[k] == tid);
    deg[tid] = c;
  }
  barrier();
  if (tid == 0) {
    int o = 0;
    for (int n = 0; n <= E; ++n) {
      inc_off[n] = o;
      o += deg[n];
    }
    inc_off[E + 1] = o;
  }
  barrier();
  if (tid <= E) {
    int o = inc_off[tid];
    for (int k = 0; k < ne; ++k) {
      if (es[k] == tid) inc[o++] = (unsigned short)(k << 1);
      if (ed[k] == tid) inc[o++] = (unsigned short)((k << 1) | 1);
    }
  }
  barrier();
}

// What a kernel that keeps a layout for many steps (the loop kernel) stages ONCE per launch instead of once per step: the
// layout's edges, packed src | dst << 6 | attr << 12 (<= REL_MAX_EDGE of them), and the cluster centres [4][32].  A per-step
// kernel passes {nullptr, nullptr}: relation_sgd then stages both from global memory itself.
struct RelPersist {
  const unsigned* edges;
  const float* centres;
};
constexpr int kRelPassBatch = 8;  // (element, coordinate) rows a 16-lane group has in flight at once

// lg(e, x) -> the n_bin body-bin logits of element e, coordinate x (LDS, updated in place; 32 readable floats);
// pr(e, x) -> n_bin floats of LDS scratch for their softmax (32 readable floats);  barrier() -> workgroup barrier.
// Graph: RelGraph, possibly in the kernel-argument address space (the loop kernel reads it there at the point of use).
template <bool PACKED, class Graph, class LgAt, class PrAt, class Barrier>
__device__ __forceinline__ void relation_sgd(const Graph& a, int e0, int ne, int tid, int E, int NB, LgAt lg, PrAt pr,
                                             float* scratch, const int* inc_off, const unsigned short* inc, RelPersist pers,
                                             Barrier barrier) {
  float* bbox = scratch + kRelBboxOff;
  float* grad = scratch + kRelGradOff;
  float* eg = scratch + kRelEgOff;
  const int* node_of = reinterpret_cast<const int*>(scratch + kRelNodeOff);
  int* es = reinterpret_cast<int*>(scratch + kRelEdgeOff);
  int* ed = es + REL_MAX_EDGE;
  int* ea = ed + REL_MAX_EDGE;
  auto stage_edges = [&](int eb, int nb) {
    for (int k = tid; k < nb; k += 256) {
      es[k] = a.edge_src[e0 + eb + k];
      ed[k] = a.edge_dst[e0 + eb + k];
      ea[k] = a.edge_attr[e0 + eb + k];
    }
  };
  const bool one_block = ne <= REL_MAX_EDGE;
  const float* cen = pers.centres;
  if constexpr (!PACKED) {
    if (one_block) stage_edges(0, ne);  // (visible behind the first barrier below)
    float* cs = scratch + kRelCentreOff;
    if (tid < 4 * 32) cs[tid] = (tid & 31) < NB ? a.centres[(tid >> 5) * NB + (tid & 31)] : 0.f;
    cen = cs;
    barrier();
  }
  const bool by_node = one_block && inc_off[0] >= 0;
  const int grp = tid >> 4, l16 = tid & 15;
  const ldm_post::DppGroup<16, false> g{l16};
  const bool ok0 = l16 < NB, ok1 = l16 + 16 < NB;
  // The rows of group grp: (element (grp >> 2) + 4 r, coordinate grp & 3), r = 0 .. — the coordinate, hence the centres, is the
  // same for all of them.  A row that is not a graph node's (or past E) works on a dummy row of the scratch: no branch
  // inside a batch, so the LDS reads, the three 16-lane reductions and the exp / divide chains of kRelPassBatch rows
  // interleave (r04: one row at a time left ~8 000 cycles of a pass to LDS and DPP latency).
  const int x = grp & 3;
  const float c0 = cen[x * 32 + l16], c1 = cen[x * 32 + l16 + 16];
  float* const dummy = scratch + kRelDummyOff;
  constexpr int NR = REL_MAX_ELEM / 4;
  // One pass per iteration over the (element, coordinate) pairs, a 16-lane row each (two bins per lane): the SGD step of
  // the PREVIOUS iteration on the pair's logits (its softmax, its expected coordinate and the node gradient are still in
  // LDS), then the softmax / expectation of the updated logits.  A row reads and writes only its own pair's entries, so the
  // update needs no barrier of its own (3 barriers per iteration, no integer divisions).
  for (int it = 0; it <= a.num_update; ++it) {
    const bool upd = it > 0, soft = it < a.num_update;
#pragma unroll 1
    for (int rb = 0; rb < NR && (rb * 4 + (grp >> 2)) < E; rb += kRelPassBatch) {
      float *L[kRelPassBatch], *P[kRelPassBatch], *BB[kRelPassBatch];
      const float* GR[kRelPassBatch];
      float v0[kRelPassBatch], v1[kRelPassBatch];
#pragma unroll
      for (int r = 0; r < kRelPassBatch; ++r) {
        const int e = (grp >> 2) + 4 * (rb + r);
        const int node = e < E ? node_of[e] : -1;
        const bool valid = node >= 0;
        L[r] = valid ? lg(e, x) : dummy;
        P[r] = valid ? pr(e, x) : dummy + 32;
        BB[r] = valid ? bbox + node * 4 + x : dummy + 64;
        GR[r] = valid ? grad + node * 4 + x : dummy + 64;
      }
#pragma unroll
      for (int r = 0; r < kRelPassBatch; ++r) {
        const float a0 = L[r][l16], a1 = L[r][l16 + 16];
        v0[r] = ok0 ? a0 : -INFINITY;
        v1[r] = ok1 ? a1 : -INFINITY;
      }
      if (upd) {  // ---- SGD step through the softmax expectation (iteration it - 1)
        float q0[kRelPassBatch], q1[kRelPassBatch], bo[kRelPassBatch], go[kRelPassBatch];
#pragma unroll
        for (int r = 0; r < kRelPassBatch; ++r) {
          q0[r] = P[r][l16];
          q1[r] = P[r][l16 + 16];
          bo[r] = *BB[r];
          go[r] = *GR[r];
        }
#pragma unroll
        for (int r = 0; r < kRelPassBatch; ++r) {
          if (ok0) v0[r] -= a.step * (q0[r] * (c0 - bo[r]) * go[r]);
          if (ok1) v1[r] -= a.step * (q1[r] * (c1 - bo[r]) * go[r]);
        }
        if (ok0) {
#pragma unroll
          for (int r = 0; r < kRelPassBatch; ++r) L[r][l16] = v0[r];
        }
        if (ok1) {
#pragma unroll
          for (int r = 0; r < kRelPassBatch; ++r) L[r][l16 + 16] = v1[r];
        }
      }
      if (!soft) continue;
      float p0[kRelPassBatch], p1[kRelPassBatch], bb[kRelPassBatch];
#pragma unroll
      for (int r = 0; r < kRelPassBatch; ++r) {
        const float mx = g.gmax(fmaxf(v0[r], v1[r]));
        const float x0 = ok0 ? expf(v0[r] - mx) : 0.f, x1 = ok1 ? expf(v1[r] - mx) : 0.f;
        const float sm = g.gsum(x0 + x1);
        p0[r] = x0 / sm;
        p1[r] = x1 / sm;
        bb[r] = g.gsum(p0[r] * c0 + p1[r] * c1);
      }
      if (ok0) {
#pragma unroll
        for (int r = 0; r < kRelPassBatch; ++r) P[r][l16] = p0[r];
      }
      if (ok1) {
#pragma unroll
        for (int r = 0; r < kRelPassBatch; ++r) P[r][l16 + 16] = p1[r];
      }
      if (l16 == 0) {
#pragma unroll
        for (int r = 0; r < kRelPassBatch; ++r) *BB[r] = bb[r];
      }
    }
    if (it == a.num_update) break;
    barrier();
    // ---- per-edge hinge sub-gradients (clg/const.py), REL_MAX_EDGE edges at a time
    if (tid < (E + 1) * 4) grad[tid] = 0.f;
    for (int eb = 0; eb < ne; eb += REL_MAX_EDGE) {
      const int nb = min(REL_MAX_EDGE, ne - eb);
      if (!PACKED && !one_block) {
        if (eb > 0) barrier();  // (the previous block's node sums have read es / ed / eg)
        stage_edges(eb, nb);
        barrier();
      }
      for (int k = tid; k < nb; k += 256) {
        int s, d, at;
        if constexpr (PACKED) {
          const unsigned pk = pers.edges[k];
          s = (int)(pk & 63u); d = (int)((pk >> 6) & 63u); at = (int)(pk >> 12);
        } else {
          s = es[k]; d = ed[k]; at = ea[k];
        }
        const float xs = bbox[s * 4], ys = bbox[s * 4 + 1], ws = bbox[s * 4 + 2], hs = bbox[s * 4 + 3];
        const float xd = bbox[d * 4], yd = bbox[d * 4 + 1], wd = bbox[d * 4 + 2], hd = bbox[d * 4 + 3];
        const float eps = 1e-8f;
        float gs[4] = {0.f, 0.f, 0.f, 0.f}, gd[4] = {0.f, 0.f, 0.f, 0.f};
        {  // relative size (const.py:56-106): a = w*h ; both canvas variants share the formula
          const float as = ws * hs, ad = wd * hd;
          const float sml = 0.9f * as, lgv = 1.1f * as;  // (1 -/+ REL_SIZE_ALPHA) * a1
          float gas = 0.f, gad = 0.f;
          if (at & (1 << 1)) { if (ad - sml > 0.f) { gad += 1.f; gas -= 0.9f; } }
          if (at & (1 << 2)) {
            if ((sml - ad) + eps > 0.f) { gas += 0.9f; gad -= 1.f; }
            if ((ad - lgv) + eps > 0.f) { gad += 1.f; gas -= 1.1f; }
          }
          if (at & (1 << 3)) { if (lgv - ad > 0.f) { gas += 1.1f; gad -= 1.f; } }
          gs[2] += gas * hs; gs[3] += gas * ws;
          gd[2] += gad * hd; gd[3] += gad * wd;
        }
        if (s == 0) {  // location w.r.t. the canvas (const.py:109-157): centre-y thirds of the dst element
          const float y_sm = (float)(1.0 / 3), y_lg = (float)(2.0 / 3);
          if (at & (1 << 6)) { if (yd - y_sm > 0.f) gd[1] += 1.f; }
          if (at & (1 << 9)) {
            if ((y_sm - yd) + eps > 0.f) gd[1] -= 1.f;
            if ((yd - y_lg) + eps > 0.f) gd[1] += 1.f;
          }
          if (at & (1 << 8)) { if (y_lg - yd > 0.f) gd[1] -= 1.f; }
        } else {  // pairwise location (const.py:160-218) on l,t,r,b = xc -/+ w/2, yc -/+ h/2
          const float l1 = xs - ws / 2, t1 = ys - hs / 2, r1 = xs + ws / 2, b1 = ys + hs / 2;
          const float l2 = xd - wd / 2, t2 = yd - hd / 2, r2 = xd + wd / 2, b2 = yd + hd / 2;
          float gl1 = 0.f, gt1 = 0.f, gr1 = 0.f, gb1 = 0.f, gl2 = 0.f, gt2 = 0.f, gr2 = 0.f, gb2 = 0.f;
          if (at & (1 << 6)) { if (b2 - t1 > 0.f) { gb2 += 1.f; gt1 -= 1.f; } }
          if (at & (1 << 8)) { if (b1 - t2 > 0.f) { gb1 += 1.f; gt2 -= 1.f; } }
          if (at & (1 << 5)) { if (r2 - l1 > 0.f) { gr2 += 1.f; gl1 -= 1.f; } }
          if (at & (1 << 7)) { if (r1 - l2 > 0.f) { gr1 += 1.f; gl2 -= 1.f; } }
          if (at & (1 << 9)) {
            if ((l1 - r2) + eps > 0.f) { gl1 += 1.f; gr2 -= 1.f; }
            if ((l2 - r1) + eps > 0.f) { gl2 += 1.f; gr1 -= 1.f; }
          }
          const float nx = (float)(((at >> 5) & 1) + ((at >> 7) & 1) + ((at >> 9) & 1));  // LEFT / RIGHT / CENTER add t1<b2, t2<b1
          if (nx > 0.f) {
            if ((t1 - b2) + eps > 0.f) { gt1 += nx; gb2 -= nx; }
            if ((t2 - b1) + eps > 0.f) { gt2 += nx; gb1 -= nx; }
          }
          gs[0] += gl1 + gr1; gs[2] += (gr1 - gl1) * 0.5f; gs[1] += gt1 + gb1; gs[3] += (gb1 - gt1) * 0.5f;
          gd[0] += gl2 + gr2; gd[2] += (gr2 - gl2) * 0.5f; gd[1] += gt2 + gb2; gd[3] += (gb2 - gt2) * 0.5f;
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          eg[k * 8 + x] = gs[x];
          eg[k * 8 + 4 + x] = gd[x];
        }
      }
      barrier();
      // ---- node gradients: deterministic sum in edge order
      if (tid < (E + 1) * 4) {
        const int node = tid >> 2, x = tid & 3;
        float gsum = grad[tid];
        if (by_node) {
#pragma unroll 4
          for (int j = inc_off[node]; j < inc_off[node + 1]; ++j) {
            const int ent = inc[j];
            gsum += eg[(ent >> 1) * 8 + (ent & 1) * 4 + x];
          }
        } else {
          for (int k = 0; k < nb; ++k) {
            if (es[k] == node) gsum += eg[k * 8 + x];
            if (ed[k] == node) gsum += eg[k * 8 + 4 + x];
          }
        }
        grad[tid] = gsum;
      }
    }  // edge blocks
    barrier();
  }
  barrier();  // (the last SGD step's writes: the caller reads the rows next)
}

}  // namespace ldm

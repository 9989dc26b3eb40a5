// cond=relation: the logit adjustment of the reference's sampler as ONE kernel with the analytic gradient.
//
// Reference: update() in trainer/models/categorical_diffusion/logit_adjustment.py:88-126 runs `relation_num_update`
// plain-SGD steps (lr = relation_lambda, t >= 10 only) on  mean_{graph, f} cost_f  w.r.t. the (B,C,S) log-probability
// tensor, where the 14 costs f (trainer/models/clg/const.py:221-236) are hinge losses on the EXPECTED boxes
//     bbox[node, x] = sum_n softmax_n(logp[node, bins of x]) * centre_x[n]        (_stochastic_convert, l.16-85,
//                                                                                  mode = "average")
// of the canvas node (fixed) and of every element whose conditioned category is not PAD.  Autograd there; here
//     d mean / d logit[node,n,x] = 1/(14 B) * p_n (c_n - bbox_x) * G[node,x],   G = sum over the node's edges of the
// hinge sub-gradients (relu'(z) = [z > 0]) of the area / centre-y / left-top-right-bottom terms.
// One workgroup per layout: 25 elements x 4 coordinates x 32 bins of logits live in LDS for all iterations.
// Edge sums run in edge order (deterministic); results agree with the autograd reference to fp32 rounding.
#include "ldm_kernels.h"

namespace ldm {

constexpr int REL_MAX_ELEM = 32;
constexpr int REL_MAX_EDGE = 512;

__global__ __launch_bounds__(256) void relation_update_k(RelArgs a) {
  __shared__ float lg[REL_MAX_ELEM * 4 * 32];   // logits of the bbox sub-vocabularies
  __shared__ float pr[REL_MAX_ELEM * 4 * 32];   // their softmax
  __shared__ float bbox[(REL_MAX_ELEM + 1) * 4];
  __shared__ float grad[(REL_MAX_ELEM + 1) * 4];
  __shared__ float eg[REL_MAX_EDGE * 8];        // per-edge gradient wrt (x,y,w,h) of src and dst
  __shared__ int node_of[REL_MAX_ELEM];         // element -> node index (1..), -1 = not in the graph
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int E = a.S / a.A, NB = a.n_bin;
  const int e0 = a.edge_off[b], ne = a.edge_off[b + 1] - e0;
  if (tid == 0) {
    int k = 1;  // node 0 = canvas
    for (int e = 0; e < E; ++e) node_of[e] = (a.cond_seq[(size_t)b * a.S + e * a.A] != a.pad_id) ? k++ : -1;
  }
  __syncthreads();
  const int n_item = E * 4 * NB;
  // the lanes run along the contiguous axis of the log-probabilities: the ELEMENT index for the API's (B, C, S) layout
  // (25 reads inside one 500-byte class row instead of 64 rows per wave instruction), the bin index for (B, S, C)
  auto at = [&](int e, int x, int n) -> size_t {
    const size_t cls = (size_t)(a.n_category + x * NB + n), pos = (size_t)(e * a.A + 1 + x);
    return a.logp_tm ? ((size_t)b * a.S + pos) * a.C + cls : ((size_t)b * a.C + cls) * a.S + pos;
  };
  for (int i = tid; i < n_item; i += 256) {
    int e, x, n;
    if (a.logp_tm) { e = i / (4 * NB); x = (i / NB) % 4; n = i % NB; } else { e = i % E; x = (i / E) / NB; n = (i / E) % NB; }
    lg[(e * 4 + x) * NB + n] = node_of[e] > 0 ? a.logp[at(e, x, n)] : 0.f;
  }
  if (tid < 4) bbox[tid] = a.centres[tid * NB + a.canvas_bins[tid]];  // canvas: one-hot expectation
  __syncthreads();
  const int half = tid >> 5, ln = tid & 31;  // 8 groups of 32 lanes: one (element, coordinate) softmax each
  for (int it = 0; it < a.num_update; ++it) {
    for (int pidx = half; pidx < E * 4; pidx += 8) {
      const int e = pidx >> 2, x = pidx & 3;
      if (node_of[e] < 0) continue;  // (uniform per 32-lane group)
      const float v = ln < NB ? lg[pidx * NB + ln] : -INFINITY;
      float mx = v;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 32));
      const float ex = ln < NB ? expf(v - mx) : 0.f;
      float sm = ex;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 32);
      const float p = ex / sm;
      if (ln < NB) pr[pidx * NB + ln] = p;
      float bb = ln < NB ? p * a.centres[x * NB + ln] : 0.f;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) bb += __shfl_xor(bb, o, 32);
      if (ln == 0) bbox[node_of[e] * 4 + x] = bb;
    }
    __syncthreads();
    // ---- per-edge hinge sub-gradients (clg/const.py), REL_MAX_EDGE edges at a time
    if (tid < (E + 1) * 4) grad[tid] = 0.f;
    for (int eb = 0; eb < ne; eb += REL_MAX_EDGE) {
    const int nb = min(REL_MAX_EDGE, ne - eb);
    for (int k = tid; k < nb; k += 256) {
      const int s = a.edge_src[e0 + eb + k], d = a.edge_dst[e0 + eb + k], at = a.edge_attr[e0 + eb + k];
      const float xs = bbox[s * 4], ys = bbox[s * 4 + 1], ws = bbox[s * 4 + 2], hs = bbox[s * 4 + 3];
      const float xd = bbox[d * 4], yd = bbox[d * 4 + 1], wd = bbox[d * 4 + 2], hd = bbox[d * 4 + 3];
      const float eps = 1e-8f;
      float gs[4] = {0.f, 0.f, 0.f, 0.f}, gd[4] = {0.f, 0.f, 0.f, 0.f};
      {  // relative size (const.py:56-106): a = w*h ; both canvas variants share the formula
        const float as = ws * hs, ad = wd * hd;
        const float sm = 0.9f * as, lgv = 1.1f * as;  // (1 -/+ REL_SIZE_ALPHA) * a1
        float gas = 0.f, gad = 0.f;
        if (at & (1 << 1)) { if (ad - sm > 0.f) { gad += 1.f; gas -= 0.9f; } }
        if (at & (1 << 2)) {
          if ((sm - ad) + eps > 0.f) { gas += 0.9f; gad -= 1.f; }
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
    __syncthreads();
    // ---- node gradients: deterministic sum in edge order
    if (tid < (E + 1) * 4) {
      const int node = tid >> 2, x = tid & 3;
      float g = grad[tid];
      for (int k = 0; k < nb; ++k) {
        if (a.edge_src[e0 + eb + k] == node) g += eg[k * 8 + x];
        if (a.edge_dst[e0 + eb + k] == node) g += eg[k * 8 + 4 + x];
      }
      grad[tid] = g;
    }
    __syncthreads();
    }  // edge blocks
    __syncthreads();
    // ---- SGD step through the softmax expectation
    for (int i = tid; i < n_item; i += 256) {
      const int e = i / (4 * NB), x = (i / NB) % 4, n = i % NB;
      const int node = node_of[e];
      if (node < 0) continue;
      lg[i] -= a.step * (pr[i] * (a.centres[x * NB + n] - bbox[node * 4 + x]) * grad[node * 4 + x]);
    }
    __syncthreads();
  }
  for (int i = tid; i < n_item; i += 256) {
    int e, x, n;
    if (a.logp_tm) { e = i / (4 * NB); x = (i / NB) % 4; n = i % NB; } else { e = i % E; x = (i / E) / NB; n = (i / E) % NB; }
    if (node_of[e] > 0) a.logp[at(e, x, n)] = lg[(e * 4 + x) * NB + n];
  }
}

void launch_relation_update(const RelArgs& a, hipStream_t st) {
  if (a.B <= 0 || a.num_update <= 0) return;
  hipLaunchKernelGGL(relation_update_k, dim3(a.B), dim3(256), 0, st, a);
}

}  // namespace ldm

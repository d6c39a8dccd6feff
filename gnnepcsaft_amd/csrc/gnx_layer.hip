// Native launch sequence of one PNAConv backward (gnx_pna_conv_bwd): the ~35 launches that
// gnnepcsaft_amd/functional.py::PNAConvFn.backward issues one by one from Python (0.32 ms of host time per layer, 5.5 ms
// per cfg-2 step, against 7.9 ms of GPU time) are issued here from ONE call -- same kernels, same streams, same order.
// It only composes the public entry points of gnx.h; nothing below touches a kernel directly.
//
// Reference semantics: the backward of [3P] torch_geometric.nn.PNAConv as built at
// /root/reference/gnnepcsaft/train/models.py:445-457 (restructured as DESIGN.md §4 describes: node-level P/Q products +
// 60-row bond table for pre-layer 0, per-degree-class effective weights for post-layer 0, lin o last post layer merged).
#include "gnx_common.hpp"

#include <algorithm>
#include <vector>

namespace {

inline gnx_gemm_seg seg(const float* a, int64_t lda, const float* b, int64_t ldb, int32_t k) {
  gnx_gemm_seg s;
  s.a = a;
  s.lda = lda;
  s.rowscale = nullptr;
  s.b = b;
  s.ldb = ldb;
  s.k = k;
  s._pad = 0;
  return s;
}

inline bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

struct WgradQueue {
  std::vector<gnx_wgrad_prob> q;
  void add(const float* dC, int64_t lddc, const float* A, int64_t lda, int64_t M, int32_t N, int32_t K, float* dW,
           int64_t lddw, float* dbias) {
    if (M <= 0) return;
    gnx_wgrad_prob p;
    p.dC = dC;
    p.lddc = lddc;
    p.A = A;
    p.lda = lda;
    p.rowscale = nullptr;
    p.dW = dW;
    p.lddw = lddw;
    p.dbias = dbias;
    p.M = M;
    p.N = N;
    p.K = K;
    q.push_back(p);
  }
  // up to 8 problems per launch, largest row counts first (problems of similar size share a launch)
  int32_t flush(gnx_handle* h) {
    std::stable_sort(q.begin(), q.end(), [](const gnx_wgrad_prob& a, const gnx_wgrad_prob& b) { return a.M > b.M; });
    for (size_t i = 0; i < q.size(); i += 8) {
      const int32_t n = (int32_t)std::min<size_t>(8, q.size() - i);
      const int32_t st = gnx_gemm_wgrad_batched(h, n, q.data() + i);
      if (st != GNX_OK) return st;
    }
    q.clear();
    return GNX_OK;
  }
};

#define GNX_TRY(call)                 \
  do {                                \
    const int32_t st_ = (call);       \
    if (st_ != GNX_OK) return st_;    \
  } while (0)

// runs `body` on side stream `which` (forked from the current point of the bound stream), restoring the stream on any exit
template <class F>
int32_t on_side(gnx_handle* h, int which, bool enabled, F body) {
  if (!enabled) return body();
  GNX_TRY(gnx_side_begin_n(h, which));
  const int32_t st = body();
  (void)gnx_side_end(h);
  return st;
}

}  // namespace

// upper bound of the split-weight images any product of the layer's forward or backward needs (gnx_gemm_workspace_bytes): the
// degree-class product dA = g Weff(d) (D classes, N = 4F, K = F) and the 3-segment dx product (N = F, K = 3F)
// Layout of the workspace: [0, scratch) is reused by every product of the layer in turn; behind it one region per tower and
// per tiled product whose weight images are split AHEAD on side stream 2 (GNX_OPT_SPLIT_AHEAD): forward: post-layer 0;
// backward: dA and dx.
static size_t align256(size_t v) { return (v + 255) / 256 * 256; }
static void pna_ws_sizes(int32_t T, int32_t F, int32_t D, size_t& scratch, size_t& post0, size_t& grouped, size_t& dx) {
  auto pad = [](int64_t v, int64_t m) { return (v + m - 1) / m * m; };
  grouped = align256((size_t)(D > 0 ? D : 1) * 3 * pad(4 * (int64_t)F, 128) * pad(F, 32) * 2);
  dx = align256((size_t)3 * pad(F, 128) * (3 * pad(F, 32)) * 2);
  const size_t lin = (size_t)3 * pad((int64_t)T * F, 128) * pad((int64_t)T * F, 32) * 2;  // lin / merged product, H x H
  // forward: post-layer 0 by degree class, z = x W0^T + A Weff(d)^T  (N = F, K = F + 4F)
  post0 = align256((size_t)(D > 0 ? D : 1) * 3 * pad(F, 128) * (pad(F, 32) + pad(4 * (int64_t)F, 32)) * 2);
  scratch = align256(std::max(std::max(grouped, post0), std::max(dx, lin)) + 256);
}

extern "C" size_t gnx_pna_conv_bwd_workspace_bytes(int32_t T, int32_t F, int32_t D) {
  size_t scratch, post0, grouped, dx;
  pna_ws_sizes(T, F, D, scratch, post0, grouped, dx);
  return scratch + (size_t)(T > 0 ? T : 1) * std::max(post0, grouped + dx);
}

extern "C" int32_t gnx_pna_conv_bwd(gnx_handle* h, const gnx_pna_bwd_args* a) {
  GNX_CHECK_ARG(h && a, "gnx_pna_conv_bwd: NULL argument");
  const int T = a->T, F = a->F, pre = a->pre_layers, post = a->post_layers, R = a->R, D = a->D;
  const int64_t N = a->N, E = a->E;
  const int H = T * F;
  GNX_CHECK_ARG(T >= 1 && F >= 1 && pre >= 1 && pre <= GNX_PNA_MAX_LAYERS && post >= 1 && post <= GNX_PNA_MAX_LAYERS &&
                    T <= GNX_PNA_MAX_TOWERS && D >= 1 && R >= 1,
                "gnx_pna_conv_bwd: bad layer shape T=%d F=%d pre=%d post=%d D=%d R=%d", T, F, pre, post, D, R);
  GNX_CHECK_ARG(a->params && a->grads && a->dout && a->x && a->A && a->dx && a->acc_buf, "gnx_pna_conv_bwd: NULL array");
  const bool side = a->use_side_streams != 0;
  const int per = 2 * (pre + post);
  auto pidx = [&](int t, bool is_pre, int i) { return 4 + t * per + 2 * (is_pre ? i : pre + i); };
  const float* const* P = a->params;
  float* const* G = a->grads;
  const float* lin_w = P[2];
  WgradQueue wq;

  // the weight images of the two tiled products of this backward (dA per degree class, the 3-segment dx) are split on side
  // stream 2 while the lin / hidden input-gradient products run; the products take them as they are (GNX_GEMM_PRESPLIT)
  size_t scratch = a->ws_bytes, r_post0 = 0, r_dA = 0, r_dx = 0;
  bool ahead = (h->opt[GNX_OPT_SPLIT_AHEAD] == 2 || (h->opt[GNX_OPT_SPLIT_AHEAD] == 1 && T >= 2)) && a->use_side_streams != 0 &&
               !h->on_side && N >= 4096 && a->ws != nullptr;
  if (ahead) {
    pna_ws_sizes(T, F, D, scratch, r_post0, r_dA, r_dx);
    ahead = scratch + (size_t)T * (r_dA + r_dx) <= a->ws_bytes;
    if (!ahead) scratch = a->ws_bytes;
  }
  unsigned char* const ahead_base = reinterpret_cast<unsigned char*>(a->ws) + scratch;
  auto dA_call = [&](int t, const float* gt, int extra_flags, void* ws, size_t ws_bytes) -> int32_t {
    gnx_gemm_seg s = seg(gt, H, a->weff[t], 4 * F, F);
    const int64_t stride = (int64_t)4 * F * F;
    return gnx_gemm_grouped(h, 1, &s, &stride, D, N, 4 * F, nullptr, nullptr, 0, a->dA + (int64_t)t * 4 * F, (int64_t)T * 4 * F,
                            extra_flags, a->dperm, a->tiles, a->ntiles, a->max_tiles, ws, ws_bytes);
  };
  auto dx_call = [&](int t, const float* g_post0, int extra_flags, void* ws, size_t ws_bytes) -> int32_t {
    const int k0 = pidx(t, true, 0), kp = pidx(t, false, 0);
    const float* W0 = P[k0];
    gnx_gemm_seg s3[3] = {seg(g_post0 + t * F, H, P[kp], 13 * F, F), seg(a->dP + t * F, H, W0, 3 * F, F),
                          seg(a->dQ + t * F, H, W0 + F, 3 * F, F)};
    return gnx_gemm(h, 3, s3, N, F, nullptr, nullptr, 0, a->dx + t * F, H, extra_flags, ws, ws_bytes);
  };
  if (ahead) {
    GNX_TRY(gnx_side_begin_n(h, 2));
    int32_t st = GNX_OK;
    for (int t = 0; t < T && st == GNX_OK; ++t) {  // (gbuf[0] stands in for the gradient operand: only its alignment matters)
      unsigned char* reg = ahead_base + (size_t)t * (r_dA + r_dx);
      st = dA_call(t, a->gbuf[0] + t * F, GNX_GEMM_SPLIT_ONLY, reg, r_dA);
      if (st == GNX_OK) st = dx_call(t, a->gbuf[0], GNX_GEMM_SPLIT_ONLY, reg + r_dA, r_dx);
    }
    (void)gnx_side_end(h);
    GNX_TRY(st);
  }

  // ---- lin (or lin o last post layer): dgrad into gbuf[0]
  // every gradient buffer of the chain is distinct: the queued weight-gradient problems read them at flush time
  int gi = 0;
  float* g = a->gbuf[gi];
  const float* z_last = a->zs[a->n_z - 1];
  int last_hidden;
  const bool tail = (a->defer_small & 2) != 0;   // nothing follows this layer on the main stream (see the flush below)
  const bool class_after_agg = (a->defer_small & 8) != 0 && side;  // per-class weight gradient forked behind the aggregate backward
  std::vector<std::function<int32_t()>> class_wgrads;
  const bool defer = (a->defer_small & 1) != 0;  // small accumulators pre-zeroed by the caller, their consumers run in
                                           // gnx_pna_stack_finish for all layers at once
  if (a->merged) {
    if (!defer) {
      GNX_TRY(gnx_fill(h, a->dWm, (int64_t)H * H, 0.f));
      GNX_TRY(gnx_fill(h, a->dbm, H, 0.f));
    }
    wq.add(a->dout, H, z_last, H, N, H, H, a->dWm, H, a->dbm);
    gnx_gemm_seg s = seg(a->dout, H, a->Wm, H, H);
    GNX_TRY(gnx_gemm(h, 1, &s, N, H, nullptr, z_last, H, g, H, 0, a->ws, a->ws_bytes));
    last_hidden = post - 2;
  } else {
    wq.add(a->dout, H, z_last, H, N, H, H, G[2], H, G[3]);
    gnx_gemm_seg s = seg(a->dout, H, lin_w, H, H);
    GNX_TRY(gnx_gemm(h, 1, &s, N, H, nullptr, nullptr, 0, g, H, 0, a->ws, a->ws_bytes));
    last_hidden = post - 1;
  }
  // ---- hidden post layers last..1: dgrad masked by the relu'd input activation
  for (int i = last_hidden; i >= 1; --i) {
    const float* a_prev = a->zs[i - 1];
    for (int t = 0; t < T; ++t) {
      const int k = pidx(t, false, i);
      wq.add(g + t * F, H, a_prev + t * F, H, N, F, F, G[k], F, G[k + 1]);
      gnx_gemm_seg s = seg(g + t * F, H, P[k], F, F);
      GNX_TRY(gnx_gemm(h, 1, &s, N, F, nullptr, a_prev + t * F, H, a->gbuf[gi + 1] + t * F, H, 0, a->ws, a->ws_bytes));
    }
    g = a->gbuf[++gi];
  }
  // ---- post layer 0: x-part weight gradient (queued), per-class weight gradient (side stream 0), dA
  for (int t = 0; t < T; ++t) {
    const int k = pidx(t, false, 0);
    const float* gt = g + t * F;
    const float* At = a->A + (int64_t)t * 4 * F;
    float* dWp = G[k];
    wq.add(gt, H, a->x + t * F, H, N, F, F, dWp, 13 * F, G[k + 1]);
    float* dWeff = a->dWeff + (int64_t)t * D * F * 4 * F;
    if (!defer) GNX_TRY(gnx_fill(h, dWeff, (int64_t)D * F * 4 * F, 0.f));
    auto class_wgrad = [&, gt, At, dWeff, dWp]() -> int32_t {
      GNX_TRY(gnx_gemm_wgrad_grouped(h, gt, H, At, (int64_t)T * 4 * F, N, F, 4 * F, dWeff, 4 * F, (int64_t)F * 4 * F, a->dperm,
                                     a->chunks, a->nchunks, a->max_chunks));
      return defer ? GNX_OK : gnx_pna_weff_bwd(h, dWeff, F, D, a->avg_deg_log, dWp, 13 * F);
    };
    if (class_after_agg)
      class_wgrads.push_back(class_wgrad);
    else
      GNX_TRY(on_side(h, 0, side, class_wgrad));
    if (ahead) {
      if (t == 0) GNX_TRY(gnx_side_join_n(h, 2));
      GNX_TRY(dA_call(t, gt, GNX_GEMM_PRESPLIT, ahead_base + (size_t)t * (r_dA + r_dx), r_dA));
    } else {
      GNX_TRY(dA_call(t, gt, 0, a->ws, scratch));
    }
  }
  const float* g_post0 = g;  // gradient w.r.t. post-layer 0's output: also an operand of dx below
  // the last layer of the pass: the weight gradients queued so far (lin, hidden post layers, x part of post-layer 0) go out
  // NOW, beside the edge backward, so that only the pre-layer ones are left for the end of the step
  if (tail && side && (a->defer_small & 4) != 0) GNX_TRY(on_side(h, 0, side, [&]() -> int32_t { return wq.flush(h); }));
  // ---- scatter-aggregate backward, then the pre layers last..1
  int ei = 0;
  float* ge = a->gebuf[ei];
  GNX_TRY(gnx_pna_aggregate_bwd(h, a->dA, a->hs[a->n_h - 1], a->A, a->rowptr, N, E, T, F, ge));
  for (auto& fn : class_wgrads) GNX_TRY(on_side(h, 0, side, fn));
  // With two pre layers (the reference's default) the masked input gradient of pre-layer 1, the destination sums dP and the
  // bond-table sums dTe come from ONE pass over the message gradient (gnx_pna_edge_bwd); only dQ is a pass of its own.
  bool fused_bwd = h->opt[GNX_OPT_EDGE_FUSED] == 1 && pre == 2 && a->etile_info != nullptr && E > 0 && F % 4 == 0 && F <= 128 &&
                   R <= 64 && a16(ge) && a16(a->hs[0]) && a16(a->gebuf[1]) && a16(a->dP) && a16(a->dTe);
  for (int t = 0; t < T && fused_bwd; ++t) fused_bwd = a16(P[pidx(t, true, 1)]);
  if (fused_bwd) {
    const float* W1[GNX_PNA_MAX_TOWERS];
    for (int t = 0; t < T; ++t) {
      const int k = pidx(t, true, 1);
      W1[t] = P[k];
      wq.add(ge + t * F, H, a->hs[0] + t * F, H, E, F, F, G[k], F, G[k + 1]);
    }
    if (!defer) GNX_TRY(gnx_fill(h, a->dTe, (int64_t)R * H, 0.f));
    GNX_TRY(gnx_pna_edge_bwd(h, ge, a->hs[0], a->code, a->rowptr, a->etile_info, a->etile_w, N, E, T, F, R, W1, a->gebuf[1], a->dP,
                             a->dTe));
    ge = a->gebuf[++ei];
    GNX_TRY(gnx_edge_combine_bwd(h, ge, a->rowptr, a->colptr, a->cpos, a->code, N, E, H, 0, nullptr, a->dQ, nullptr, nullptr, 0));
  } else {
    for (int i = pre - 1; i >= 1; --i) {
      const float* h_prev = a->hs[i - 1];
      for (int t = 0; t < T; ++t) {
        const int k = pidx(t, true, i);
        wq.add(ge + t * F, H, h_prev + t * F, H, E, F, F, G[k], F, G[k + 1]);
        gnx_gemm_seg s = seg(ge + t * F, H, P[k], F, F);
        GNX_TRY(gnx_gemm(h, 1, &s, E, F, nullptr, h_prev + t * F, H, a->gebuf[ei + 1] + t * F, H, 0, a->ws, a->ws_bytes));
      }
      ge = a->gebuf[++ei];
    }
    // ---- message assembly backward: dP, dQ; input gradient
    GNX_TRY(gnx_edge_combine_bwd(h, ge, a->rowptr, a->colptr, a->cpos, a->code, N, E, H, 0, a->dP, a->dQ, nullptr, nullptr, 0));
  }
  for (int t = 0; t < T; ++t) {
    float* dW0 = G[pidx(t, true, 0)];
    wq.add(a->dP + t * F, H, a->x + t * F, H, N, F, F, dW0, 3 * F, nullptr);
    wq.add(a->dQ + t * F, H, a->x + t * F, H, N, F, F, dW0 + F, 3 * F, nullptr);
  }
  // the queue is complete: with defer_small bit 4 the batched weight gradients start HERE, beside the dx product (matrix-
  // bound), instead of behind it, beside the next layer's BatchNorm backward (memory-bound)
  auto flush_batched = [&]() -> int32_t {
    // the LAST layer of a backward pass has the chip to itself: its batched weight gradients take every CU instead of the
    // workgroup budget that leaves room for the main stream (the end of a step waits for exactly this launch)
    const int wgs_saved = h->opt[GNX_OPT_WGRAD_WGS];
    if (tail && wgs_saved == 0) h->opt[GNX_OPT_WGRAD_WGS] = h->num_cus > 0 ? h->num_cus : 256;
    const int32_t fst = wq.flush(h);
    h->opt[GNX_OPT_WGRAD_WGS] = wgs_saved;
    return fst;
  };
  const bool flush_before_dx = (a->defer_small & 16) != 0 && side && !tail;
  if (flush_before_dx) GNX_TRY(on_side(h, 0, side, flush_batched));
  for (int t = 0; t < T; ++t) {
    if (ahead)
      GNX_TRY(dx_call(t, g_post0, GNX_GEMM_PRESPLIT, ahead_base + (size_t)t * (r_dA + r_dx) + r_dA, r_dx));
    else
      GNX_TRY(dx_call(t, g_post0, 0, a->ws, scratch));
  }
  // ---- bond-table gradient chain on side stream 1 (feeds parameter gradients and the shared accumulator only)
  GNX_TRY(on_side(h, 1, side, [&]() -> int32_t {
    if (!fused_bwd) {
      if (!defer) GNX_TRY(gnx_fill(h, a->dTe, (int64_t)R * H, 0.f));
      if (E > 0) GNX_TRY(gnx_key_segment_sum(h, ge, a->code_pos, a->code, E, H, a->dTe));
    }
    if (defer) return GNX_OK;
    for (int t = 0; t < T; ++t) {
      const int k0 = pidx(t, true, 0);
      GNX_TRY(gnx_gemm_wgrad(h, a->dTe + t * F, H, a->EE, F, nullptr, R, F, F, G[k0] + 2 * F, 3 * F, G[k0 + 1]));
      gnx_gemm_seg s = seg(a->dTe + t * F, H, P[k0] + 2 * F, 3 * F, F);
      GNX_TRY(gnx_gemm(h, 1, &s, R, F, nullptr, nullptr, 0, a->dEE, F, t > 0 ? GNX_GEMM_ACCUMULATE : 0, nullptr, 0));
    }
    GNX_TRY(gnx_gemm_wgrad(h, a->dEE, F, a->BE, H, nullptr, R, F, H, G[0], H, G[1]));
    if (a->acc_first) GNX_TRY(gnx_fill(h, a->acc_buf, (int64_t)R * H, 0.f));
    gnx_gemm_seg s = seg(a->dEE, F, P[0], H, F);
    return gnx_gemm(h, 1, &s, R, H, nullptr, nullptr, 0, a->acc_buf, H, GNX_GEMM_ACCUMULATE, nullptr, 0);
  }));
  // ---- the layer's weight gradients in batched launches on side stream 0, then what hangs off dWm
  GNX_TRY(on_side(h, 0, side, [&]() -> int32_t {
    if (!flush_before_dx) GNX_TRY(flush_batched());
    if (!a->merged || defer) return GNX_OK;
    const float* dbm = a->dbm;
    for (int t = 0; t < T; ++t) {
      const int k = pidx(t, false, post - 1);
      const float* Wt = P[k];
      const float* bt = P[k + 1];
      // d lin_w[:, t] += dWm[:, t] W_t^T + dbm b_t^T ; dW_t += lin_w[:, t]^T dWm[:, t] ; db_t += dbm lin_w[:, t]
      gnx_gemm_seg s1 = seg(a->dWm + t * F, H, Wt, F, F);
      GNX_TRY(gnx_gemm(h, 1, &s1, H, F, nullptr, nullptr, 0, G[2] + t * F, H, GNX_GEMM_ACCUMULATE | GNX_GEMM_B_TRANS, nullptr, 0));
      gnx_gemm_seg s2 = seg(dbm, 1, bt, 1, 1);
      GNX_TRY(gnx_gemm(h, 1, &s2, H, F, nullptr, nullptr, 0, G[2] + t * F, H, GNX_GEMM_ACCUMULATE | GNX_GEMM_B_TRANS, nullptr, 0));
      GNX_TRY(gnx_gemm_wgrad(h, lin_w + t * F, H, a->dWm + t * F, H, nullptr, H, F, F, G[k], F, nullptr));
      gnx_gemm_seg s3 = seg(dbm, H, lin_w + t * F, H, H);
      GNX_TRY(gnx_gemm(h, 1, &s3, 1, F, nullptr, nullptr, 0, G[k + 1], F, GNX_GEMM_ACCUMULATE, nullptr, 0));
    }
    return gnx_axpy(h, G[3], dbm, H, 1.0f);
  }));
  return GNX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Weight-only part of a PNAConv forward (functional._pna_weight_only): EE = BondEmb W_enc^T + b_enc, Te (edge slice of
// pre-layer 0 on the 60-row bond table), Weff(d) per tower, merged lin o last post layer.  5 + 3(T-1) tiny launches.
// ---------------------------------------------------------------------------------------------------------------
extern "C" int32_t gnx_pna_weight_only(gnx_handle* h, const float* BE, int32_t R, int32_t T, int32_t F, int32_t pre_layers,
                                       int32_t post_layers, int32_t D, float avg_deg_log, const float* const* params,
                                       int32_t merged, float* EE, float* Te, float* const* weff, float* Wm, float* bm) {
  GNX_CHECK_ARG(h && BE && params && EE && Te && R > 0 && T >= 1 && T <= GNX_PNA_MAX_TOWERS && F >= 1 && pre_layers >= 1 &&
                    post_layers >= 1, "gnx_pna_weight_only: bad argument");
  const int H = T * F, per = 2 * (pre_layers + post_layers);
  gnx_gemm_seg s = seg(BE, H, params[0], H, H);
  GNX_TRY(gnx_gemm(h, 1, &s, R, F, params[1], nullptr, 0, EE, F, GNX_GEMM_B_TRANS, nullptr, 0));
  for (int t = 0; t < T; ++t) {
    const float* W0 = params[4 + t * per];
    gnx_gemm_seg se = seg(EE, F, W0 + 2 * F, 3 * F, F);
    GNX_TRY(gnx_gemm(h, 1, &se, R, F, params[4 + t * per + 1], nullptr, 0, Te + t * F, H, GNX_GEMM_B_TRANS, nullptr, 0));
    if (D > 0) {
      GNX_CHECK_ARG(weff && weff[t], "gnx_pna_weight_only: weff[%d] is NULL", t);
      GNX_TRY(gnx_pna_weff(h, params[4 + t * per + 2 * pre_layers], 13 * F, F, D, avg_deg_log, weff[t]));
    }
  }
  if (merged) {
    GNX_CHECK_ARG(Wm && bm && post_layers > 1, "gnx_pna_weight_only: merged without Wm / bm");
    const float* lin_w = params[2];
    for (int t = 0; t < T; ++t) {
      const int k = 4 + t * per + 2 * (pre_layers + post_layers - 1);
      gnx_gemm_seg sw = seg(lin_w + t * F, H, params[k], F, F);                        // Wm[:, t] = lin_w[:, t] @ W_last_t
      GNX_TRY(gnx_gemm(h, 1, &sw, H, F, nullptr, nullptr, 0, Wm + t * F, H, 0, nullptr, 0));
      gnx_gemm_seg sb = seg(params[k + 1], F, lin_w + t * F, H, F);                    // bm (+)= b_last_t @ lin_w[:, t]^T
      GNX_TRY(gnx_gemm(h, 1, &sb, 1, H, t == 0 ? params[3] : nullptr, nullptr, 0, bm, H,
                       GNX_GEMM_B_TRANS | (t > 0 ? GNX_GEMM_ACCUMULATE : 0), nullptr, 0));
    }
  }
  return GNX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Forward launch sequence of one PNAConv (functional.PNAConvFn.forward after the weight-only part): node-level P / Q
// products, message assembly, remaining pre layers, scatter-aggregate, post-layer 0 by degree class, hidden post
// layers, lin (or the merged product).  Every activation the backward needs is written into caller-owned buffers.
// ---------------------------------------------------------------------------------------------------------------
extern "C" int32_t gnx_pna_conv_fwd(gnx_handle* h, const gnx_pna_fwd_args* a) {
  GNX_CHECK_ARG(h && a, "gnx_pna_conv_fwd: NULL argument");
  const int T = a->T, F = a->F, pre = a->pre_layers, post = a->post_layers, D = a->D;
  const int64_t N = a->N, E = a->E;
  const int H = T * F;
  GNX_CHECK_ARG(T >= 1 && T <= GNX_PNA_MAX_TOWERS && F >= 1 && pre >= 1 && pre <= GNX_PNA_MAX_LAYERS && post >= 1 &&
                    post <= GNX_PNA_MAX_LAYERS && D >= 1, "gnx_pna_conv_fwd: bad layer shape");
  GNX_CHECK_ARG(a->params && a->x && a->Te && a->P && a->Q && a->A && a->out, "gnx_pna_conv_fwd: NULL array");
  const int per = 2 * (pre + post);
  const float* const* W = a->params;
  // post-layer 0's weight images (x part + Weff(d), per tower) are split on side stream 2 while the node products and the
  // edge pipeline run; the product then takes them as they are (GNX_GEMM_PRESPLIT)
  const int post0_flags = GNX_GEMM_B_TRANS | (post > 1 ? GNX_GEMM_RELU : 0);
  const int post0_rows = a->tile_rows == 96 ? 96 : 128;
  size_t scratch = a->ws_bytes, r_post0 = 0, r_grouped = 0, r_dx = 0;
  // (measured: cfg-5, four towers = twelve splits per layer: -0.44 ms per step; cfg-2, one tower: +0.02 ms -- the fork / join
  // events cost what the three 7-us splits saved -- so: 1 = with two or more towers, 2 = always, 0 = never)
  bool ahead = (h->opt[GNX_OPT_SPLIT_AHEAD] == 2 || (h->opt[GNX_OPT_SPLIT_AHEAD] == 1 && T >= 2)) && !h->on_side && N >= 4096 &&
               a->ws != nullptr;
  if (ahead) {
    pna_ws_sizes(T, F, D, scratch, r_post0, r_grouped, r_dx);
    ahead = scratch + (size_t)T * r_post0 <= a->ws_bytes;
    if (!ahead) scratch = a->ws_bytes;
  }
  unsigned char* const ahead_base = reinterpret_cast<unsigned char*>(a->ws) + scratch;
  auto post0_call = [&](int t, int extra_flags, void* ws, size_t ws_bytes) -> int32_t {
    const int k = 4 + t * per + 2 * pre;
    gnx_gemm_seg s2[2] = {seg(a->x + t * F, H, W[k], 13 * F, F),
                          seg(a->A + (int64_t)t * 4 * F, (int64_t)T * 4 * F, a->weff[t], 4 * F, 4 * F)};
    const int64_t strides[2] = {0, (int64_t)4 * F * F};
    return gnx_gemm_grouped_rows(h, 2, s2, strides, D, N, F, W[k + 1], nullptr, 0, a->zs[0] + t * F, H, post0_flags | extra_flags,
                                 a->dperm, a->tiles, a->ntiles, a->max_tiles, ws, ws_bytes, post0_rows);
  };
  if (ahead) {
    GNX_TRY(gnx_side_begin_n(h, 2));
    int32_t st = GNX_OK;
    for (int t = 0; t < T && st == GNX_OK; ++t) st = post0_call(t, GNX_GEMM_SPLIT_ONLY, ahead_base + (size_t)t * r_post0, r_post0);
    (void)gnx_side_end(h);
    GNX_TRY(st);
  }
  for (int t = 0; t < T; ++t) {
    const float* W0 = W[4 + t * per];
    gnx_gemm_seg sp = seg(a->x + t * F, H, W0, 3 * F, F), sq = seg(a->x + t * F, H, W0 + F, 3 * F, F);
    GNX_TRY(gnx_gemm(h, 1, &sp, N, F, nullptr, nullptr, 0, a->P + t * F, H, GNX_GEMM_B_TRANS, a->ws, scratch));
    GNX_TRY(gnx_gemm(h, 1, &sq, N, F, nullptr, nullptr, 0, a->Q + t * F, H, GNX_GEMM_B_TRANS, a->ws, scratch));
  }
  // edge pipeline: message assembly -> pre layers 1.. -> scatter-aggregate.  With two pre layers (the reference's default)
  // it is ONE launch (gnx_pna_edge_fwd: h1 and the messages are written once and never read back, bit-identical results)
  bool fused = h->opt[GNX_OPT_EDGE_FUSED] != 0 && pre == 2 && a->etile_info != nullptr && E > 0 && F % 4 == 0 && F <= 128 &&
               a16(a->P) && a16(a->Q) && a16(a->Te) && a16(a->hs[0]) && a16(a->hs[1]) && a16(a->A);
  const float* W1[GNX_PNA_MAX_TOWERS];
  const float* b1[GNX_PNA_MAX_TOWERS];
  for (int t = 0; t < T && fused; ++t) {
    W1[t] = W[4 + t * per + 2];
    b1[t] = W[4 + t * per + 3];
    fused = a16(W1[t]);
  }
  if (fused) {
    GNX_TRY(gnx_pna_edge_fwd(h, a->P, a->Q, a->Te, a->src, a->dst, a->code, a->rowptr, a->etile_info, a->etile_w, N, E, T, F,
                             W1, b1, a->hs[0], a->hs[1], a->A));
  } else {
    GNX_TRY(gnx_edge_combine_fwd(h, a->P, a->Q, a->Te, a->src, a->dst, a->code, E, H, pre > 1 ? 1 : 0, a->hs[0]));
    for (int i = 1; i < pre; ++i)
      for (int t = 0; t < T; ++t) {
        const int k = 4 + t * per + 2 * i;
        gnx_gemm_seg s = seg(a->hs[i - 1] + t * F, H, W[k], F, F);
        GNX_TRY(gnx_gemm(h, 1, &s, E, F, W[k + 1], nullptr, 0, a->hs[i] + t * F, H,
                         GNX_GEMM_B_TRANS | (i < pre - 1 ? GNX_GEMM_RELU : 0), a->ws, scratch));
      }
    GNX_TRY(gnx_pna_aggregate_fwd(h, a->hs[pre - 1], a->rowptr, N, E, T, F, a->A));
  }
  if (ahead) GNX_TRY(gnx_side_join_n(h, 2));
  for (int t = 0; t < T; ++t) {
    if (ahead)
      GNX_TRY(post0_call(t, GNX_GEMM_PRESPLIT, ahead_base + (size_t)t * r_post0, r_post0));
    else
      GNX_TRY(post0_call(t, 0, a->ws, scratch));
  }
  const int hidden_end = a->merged ? post - 1 : post;  // hidden layers 1 .. hidden_end-1 are evaluated one by one
  int zi = 0;
  for (int i = 1; i < hidden_end; ++i, ++zi)
    for (int t = 0; t < T; ++t) {
      const int k = 4 + t * per + 2 * (pre + i);
      gnx_gemm_seg s = seg(a->zs[zi] + t * F, H, W[k], F, F);
      GNX_TRY(gnx_gemm(h, 1, &s, N, F, W[k + 1], nullptr, 0, a->zs[zi + 1] + t * F, H,
                       GNX_GEMM_B_TRANS | (i < post - 1 ? GNX_GEMM_RELU : 0), a->ws, a->ws_bytes));
    }
  gnx_gemm_seg sl = a->merged ? seg(a->zs[zi], H, a->Wm, H, H) : seg(a->zs[zi], H, W[2], H, H);
  return gnx_gemm(h, 1, &sl, N, H, a->merged ? a->bm : W[3], nullptr, 0, a->out, H, GNX_GEMM_B_TRANS, a->ws, a->ws_bytes);
}

// ---------------------------------------------------------------------------------------------------------------
// Model-level batching of the weight-only work.  Everything below depends on weights (and 60-row
// tables) only; issued per layer it was ~14 launches of ~5-10 us each per layer and direction.  Here: all layers at once.
// ---------------------------------------------------------------------------------------------------------------
namespace {
inline gnx_small_prob sprob(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float* C, int64_t ldc,
                            int M, int N, int K, int flags) {
  gnx_small_prob q;
  q.A = A;
  q.lda = lda;
  q.B = B;
  q.ldb = ldb;
  q.bias = bias;
  q.C = C;
  q.ldc = ldc;
  q.M = M;
  q.N = N;
  q.K = K;
  q.flags = flags;
  return q;
}
}  // namespace

extern "C" int32_t gnx_pna_weight_only_all(gnx_handle* h, int32_t L, const float* BE, int32_t R, int32_t T, int32_t F,
                                           int32_t pre_layers, int32_t post_layers, int32_t D, const float* avg_deg_log,
                                           const float* const* params, int32_t merged, float* const* EE, float* const* Te,
                                           float* const* weff, float* const* Wm, float* const* bm) {
  GNX_CHECK_ARG(h && BE && params && EE && Te && avg_deg_log && L >= 1 && R > 0 && T >= 1 && T <= GNX_PNA_MAX_TOWERS && F >= 1 &&
                    pre_layers >= 1 && post_layers >= 1, "gnx_pna_weight_only_all: bad argument");
  GNX_CHECK_ARG(!merged || (Wm && bm && post_layers > 1), "gnx_pna_weight_only_all: merged without Wm / bm");
  GNX_CHECK_ARG(D <= 0 || weff, "gnx_pna_weight_only_all: weff is NULL");
  const int H = T * F, per = 2 * (pre_layers + post_layers), np = 4 + T * per;
  std::vector<gnx_small_prob> first, second;
  std::vector<std::vector<gnx_small_prob>> later(T > 2 ? T - 2 : 0);  // bm's towers 2.. : one more launch each
  std::vector<const float*> wsrc;
  std::vector<float*> wdst;
  std::vector<float> wavg;
  for (int l = 0; l < L; ++l) {
    const float* const* P = params + (size_t)l * np;
    // EE = BondEmb W_enc^T + b_enc
    first.push_back(sprob(BE, H, P[0], H, P[1], EE[l], F, R, F, H, GNX_SB_B_TRANS));
    for (int t = 0; t < T; ++t) {
      const float* W0 = P[4 + t * per];
      // Te[:, t] = EE W_e^T + b  (the edge slice of pre-layer 0 on the bond table): needs EE -> second launch
      second.push_back(sprob(EE[l], F, W0 + 2 * F, 3 * F, P[4 + t * per + 1], Te[l] + t * F, H, R, F, F, GNX_SB_B_TRANS));
      if (D > 0) {
        wsrc.push_back(P[4 + t * per + 2 * pre_layers]);
        wdst.push_back(weff[(size_t)l * T + t]);
        wavg.push_back(avg_deg_log[l]);
      }
      if (merged) {
        const int k = 4 + t * per + 2 * (pre_layers + post_layers - 1);
        // Wm[:, t] = lin_w[:, t] @ W_last_t ;  bm (+)= b_last_t @ lin_w[:, t]^T (+ lin_b with the first tower)
        first.push_back(sprob(P[2] + t * F, H, P[k], F, nullptr, Wm[l] + t * F, H, H, F, F, 0));
        if (t == 0)
          first.push_back(sprob(P[k + 1], F, P[2], H, P[3], bm[l], H, 1, H, F, GNX_SB_B_TRANS));
        else  // further towers add on top of the previous one's result, one launch later each: a deterministic forward
          (t == 1 ? second : later[t - 2])
              .push_back(sprob(P[k + 1], F, P[2] + t * F, H, nullptr, bm[l], H, 1, H, F, GNX_SB_B_TRANS | GNX_SB_ACCUMULATE));
      }
    }
  }
  GNX_TRY(gnx_gemm_small_batched(h, (int32_t)first.size(), first.data()));
  GNX_TRY(gnx_gemm_small_batched(h, (int32_t)second.size(), second.data()));
  for (auto& v : later) GNX_TRY(gnx_gemm_small_batched(h, (int32_t)v.size(), v.data()));
  if (D > 0)
    GNX_TRY(gnx_pna_weff_batched(h, (int32_t)wsrc.size(), wsrc.data(), 13 * F, F, D, wavg.data(), wdst.data()));
  return GNX_OK;
}

extern "C" int32_t gnx_pna_stack_finish(gnx_handle* h, const gnx_pna_finish_args* a) {
  GNX_CHECK_ARG(h && a, "gnx_pna_stack_finish: NULL argument");
  const int L = a->L, T = a->T, F = a->F, pre = a->pre_layers, post = a->post_layers, R = a->R, D = a->D;
  GNX_CHECK_ARG(L >= 1 && T >= 1 && T <= GNX_PNA_MAX_TOWERS && F >= 1 && pre >= 1 && post >= 1 && R >= 1 && D >= 1,
                "gnx_pna_stack_finish: bad shape");
  GNX_CHECK_ARG(a->BE && a->acc_buf && a->ones && a->avg_deg_log && a->params && a->grads && a->EE && a->dTe && a->dEE &&
                    a->dWeff && (!a->merged || (a->dWm && a->dbm)), "gnx_pna_stack_finish: NULL array");
  const int H = T * F, per = 2 * (pre + post), np = 4 + T * per;
  const bool side = a->use_side_streams != 0;
  // ---- side stream 1: from every layer's dTe (by-code segment sums) to the bond-embedding gradient
  GNX_TRY(on_side(h, 1, side, [&]() -> int32_t {
    std::vector<gnx_small_prob> s1, s2;
    for (int l = 0; l < L; ++l) {
      const float* const* P = a->params + (size_t)l * np;
      float* const* G = a->grads + (size_t)l * np;
      for (int t = 0; t < T; ++t) {
        const int k0 = 4 + t * per;
        const float* dTe = a->dTe[l] + t * F;
        // d W0[:, 2F:3F] += dTe_t^T EE ;  d b0 += column sums of dTe_t ;  dEE += dTe_t W0[:, 2F:3F]
        s1.push_back(sprob(dTe, H, a->EE[l], F, nullptr, G[k0] + 2 * F, 3 * F, F, F, R, GNX_SB_A_TRANS | GNX_SB_ACCUMULATE));
        s1.push_back(sprob(a->ones, R, dTe, H, nullptr, G[k0 + 1], F, 1, F, R, GNX_SB_ACCUMULATE));
        s1.push_back(sprob(dTe, H, P[k0] + 2 * F, 3 * F, nullptr, a->dEE[l], F, R, F, F, GNX_SB_ATOMIC));
      }
      // d W_enc += dEE^T BondEmb ;  d b_enc += column sums of dEE ;  acc += dEE W_enc
      s2.push_back(sprob(a->dEE[l], F, a->BE, H, nullptr, G[0], H, F, H, R, GNX_SB_A_TRANS | GNX_SB_ACCUMULATE));
      s2.push_back(sprob(a->ones, R, a->dEE[l], F, nullptr, G[1], F, 1, F, R, GNX_SB_ACCUMULATE));
      s2.push_back(sprob(a->dEE[l], F, P[0], H, nullptr, a->acc_buf, H, R, H, F, GNX_SB_ATOMIC));
    }
    GNX_TRY(gnx_gemm_small_batched(h, (int32_t)s1.size(), s1.data()));
    return gnx_gemm_small_batched(h, (int32_t)s2.size(), s2.data());
  }));
  // ---- side stream 0, behind the layers' weight gradients: lin / last post layer from dWm, dbm; post-layer 0 from dWeff
  GNX_TRY(on_side(h, 0, side, [&]() -> int32_t {
    std::vector<const float*> dweff;
    std::vector<float*> dwp;
    std::vector<float> avg;
    std::vector<gnx_small_prob> u1, u2;
    for (int l = 0; l < L; ++l) {
      const float* const* P = a->params + (size_t)l * np;
      float* const* G = a->grads + (size_t)l * np;
      for (int t = 0; t < T; ++t) {
        dweff.push_back(a->dWeff[(size_t)l * T + t]);
        dwp.push_back(G[4 + t * per + 2 * pre]);
        avg.push_back(a->avg_deg_log[l]);
        if (!a->merged) continue;
        const int k = 4 + t * per + 2 * (pre + post - 1);
        const float* dWm = a->dWm[l] + t * F;
        const float* lin_w = P[2] + t * F;
        // d lin_w[:, t] += dWm[:, t] W_t^T (+ dbm b_t^T, one launch later: same output) ;  dW_t += lin_w[:, t]^T dWm[:, t] ;
        // db_t += dbm lin_w[:, t]
        u1.push_back(sprob(dWm, H, P[k], F, nullptr, G[2] + t * F, H, H, F, F, GNX_SB_B_TRANS | GNX_SB_ACCUMULATE));
        u2.push_back(sprob(a->dbm[l], 1, P[k + 1], 1, nullptr, G[2] + t * F, H, H, F, 1, GNX_SB_B_TRANS | GNX_SB_ACCUMULATE));
        u1.push_back(sprob(lin_w, H, dWm, H, nullptr, G[k], F, F, F, H, GNX_SB_A_TRANS | GNX_SB_ACCUMULATE));
        u1.push_back(sprob(a->dbm[l], H, lin_w, H, nullptr, G[k + 1], F, 1, F, H, GNX_SB_ACCUMULATE));
      }
      if (a->merged)  // d lin_b += dbm
        u1.push_back(sprob(a->ones, 1, a->dbm[l], H, nullptr, G[3], H, 1, H, 1, GNX_SB_ACCUMULATE));
    }
    GNX_TRY(gnx_pna_weff_bwd_batched(h, (int32_t)dweff.size(), dweff.data(), F, D, avg.data(), dwp.data(), 13 * F));
    GNX_TRY(gnx_gemm_small_batched(h, (int32_t)u1.size(), u1.data()));
    return gnx_gemm_small_batched(h, (int32_t)u2.size(), u2.data());
  }));
  return GNX_OK;
}

// The gather / scatter side of message passing, written against the dst-sorted CSR produced by gnx_pack_csr:
// every "scatter" of the reference (scatter_add_ / scatter_reduce_ over edge_index[1]) becomes a per-destination
// sequential reduction over a CONTIGUOUS run of rows -- coalesced streaming reads, no atomics, and the same summation
// order as the CPU scatter (edges of one destination in ascending edge id), so sums/means are reproducible.
//
// Thread mapping everywhere: one thread owns VEC (=4 when the width allows, else 1) consecutive channels of one
// destination row; G = H/VEC threads cover a row, so a 64-lane wave reads whole 1 KiB-aligned runs of a row-major
// [rows, H] array with 16-byte lane accesses.
#include "gnx_common.hpp"

#include <cmath>
#include <cstdlib>

int32_t gnx_code_scatter_add(gnx_handle* h, const int32_t* code, int64_t E, int R, const float* g, int H,
                             float* dtable, void* ws, size_t ws_bytes);

template <int VEC>
struct vec_t;
template <>
struct vec_t<4> {
  typedef f32x4 type;
};
template <>
struct vec_t<1> {
  typedef float type;
};

template <int VEC>
__device__ __forceinline__ void vload(float (&r)[VEC], const float* p) {
  if constexpr (VEC == 4) {
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    r[0] = v.x;
    r[1] = v.y;
    r[2] = v.z;
    r[3] = v.w;
  } else {
    r[0] = p[0];
  }
}
template <int VEC>
__device__ __forceinline__ void vstore(float* p, const float (&r)[VEC]) {
  if constexpr (VEC == 4) {
    f32x4 v = {r[0], r[1], r[2], r[3]};
    *reinterpret_cast<f32x4*>(p) = v;
  } else {
    p[0] = r[0];
  }
}

// clamp / mask constants of [3P] StdAggregation: var.clamp(min=1e-5).sqrt(), masked to 0 where <= sqrt(1e-5)
#define STD_VAR_MIN 1e-5f
#define STD_MASK_AT 0.0031622776601683794f

// ---------------------------------------------------------------------------------------------------------------
// PNA multi-aggregate forward: m[E, H] (CSR order) -> A[N, T, 4F] = per tower [mean | min | max | std]
// ---------------------------------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(256) k_pna_agg_fwd(const float* __restrict__ m, const int* __restrict__ rowptr,
                                                     int64_t N, int T, int F, float* __restrict__ A) {
  const int H = T * F;
  const int G = H / VEC;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * G) return;
  int64_t n = t / G;
  int c = (int)(t % G) * VEC;
  int p0 = rowptr[n], p1 = rowptr[n + 1];
  float s[VEC], s2[VEC], mn[VEC], mx[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    s[v] = 0.f;
    s2[v] = 0.f;
    mn[v] = INFINITY;
    mx[v] = -INFINITY;
  }
  const float* mp = m + (int64_t)p0 * H + c;
  int p = p0;
  // two rows in flight per iteration (degrees are 1..4 for molecules)
  for (; p + 1 < p1; p += 2, mp += 2 * (int64_t)H) {
    float a[VEC], b[VEC];
    vload<VEC>(a, mp);
    vload<VEC>(b, mp + H);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      s[v] = __fadd_rn(__fadd_rn(s[v], a[v]), b[v]);
      s2[v] = __fadd_rn(__fadd_rn(s2[v], __fmul_rn(a[v], a[v])), __fmul_rn(b[v], b[v]));
      mn[v] = fminf(mn[v], fminf(a[v], b[v]));
      mx[v] = fmaxf(mx[v], fmaxf(a[v], b[v]));
    }
  }
  if (p < p1) {
    float a[VEC];
    vload<VEC>(a, mp);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      s[v] = __fadd_rn(s[v], a[v]);
      s2[v] = __fadd_rn(s2[v], __fmul_rn(a[v], a[v]));
      mn[v] = fminf(mn[v], a[v]);
      mx[v] = fmaxf(mx[v], a[v]);
    }
  }
  const int d = p1 - p0;
  const float cnt = (float)(d > 0 ? d : 1);
  float mean[VEC], sd[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    mean[v] = __fdiv_rn(s[v], cnt);
    float mean2 = __fdiv_rn(s2[v], cnt);
    float var = __fsub_rn(mean2, __fmul_rn(mean[v], mean[v]));
    float o = __fsqrt_rn(fmaxf(var, STD_VAR_MIN));
    sd[v] = (o <= STD_MASK_AT) ? 0.f : o;
    if (d == 0) {
      mn[v] = 0.f;
      mx[v] = 0.f;
    }
  }
  int tw = c / F, f = c % F;
  float* o = A + (n * T + tw) * (int64_t)(4 * F) + f;
  vstore<VEC>(o, mean);
  vstore<VEC>(o + F, mn);
  vstore<VEC>(o + 2 * F, mx);
  vstore<VEC>(o + 3 * F, sd);
}

extern "C" int32_t gnx_pna_aggregate_fwd(gnx_handle* h, const float* m, const int32_t* rowptr, int64_t N, int64_t E,
                                         int32_t T, int32_t F, float* A) {
  GNX_CHECK_ARG(h && rowptr && T > 0 && F > 0 && N >= 0 && E >= 0, "gnx_pna_aggregate_fwd: bad argument");
  GNX_CHECK_ARG(N == 0 || A, "gnx_pna_aggregate_fwd: A is NULL");
  if (N == 0) return GNX_OK;
  // algorithmic bytes: read the messages 4EH and the index 4E, write the four aggregates 16NH
  gnx_prof_scope prof(h, GNX_K_PNA_AGG_FWD, 4.0 * E * T * F + 4.0 * E + 16.0 * N * T * F, 0.0, 0.0, true);
  if (F % 4 == 0) {
    int64_t th = N * (T * F / 4);
    GNX_LAUNCH_TIMED(prof, k_pna_agg_fwd<4>, dim3((unsigned)gnx_cdiv(th, 256)), dim3(256), 0, h->stream, m, rowptr, N,
                     (int)T, (int)F, A);
  } else {
    int64_t th = N * (T * F);
    GNX_LAUNCH_TIMED(prof, k_pna_agg_fwd<1>, dim3((unsigned)gnx_cdiv(th, 256)), dim3(256), 0, h->stream, m, rowptr, N,
                     (int)T, (int)F, A);
  }
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// backward: dm[p,c] = dmean/cnt + [m==min] dmin/#ties + [m==max] dmax/#ties + [std>0] dstd (m-mean)/(cnt std)
// ---------------------------------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(256) k_pna_agg_bwd(const float* __restrict__ dA, const float* __restrict__ m,
                                                     const float* __restrict__ A, const int* __restrict__ rowptr,
                                                     int64_t N, int T, int F, float* __restrict__ dm, int centered) {
  const int H = T * F;
  const int G = H / VEC;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * G) return;
  int64_t n = t / G;
  int c = (int)(t % G) * VEC;
  int p0 = rowptr[n], p1 = rowptr[n + 1];
  if (p1 <= p0) return;
  int tw = c / F, f = c % F;
  int64_t ao = (n * T + tw) * (int64_t)(4 * F) + f;
  float mean[VEC], mn[VEC], mx[VEC], sd[VEC], gmean[VEC], gmn[VEC], gmx[VEC], gsd[VEC];
  vload<VEC>(mean, A + ao);
  vload<VEC>(mn, A + ao + F);
  vload<VEC>(mx, A + ao + 2 * F);
  vload<VEC>(sd, A + ao + 3 * F);
  vload<VEC>(gmean, dA + ao);
  vload<VEC>(gmn, dA + ao + F);
  vload<VEC>(gmx, dA + ao + 2 * F);
  vload<VEC>(gsd, dA + ao + 3 * F);
  // torch's scatter_reduce(amin/amax) backward divides by (#src ties + [self == result]) with self = the zero-filled
  // output buffer, also under include_self=False: an extremum that is exactly 0 counts one extra tie.  The reference's
  // CPU path behaves that way (verified on torch 2.10: src [0,-1,0] -> grads [1/3,0,1/3]); reproduced here.
  float nmn[VEC], nmx[VEC], c2[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    nmn[v] = (mn[v] == 0.f) ? 1.f : 0.f;
    nmx[v] = (mx[v] == 0.f) ? 1.f : 0.f;
    c2[v] = 0.f;
  }
  const float* mp = m + (int64_t)p0 * H + c;
  for (int p = p0; p < p1; ++p, mp += H) {
    float a[VEC];
    vload<VEC>(a, mp);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      nmn[v] += (a[v] == mn[v]) ? 1.f : 0.f;
      nmx[v] += (a[v] == mx[v]) ? 1.f : 0.f;
      const float dv = a[v] - mean[v];
      c2[v] = fmaf(dv, dv, c2[v]);
    }
  }
  const float cnt = (float)(p1 - p0);
  float k_mean[VEC], k_mn[VEC], k_mx[VEC], k_sd[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    k_mean[v] = gmean[v] / cnt;
    k_mn[v] = gmn[v] / nmn[v];
    k_mx[v] = gmx[v] / nmx[v];
    // masked or not is the forward's decision (sd, bit for bit); the divisor is the centred two-pass std, free of the
    // cancellation error of mean(x^2) - mean(x)^2 (centered = 0: the forward's value, like the CPU path's backward)
    const float sdiv = (centered && c2[v] > 0.f) ? sqrtf(c2[v] / cnt) : sd[v];
    k_sd[v] = (sd[v] > 0.f) ? gsd[v] / (cnt * sdiv) : 0.f;
  }
  mp = m + (int64_t)p0 * H + c;
  float* dp = dm + (int64_t)p0 * H + c;
  for (int p = p0; p < p1; ++p, mp += H, dp += H) {
    float a[VEC], o[VEC];
    vload<VEC>(a, mp);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float r = k_mean[v] + k_sd[v] * (a[v] - mean[v]);
      r += (a[v] == mn[v]) ? k_mn[v] : 0.f;
      r += (a[v] == mx[v]) ? k_mx[v] : 0.f;
      o[v] = r;
    }
    vstore<VEC>(dp, o);
  }
}


// Variant that does not read the saved aggregate A: mean / min / max / std are recomputed from the message rows with the
// forward's exact arithmetic (same bits, so the std mask decision is identical), which drops 16NH bytes of reads — the
// rows of one destination (<= a few KB) are re-read from L1/L2 by the tie-count and write passes.
template <int VEC>
__global__ void __launch_bounds__(256) k_pna_agg_bwd_rc(const float* __restrict__ dA, const float* __restrict__ m,
                                                        const int* __restrict__ rowptr, int64_t N, int T, int F,
                                                        float* __restrict__ dm, int centered, int small_deg) {
  const int H = T * F;
  const int G = H / VEC;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * G) return;
  int64_t n = t / G;
  int c = (int)(t % G) * VEC;
  int p0 = rowptr[n], p1 = rowptr[n + 1];
  if (p1 <= p0) return;
  int tw = c / F, f = c % F;
  int64_t ao = (n * T + tw) * (int64_t)(4 * F) + f;
  float gmean[VEC], gmn[VEC], gmx[VEC], gsd[VEC];
  vload<VEC>(gmean, dA + ao);
  vload<VEC>(gmn, dA + ao + F);
  vload<VEC>(gmx, dA + ao + 2 * F);
  vload<VEC>(gsd, dA + ao + 3 * F);
  float s[VEC], s2[VEC], mn[VEC], mx[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    s[v] = 0.f;
    s2[v] = 0.f;
    mn[v] = INFINITY;
    mx[v] = -INFINITY;
  }
  // (measured and rejected, round 2: keeping the node's first 8 messages in registers for the three passes -- 71 vs 67 us at
  // cfg-2, 0.43 vs 0.59 of HBM as run at cfg-5: the extra 32 VGPRs cost more occupancy than the L1-resident re-reads cost)
  // (the same treatment of k_edge_combine_bwd's dQ gather and of k_gine_fwd / k_gine_bwd_dx -- indices, then rows, four at a time
  // -- was measured and rejected: cfg-3 20.15-20.23 vs 19.99-20.00 ms, cfg-2 neutral; those kernels sit at 0.76 of HBM with
  // one row in flight per thread and lose more occupancy to the 32 extra VGPRs than the loads gain)
  if (small_deg && p1 - p0 <= 4) {
    // in-degree <= 4 (every atom of an organic molecule): the row's messages are loaded ONCE, all four loads in flight
    // together (clamped addresses past the row's end), and the three passes run on registers -- same operations in the
    // same order per element as the loops below
    const int deg = p1 - p0;
    float a[4][VEC];
#pragma unroll
    for (int e = 0; e < 4; ++e) vload<VEC>(a[e], m + (int64_t)(p0 + (e < deg ? e : 0)) * H + c);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < deg) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          s[v] = __fadd_rn(s[v], a[e][v]);
          s2[v] = __fadd_rn(s2[v], __fmul_rn(a[e][v], a[e][v]));
          mn[v] = fminf(mn[v], a[e][v]);
          mx[v] = fmaxf(mx[v], a[e][v]);
        }
      }
    const float cnt = (float)deg;
    float mean[VEC], sd[VEC], nmn[VEC], nmx[VEC], c2[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      c2[v] = 0.f;
      mean[v] = __fdiv_rn(s[v], cnt);
      float mean2 = __fdiv_rn(s2[v], cnt);
      float var = __fsub_rn(mean2, __fmul_rn(mean[v], mean[v]));
      float o = __fsqrt_rn(fmaxf(var, STD_VAR_MIN));
      sd[v] = (o <= STD_MASK_AT) ? 0.f : o;
      nmn[v] = (mn[v] == 0.f) ? 1.f : 0.f;
      nmx[v] = (mx[v] == 0.f) ? 1.f : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < deg) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          nmn[v] += (a[e][v] == mn[v]) ? 1.f : 0.f;
          nmx[v] += (a[e][v] == mx[v]) ? 1.f : 0.f;
          const float dv = a[e][v] - mean[v];
          c2[v] = fmaf(dv, dv, c2[v]);
        }
      }
    float k_mean[VEC], k_mn[VEC], k_mx[VEC], k_sd[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      k_mean[v] = gmean[v] / cnt;
      k_mn[v] = gmn[v] / nmn[v];
      k_mx[v] = gmx[v] / nmx[v];
      const float sdiv = (centered && c2[v] > 0.f) ? sqrtf(c2[v] / cnt) : sd[v];
      k_sd[v] = (sd[v] > 0.f) ? gsd[v] / (cnt * sdiv) : 0.f;
    }
    float* dp = dm + (int64_t)p0 * H + c;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < deg) {
        float o[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          float r = k_mean[v] + k_sd[v] * (a[e][v] - mean[v]);
          r += (a[e][v] == mn[v]) ? k_mn[v] : 0.f;
          r += (a[e][v] == mx[v]) ? k_mx[v] : 0.f;
          o[v] = r;
        }
        vstore<VEC>(dp + (int64_t)e * H, o);
      }
    return;
  }
  const float* mp = m + (int64_t)p0 * H + c;
  for (int p = p0; p < p1; ++p, mp += H) {  // same order and roundings as k_pna_agg_fwd
    float a[VEC];
    vload<VEC>(a, mp);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      s[v] = __fadd_rn(s[v], a[v]);
      s2[v] = __fadd_rn(s2[v], __fmul_rn(a[v], a[v]));
      mn[v] = fminf(mn[v], a[v]);
      mx[v] = fmaxf(mx[v], a[v]);
    }
  }
  const float cnt = (float)(p1 - p0);
  float mean[VEC], sd[VEC], nmn[VEC], nmx[VEC], c2[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    c2[v] = 0.f;
    mean[v] = __fdiv_rn(s[v], cnt);
    float mean2 = __fdiv_rn(s2[v], cnt);
    float var = __fsub_rn(mean2, __fmul_rn(mean[v], mean[v]));
    float o = __fsqrt_rn(fmaxf(var, STD_VAR_MIN));
    sd[v] = (o <= STD_MASK_AT) ? 0.f : o;
    nmn[v] = (mn[v] == 0.f) ? 1.f : 0.f;  // torch's zero-filled self counts as a tie (see k_pna_agg_bwd)
    nmx[v] = (mx[v] == 0.f) ? 1.f : 0.f;
  }
  mp = m + (int64_t)p0 * H + c;
  for (int p = p0; p < p1; ++p, mp += H) {
    float a[VEC];
    vload<VEC>(a, mp);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      nmn[v] += (a[v] == mn[v]) ? 1.f : 0.f;
      nmx[v] += (a[v] == mx[v]) ? 1.f : 0.f;
      const float dv = a[v] - mean[v];
      c2[v] = fmaf(dv, dv, c2[v]);
    }
  }
  float k_mean[VEC], k_mn[VEC], k_mx[VEC], k_sd[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    k_mean[v] = gmean[v] / cnt;
    k_mn[v] = gmn[v] / nmn[v];
    k_mx[v] = gmx[v] / nmx[v];
    // masked or not is the forward's decision (sd, bit for bit); the divisor is the centred two-pass std, free of the
    // cancellation error of mean(x^2) - mean(x)^2 (centered = 0: the forward's value, like the CPU path's backward)
    const float sdiv = (centered && c2[v] > 0.f) ? sqrtf(c2[v] / cnt) : sd[v];
    k_sd[v] = (sd[v] > 0.f) ? gsd[v] / (cnt * sdiv) : 0.f;
  }
  mp = m + (int64_t)p0 * H + c;
  float* dp = dm + (int64_t)p0 * H + c;
  for (int p = p0; p < p1; ++p, mp += H, dp += H) {
    float a[VEC], o[VEC];
    vload<VEC>(a, mp);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float r = k_mean[v] + k_sd[v] * (a[v] - mean[v]);
      r += (a[v] == mn[v]) ? k_mn[v] : 0.f;
      r += (a[v] == mx[v]) ? k_mx[v] : 0.f;
      o[v] = r;
    }
    vstore<VEC>(dp, o);
  }
}

extern "C" int32_t gnx_pna_aggregate_bwd(gnx_handle* h, const float* dA, const float* m, const float* A,
                                         const int32_t* rowptr, int64_t N, int64_t E, int32_t T, int32_t F, float* dm) {
  GNX_CHECK_ARG(h && rowptr && T > 0 && F > 0 && N >= 0 && E >= 0, "gnx_pna_aggregate_bwd: bad argument");
  GNX_CHECK_ARG(N == 0 || (dA && A), "gnx_pna_aggregate_bwd: NULL argument");
  if (N == 0) return GNX_OK;
  // read the aggregate gradient 16NH, re-read the messages 4EH + index 4E, write the message gradient 4EH
  gnx_prof_scope prof(h, GNX_K_PNA_AGG_BWD, 16.0 * N * T * F + 8.0 * E * T * F + 4.0 * E, 0.0, 0.0, true);
  {
    if (h->opt[GNX_OPT_AGG_BWD_RECOMPUTE] != 0) {
      if (F % 4 == 0)
        GNX_LAUNCH_TIMED(prof, k_pna_agg_bwd_rc<4>, dim3((unsigned)gnx_cdiv(N * (T * F / 4), 256)), dim3(256), 0,
                         h->stream, dA, m, rowptr, N, (int)T, (int)F, dm, h->opt[GNX_OPT_STD_BWD_CENTERED],
                         h->opt[GNX_OPT_AGG_BWD_RECOMPUTE] == 1 ? 1 : 0);
      else
        GNX_LAUNCH_TIMED(prof, k_pna_agg_bwd_rc<1>, dim3((unsigned)gnx_cdiv(N * (int64_t)(T * F), 256)), dim3(256), 0,
                         h->stream, dA, m, rowptr, N, (int)T, (int)F, dm, h->opt[GNX_OPT_STD_BWD_CENTERED],
                         h->opt[GNX_OPT_AGG_BWD_RECOMPUTE] == 1 ? 1 : 0);
      GNX_LAUNCH_CHECK();
      return GNX_OK;
    }
  }
  if (F % 4 == 0) {
    int64_t th = N * (T * F / 4);
    GNX_LAUNCH_TIMED(prof, k_pna_agg_bwd<4>, dim3((unsigned)gnx_cdiv(th, 256)), dim3(256), 0, h->stream, dA, m, A, rowptr,
                     N, (int)T, (int)F, dm, h->opt[GNX_OPT_STD_BWD_CENTERED]);
  } else {
    int64_t th = N * (T * F);
    GNX_LAUNCH_TIMED(prof, k_pna_agg_bwd<1>, dim3((unsigned)gnx_cdiv(th, 256)), dim3(256), 0, h->stream, dA, m, A, rowptr,
                     N, (int)T, (int)F, dm, h->opt[GNX_OPT_STD_BWD_CENTERED]);
  }
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// PNA message assembly: h1[p,:] = relu(P[dst[p]] + Q[src[p]] + Te[code[p]])
// ---------------------------------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(256) k_edge_combine_fwd(const float* __restrict__ P, const float* __restrict__ Q,
                                                          const float* __restrict__ Te, const int* __restrict__ src,
                                                          const int* __restrict__ dst, const int* __restrict__ code,
                                                          int64_t E, int H, int relu, float* __restrict__ h1) {
  const int G = H / VEC;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= E * G) return;
  int64_t p = t / G;
  int c = (int)(t % G) * VEC;
  float a[VEC], b[VEC], e[VEC], o[VEC];
  vload<VEC>(a, P + (int64_t)dst[p] * H + c);
  vload<VEC>(b, Q + (int64_t)src[p] * H + c);
  vload<VEC>(e, Te + (int64_t)code[p] * H + c);
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    float r = (a[v] + b[v]) + e[v];
    o[v] = relu ? fmaxf(r, 0.f) : r;
  }
  vstore<VEC>(h1 + p * H + c, o);
}

extern "C" int32_t gnx_edge_combine_fwd(gnx_handle* h, const float* P, const float* Q, const float* Te,
                                        const int32_t* src, const int32_t* dst, const int32_t* code, int64_t E,
                                        int32_t H, int32_t relu, float* h1) {
  GNX_CHECK_ARG(h && H > 0 && E >= 0, "gnx_edge_combine_fwd: bad argument");
  if (E == 0) return GNX_OK;
  GNX_CHECK_ARG(P && Q && Te && src && dst && code && h1, "gnx_edge_combine_fwd: NULL argument");
  gnx_prof_scope prof(h, GNX_K_EDGE_COMBINE_FWD, 12.0 * E * H + 12.0 * E);  // gather P, Q rows, write h; 3 indices
  if (H % 4 == 0)
    hipLaunchKernelGGL(k_edge_combine_fwd<4>, dim3((unsigned)gnx_cdiv(E * (H / 4), 256)), dim3(256), 0, h->stream, P, Q,
                       Te, src, dst, code, E, (int)H, (int)relu, h1);
  else
    hipLaunchKernelGGL(k_edge_combine_fwd<1>, dim3((unsigned)gnx_cdiv(E * H, 256)), dim3(256), 0, h->stream, P, Q, Te,
                       src, dst, code, E, (int)H, (int)relu, h1);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

// dP[i] = sum over CSR row i of g ; dQ[j] = sum over cpos list of j of g
template <int VEC>
__global__ void __launch_bounds__(256) k_edge_combine_bwd(const float* __restrict__ g, const int* __restrict__ rowptr,
                                                          const int* __restrict__ colptr, const int* __restrict__ cpos,
                                                          int64_t N, int H, float* __restrict__ dP,
                                                          float* __restrict__ dQ) {
  const int G = H / VEC;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * G) return;
  int64_t n = t / G;
  int c = (int)(t % G) * VEC;
  float s[VEC], q[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    s[v] = 0.f;
    q[v] = 0.f;
  }
  if (dP != nullptr) {  // (NULL: the destination sums were already formed by gnx_pna_edge_bwd)
    int p0 = rowptr[n], p1 = rowptr[n + 1];
    for (int p = p0; p < p1; ++p) {
      float a[VEC];
      vload<VEC>(a, g + (int64_t)p * H + c);
#pragma unroll
      for (int v = 0; v < VEC; ++v) s[v] += a[v];
    }
  }
  int c0 = colptr[n], c1 = colptr[n + 1];
  for (int k = c0; k < c1; ++k) {
    float a[VEC];
    vload<VEC>(a, g + (int64_t)cpos[k] * H + c);
#pragma unroll
    for (int v = 0; v < VEC; ++v) q[v] += a[v];
  }
  if (dP != nullptr) vstore<VEC>(dP + n * H + c, s);
  vstore<VEC>(dQ + n * H + c, q);
}

extern "C" int32_t gnx_edge_combine_bwd(gnx_handle* h, const float* g, const int32_t* rowptr, const int32_t* colptr,
                                        const int32_t* cpos, const int32_t* code, int64_t N, int64_t E, int32_t H,
                                        int32_t R, float* dP, float* dQ, float* dTe, void* ws, size_t ws_bytes) {
  GNX_CHECK_ARG(h && H > 0 && N >= 0 && E >= 0, "gnx_edge_combine_bwd: bad argument");
  if (N == 0) return GNX_OK;
  GNX_CHECK_ARG(rowptr && colptr && dQ && (E == 0 || (g && cpos)), "gnx_edge_combine_bwd: NULL argument");
  // read g once, write dP, dQ (dP == NULL: only the by-source sums dQ)
  gnx_prof_scope prof(h, GNX_K_EDGE_COMBINE_BWD, 4.0 * E * H + (dP ? 8.0 : 4.0) * N * H + 4.0 * E + 8.0 * N);
  if (H % 4 == 0)
    hipLaunchKernelGGL(k_edge_combine_bwd<4>, dim3((unsigned)gnx_cdiv(N * (H / 4), 256)), dim3(256), 0, h->stream, g,
                       rowptr, colptr, cpos, N, (int)H, dP, dQ);
  else
    hipLaunchKernelGGL(k_edge_combine_bwd<1>, dim3((unsigned)gnx_cdiv(N * H, 256)), dim3(256), 0, h->stream, g, rowptr,
                       colptr, cpos, N, (int)H, dP, dQ);
  GNX_LAUNCH_CHECK();
  if (dTe != nullptr && E > 0) {
    GNX_CHECK_ARG(code && R > 0, "gnx_edge_combine_bwd: code/R missing for dTe");
    return gnx_code_scatter_add(h, code, E, R, g, H, dTe, ws, ws_bytes);
  }
  return GNX_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// dTable[r,:] += sum over the items of key r of g[item,:]  through the inverted index (pos grouped by key, ptr).
// One workgroup = SEG_CHUNK consecutive entries of pos; a thread owns 4 channels and sums rows in registers, flushing
// with one atomic per (key run inside the chunk): ~ (#chunks + #keys) x H atomics in total instead of items x H.
// ---------------------------------------------------------------------------------------------------------------
#define SEG_CHUNK 128

template <int VEC>
__global__ void __launch_bounds__(256) k_key_segment_sum(const float* __restrict__ g, const int* __restrict__ pos,
                                                         const int* __restrict__ key, int64_t E, int H,
                                                         float* __restrict__ dtable) {
  const int G = H / VEC;                 // threads per row
  const int lanes = 256 / G > 0 ? 256 / G : 1;  // row lanes per block
  const int cg = threadIdx.x % G, rl = threadIdx.x / G;
  if (rl >= lanes) return;
  const int c = cg * VEC;
  const int64_t i0 = (int64_t)blockIdx.x * SEG_CHUNK;
  int64_t i1 = i0 + SEG_CHUNK;
  if (i1 > E) i1 = E;
  // each row lane walks a contiguous sub-run so that key changes are rare inside it
  const int64_t per = (i1 - i0 + lanes - 1) / lanes;
  int64_t a = i0 + (int64_t)rl * per, b = a + per;
  if (b > i1) b = i1;
  if (a >= b) return;
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  int cur = key[pos[a]];
  constexpr int UN = 4;  // items in flight: index loads batched, then the row gathers batched (2 round trips per 4).
  // (64-item chunks with 8 in flight were measured 2x SLOWER: 118 vs 56 us at cfg-2 -- twice the atomic flushes.)
  for (int64_t i = a; i < b; i += UN) {
    int p[UN], k[UN];
    float r[UN][VEC];
#pragma unroll
    for (int j = 0; j < UN; ++j) p[j] = pos[(i + j < b) ? i + j : b - 1];
#pragma unroll
    for (int j = 0; j < UN; ++j) k[j] = key[p[j]];
#pragma unroll
    for (int j = 0; j < UN; ++j) vload<VEC>(r[j], g + (int64_t)p[j] * H + c);
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      if (i + j < b) {
        if (k[j] != cur) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) atomicAdd(&dtable[(int64_t)cur * H + c + v], acc[v]);
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
          cur = k[j];
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += r[j][v];
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) atomicAdd(&dtable[(int64_t)cur * H + c + v], acc[v]);
}

extern "C" int32_t gnx_key_segment_sum(gnx_handle* h, const float* g, const int32_t* pos, const int32_t* key, int64_t E,
                                       int32_t H, float* dtable) {
  GNX_CHECK_ARG(h && H > 0 && E >= 0, "gnx_key_segment_sum: bad argument");
  if (E == 0) return GNX_OK;
  GNX_CHECK_ARG(g && pos && key && dtable, "gnx_key_segment_sum: NULL argument");
  gnx_prof_scope prof(h, GNX_K_KEY_SEGMENT_SUM, 4.0 * E * H + 8.0 * E);
  const unsigned blocks = (unsigned)gnx_cdiv(E, SEG_CHUNK);
  if (H % 4 == 0 && H / 4 <= 256)
    hipLaunchKernelGGL(k_key_segment_sum<4>, dim3(blocks), dim3(256), 0, h->stream, g, pos, key, E, (int)H, dtable);
  else if (H <= 256)
    hipLaunchKernelGGL(k_key_segment_sum<1>, dim3(blocks), dim3(256), 0, h->stream, g, pos, key, E, (int)H, dtable);
  else {
    gnx_set_error("gnx_key_segment_sum: H=%d not supported (H %% 4 == 0 and H <= 1024, or H <= 256)", H);
    return GNX_E_INVALID;
  }
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// GINE: out[i] = (1+eps) x[i] + sum_{p in row i} relu(x[src[p]] + Le[code[p]])
// ---------------------------------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(256) k_gine_fwd(const float* __restrict__ x, const float* __restrict__ Le,
                                                  const int* __restrict__ rowptr, const int* __restrict__ src,
                                                  const int* __restrict__ code, int64_t N, int H, float eps,
                                                  float* __restrict__ out) {
  const int G = H / VEC;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * G) return;
  int64_t n = t / G;
  int c = (int)(t % G) * VEC;
  float s[VEC], xi[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) s[v] = 0.f;
  int p0 = rowptr[n], p1 = rowptr[n + 1];
  for (int p = p0; p < p1; ++p) {
    float a[VEC], e[VEC];
    vload<VEC>(a, x + (int64_t)src[p] * H + c);
    vload<VEC>(e, Le + (int64_t)code[p] * H + c);
#pragma unroll
    for (int v = 0; v < VEC; ++v) s[v] += fmaxf(a[v] + e[v], 0.f);
  }
  vload<VEC>(xi, x + n * H + c);
  const float k = 1.0f + eps;
#pragma unroll
  for (int v = 0; v < VEC; ++v) s[v] = s[v] + k * xi[v];
  vstore<VEC>(out + n * H + c, s);
}

extern "C" int32_t gnx_gine_aggregate_fwd(gnx_handle* h, const float* x, const float* Le, const int32_t* rowptr,
                                          const int32_t* src, const int32_t* code, int64_t N, int64_t E, int32_t H,
                                          float eps, float* out) {
  GNX_CHECK_ARG(h && H > 0 && N >= 0 && E >= 0, "gnx_gine_aggregate_fwd: bad argument");
  if (N == 0) return GNX_OK;
  GNX_CHECK_ARG(x && Le && rowptr && out, "gnx_gine_aggregate_fwd: NULL argument");
  // gather x[src] per edge 4EH + two indices 8E, read x and write out 8NH
  gnx_prof_scope prof(h, GNX_K_GINE_AGG_FWD, 4.0 * E * H + 8.0 * E + 8.0 * N * H, 0.0, 0.0, true);
  if (H % 4 == 0)
    GNX_LAUNCH_TIMED(prof, k_gine_fwd<4>, dim3((unsigned)gnx_cdiv(N * (H / 4), 256)), dim3(256), 0, h->stream, x, Le,
                     rowptr, src, code, N, (int)H, eps, out);
  else
    GNX_LAUNCH_TIMED(prof, k_gine_fwd<1>, dim3((unsigned)gnx_cdiv(N * H, 256)), dim3(256), 0, h->stream, x, Le, rowptr,
                     src, code, N, (int)H, eps, out);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

// dx[j] = (1+eps) dout[j] + sum_{p in cpos(j)} dout[dst[p]] * (x[j] + Le[code[p]] > 0)
// gm[p] (the masked message gradient, needed for dLe) is written to a caller-free scratch only when dLe is wanted:
// here it is accumulated straight into an LDS-privatised table by a second kernel over edges.
template <int VEC>
__global__ void __launch_bounds__(256) k_gine_bwd_dx(const float* __restrict__ dout, const float* __restrict__ x,
                                                     const float* __restrict__ Le, const int* __restrict__ colptr,
                                                     const int* __restrict__ cpos, const int* __restrict__ dst,
                                                     const int* __restrict__ code, int64_t N, int H, float eps,
                                                     float* __restrict__ dx) {
  const int G = H / VEC;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * G) return;
  int64_t n = t / G;
  int c = (int)(t % G) * VEC;
  float xj[VEC], s[VEC], dj[VEC];
  vload<VEC>(xj, x + n * H + c);
  vload<VEC>(dj, dout + n * H + c);
  const float k = 1.0f + eps;
#pragma unroll
  for (int v = 0; v < VEC; ++v) s[v] = k * dj[v];
  int c0 = colptr[n], c1 = colptr[n + 1];
  for (int q = c0; q < c1; ++q) {
    int p = cpos[q];
    float d[VEC], e[VEC];
    vload<VEC>(d, dout + (int64_t)dst[p] * H + c);
    vload<VEC>(e, Le + (int64_t)code[p] * H + c);
#pragma unroll
    for (int v = 0; v < VEC; ++v) s[v] += (xj[v] + e[v] > 0.f) ? d[v] : 0.f;
  }
  vstore<VEC>(dx + n * H + c, s);
}

// dLe[code] += dout[dst[p]] * (x[src[p]] + Le[code[p]] > 0): LDS-privatised over edges (column slabs of CW)
__global__ void __launch_bounds__(256) k_gine_bwd_dle(const float* __restrict__ dout, const float* __restrict__ x,
                                                      const float* __restrict__ Le, const int* __restrict__ src,
                                                      const int* __restrict__ dst, const int* __restrict__ code,
                                                      int64_t E, int H, int R, int CW, int64_t rows_per_block,
                                                      float* __restrict__ dLe) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < R * CW; i += 256) lds[i] = 0.f;
  __syncthreads();
  const int c = tid % CW, rl = tid / CW, RL = 256 / CW;
  const int col = blockIdx.y * CW + c;
  int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > E) r1 = E;
  if (col < H) {
    for (int64_t p = r0 + rl; p < r1; p += RL) {
      int cd = code[p];
      float pre = x[(int64_t)src[p] * H + col] + Le[(int64_t)cd * H + col];
      if (pre > 0.f) atomicAdd(&lds[cd * CW + c], dout[(int64_t)dst[p] * H + col]);
    }
  }
  __syncthreads();
  for (int i = tid; i < R * CW; i += 256) {
    int r = i / CW, cc = blockIdx.y * CW + (i % CW);
    float v = lds[i];
    if (cc < H && v != 0.f) atomicAdd(&dLe[(int64_t)r * H + cc], v);
  }
}

// dLe through the inverted index (CSR positions stably grouped by bond code, as for the PNA bond-table gradient):
// a workgroup walks SEG_CHUNK consecutive entries; a thread owns 4 channels, recomputes the ReLU mask from
// x[src] + Le[code] and sums dout[dst] rows in registers, flushing one atomic per key run.  The LDS-privatised kernel
// above issues one LDS atomic per (edge, channel) at ~2 clk per lane-op: 1.0 ms per layer at cfg-3.
template <int VEC>
__global__ void __launch_bounds__(256) k_gine_dle_segment_sum(const float* __restrict__ dout, const float* __restrict__ x,
                                                              const float* __restrict__ Le, const int* __restrict__ pos,
                                                              const int* __restrict__ src, const int* __restrict__ dst,
                                                              const int* __restrict__ key, int64_t E, int H,
                                                              float* __restrict__ dLe) {
  const int G = H / VEC;
  const int lanes = 256 / G > 0 ? 256 / G : 1;
  const int cg = threadIdx.x % G, rl = threadIdx.x / G;
  if (rl >= lanes) return;
  const int c = cg * VEC;
  const int64_t i0 = (int64_t)blockIdx.x * SEG_CHUNK;
  int64_t i1 = i0 + SEG_CHUNK;
  if (i1 > E) i1 = E;
  const int64_t per = (i1 - i0 + lanes - 1) / lanes;
  int64_t a = i0 + (int64_t)rl * per, b = a + per;
  if (b > i1) b = i1;
  if (a >= b) return;
  float acc[VEC], le[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  int cur = key[pos[a]];
  vload<VEC>(le, Le + (int64_t)cur * H + c);
  constexpr int UN = 4;  // items in flight: index loads batched, then the row gathers batched
  for (int64_t i = a; i < b; i += UN) {
    int p[UN], k[UN], sj[UN], dj[UN];
    float rx[UN][VEC], rd[UN][VEC];
#pragma unroll
    for (int j = 0; j < UN; ++j) p[j] = pos[(i + j < b) ? i + j : b - 1];
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      k[j] = key[p[j]];
      sj[j] = src[p[j]];
      dj[j] = dst[p[j]];
    }
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      vload<VEC>(rx[j], x + (int64_t)sj[j] * H + c);
      vload<VEC>(rd[j], dout + (int64_t)dj[j] * H + c);
    }
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      if (i + j < b) {
        if (k[j] != cur) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) atomicAdd(&dLe[(int64_t)cur * H + c + v], acc[v]);
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
          cur = k[j];
          vload<VEC>(le, Le + (int64_t)cur * H + c);
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += (rx[j][v] + le[v] > 0.f) ? rd[j][v] : 0.f;
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) atomicAdd(&dLe[(int64_t)cur * H + c + v], acc[v]);
}

static int32_t gine_dle(gnx_handle* h, const float* dout, const float* x, const float* Le, const int32_t* src,
                        const int32_t* dst, const int32_t* code, int64_t E, int32_t H, int32_t R, float* dLe) {
  int CW = 64;
  while ((size_t)R * CW * sizeof(float) > 64 * 1024 && CW > 8) CW >>= 1;
  GNX_CHECK_ARG((size_t)R * CW * sizeof(float) <= 64 * 1024, "gnx_gine_aggregate_bwd: %d rows do not fit the LDS tile", R);
  int slabs = (int)gnx_cdiv(H, CW);
  int64_t rows_per_block = gnx_cdiv(E, gnx_cdiv(1024, slabs));
  if (rows_per_block < 256) rows_per_block = 256;
  hipLaunchKernelGGL(k_gine_bwd_dle, dim3((unsigned)gnx_cdiv(E, rows_per_block), (unsigned)slabs), dim3(256),
                     (size_t)R * CW * sizeof(float), h->stream, dout, x, Le, src, dst, code, E, (int)H, (int)R, CW,
                     rows_per_block, dLe);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

extern "C" int32_t gnx_gine_aggregate_bwd(gnx_handle* h, const float* dout, const float* x, const float* Le,
                                          const int32_t* colptr, const int32_t* cpos, const int32_t* src,
                                          const int32_t* dst, const int32_t* code, const int32_t* code_pos,
                                          int64_t N, int64_t E, int32_t H, int32_t R, float eps, float* dx, float* dLe) {
  GNX_CHECK_ARG(h && H > 0 && N >= 0 && E >= 0, "gnx_gine_aggregate_bwd: bad argument");
  if (N == 0) return GNX_OK;
  GNX_CHECK_ARG(dout && x && Le && colptr && dx, "gnx_gine_aggregate_bwd: NULL argument");
  GNX_CHECK_ARG(E == 0 || (cpos && src && dst && code), "gnx_gine_aggregate_bwd: NULL edge array with E>0");
  // dx: gather dout[dst] per edge 4EH + indices 8E, read dout and x, write dx 12NH; dLe: another 8EH of gathers
  // (one launch when the bond-table gradient is computed elsewhere: its events are then attached to the dispatch)
  gnx_prof_scope prof(h, GNX_K_GINE_AGG_BWD, 4.0 * E * H + 8.0 * E + 12.0 * N * H + (dLe ? 8.0 * E * H : 0.0), 0.0, 0.0,
                      dLe == nullptr || E == 0);
  if (H % 4 == 0)
    GNX_LAUNCH_TIMED(prof, k_gine_bwd_dx<4>, dim3((unsigned)gnx_cdiv(N * (H / 4), 256)), dim3(256), 0, h->stream, dout, x,
                     Le, colptr, cpos, dst, code, N, (int)H, eps, dx);
  else
    GNX_LAUNCH_TIMED(prof, k_gine_bwd_dx<1>, dim3((unsigned)gnx_cdiv(N * H, 256)), dim3(256), 0, h->stream, dout, x, Le,
                     colptr, cpos, dst, code, N, (int)H, eps, dx);
  GNX_LAUNCH_CHECK();
  if (dLe != nullptr && E > 0) {
    GNX_CHECK_ARG(R > 0, "gnx_gine_aggregate_bwd: R <= 0");
    if (code_pos != nullptr && H % 4 == 0 && H / 4 <= 256) {
      hipLaunchKernelGGL(k_gine_dle_segment_sum<4>, dim3((unsigned)gnx_cdiv(E, SEG_CHUNK)), dim3(256), 0, h->stream, dout,
                         x, Le, code_pos, src, dst, code, E, (int)H, dLe);
      GNX_LAUNCH_CHECK();
      return GNX_OK;
    }
    return gine_dle(h, dout, x, Le, src, dst, code, E, H, R, dLe);
  }
  return GNX_OK;
}

// dLe alone (the bond-table gradient of GINEConv.lin's output): same kernels as the tail of gnx_gine_aggregate_bwd, as
// its own entry point so that the caller can run it on a side stream while dx continues on the main one.
extern "C" int32_t gnx_gine_dle(gnx_handle* h, const float* dout, const float* x, const float* Le, const int32_t* src,
                                const int32_t* dst, const int32_t* code, const int32_t* code_pos, int64_t E, int32_t H,
                                int32_t R, float* dLe) {
  GNX_CHECK_ARG(h && H > 0 && E >= 0 && R > 0, "gnx_gine_dle: bad argument");
  if (E == 0) return GNX_OK;
  GNX_CHECK_ARG(dout && x && Le && src && dst && code && dLe, "gnx_gine_dle: NULL argument");
  gnx_prof_scope prof(h, GNX_K_KEY_SEGMENT_SUM, 8.0 * E * H + 16.0 * E);
  if (code_pos != nullptr && H % 4 == 0 && H / 4 <= 256) {
    hipLaunchKernelGGL(k_gine_dle_segment_sum<4>, dim3((unsigned)gnx_cdiv(E, SEG_CHUNK)), dim3(256), 0, h->stream, dout, x,
                       Le, code_pos, src, dst, code, E, (int)H, dLe);
    GNX_LAUNCH_CHECK();
    return GNX_OK;
  }
  return gine_dle(h, dout, x, Le, src, dst, code, E, H, R, dLe);
}

// ---------------------------------------------------------------------------------------------------------------
// contiguous segment pool over graph_ptr (global pool): add / mean / max
// ---------------------------------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(256) k_pool_fwd(const float* __restrict__ x, const int* __restrict__ ptr, int64_t B,
                                                  int H, int mode, float* __restrict__ out) {
  const int G = H / VEC;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * G) return;
  int64_t b = t / G;
  int c = (int)(t % G) * VEC;
  int p0 = ptr[b], p1 = ptr[b + 1];
  float s[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) s[v] = (mode == GNX_POOL_MAX) ? -INFINITY : 0.f;
  int p = p0;
  for (; p + 3 < p1; p += 4) {  // four rows in flight (a graph's ~20 atoms were 20 serial round trips); same order
    float a[4][VEC];
#pragma unroll
    for (int j = 0; j < 4; ++j) vload<VEC>(a[j], x + (int64_t)(p + j) * H + c);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int v = 0; v < VEC; ++v) s[v] = (mode == GNX_POOL_MAX) ? fmaxf(s[v], a[j][v]) : __fadd_rn(s[v], a[j][v]);
  }
  for (; p < p1; ++p) {
    float a[VEC];
    vload<VEC>(a, x + (int64_t)p * H + c);
#pragma unroll
    for (int v = 0; v < VEC; ++v) s[v] = (mode == GNX_POOL_MAX) ? fmaxf(s[v], a[v]) : __fadd_rn(s[v], a[v]);
  }
  int d = p1 - p0;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    if (mode == GNX_POOL_MEAN) s[v] = __fdiv_rn(s[v], (float)(d > 0 ? d : 1));
    if (mode == GNX_POOL_MAX && d == 0) s[v] = 0.f;
  }
  vstore<VEC>(out + b * H + c, s);
}

template <int VEC>
__global__ void __launch_bounds__(256) k_pool_bwd(const float* __restrict__ dout, const float* __restrict__ x,
                                                  const float* __restrict__ out, const int* __restrict__ ptr, int64_t B,
                                                  int H, int mode, float* __restrict__ dx) {
  const int G = H / VEC;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * G) return;
  int64_t b = t / G;
  int c = (int)(t % G) * VEC;
  int p0 = ptr[b], p1 = ptr[b + 1];
  if (p1 <= p0) return;
  float g[VEC];
  vload<VEC>(g, dout + b * H + c);
  if (mode == GNX_POOL_MAX) {
    float mx[VEC], nt[VEC];
    vload<VEC>(mx, out + b * H + c);
#pragma unroll
    for (int v = 0; v < VEC; ++v) nt[v] = (mx[v] == 0.f) ? 1.f : 0.f;  // zero-filled self counts as a tie (see PNA bwd)
    for (int p = p0; p < p1; ++p) {
      float a[VEC];
      vload<VEC>(a, x + (int64_t)p * H + c);
#pragma unroll
      for (int v = 0; v < VEC; ++v) nt[v] += (a[v] == mx[v]) ? 1.f : 0.f;
    }
    for (int p = p0; p < p1; ++p) {
      float a[VEC], o[VEC];
      vload<VEC>(a, x + (int64_t)p * H + c);
#pragma unroll
      for (int v = 0; v < VEC; ++v) o[v] = (a[v] == mx[v]) ? g[v] / nt[v] : 0.f;
      vstore<VEC>(dx + (int64_t)p * H + c, o);
    }
  } else {
    if (mode == GNX_POOL_MEAN) {
      float cnt = (float)(p1 - p0);
#pragma unroll
      for (int v = 0; v < VEC; ++v) g[v] = g[v] / cnt;
    }
    for (int p = p0; p < p1; ++p) vstore<VEC>(dx + (int64_t)p * H + c, g);
  }
}

extern "C" int32_t gnx_segment_pool_fwd(gnx_handle* h, const float* x, const int32_t* ptr, int64_t B, int32_t H,
                                        int32_t mode, float* out) {
  GNX_CHECK_ARG(h && H > 0 && B >= 0 && mode >= 0 && mode <= 2, "gnx_segment_pool_fwd: bad argument");
  if (B == 0) return GNX_OK;
  GNX_CHECK_ARG(ptr && out, "gnx_segment_pool_fwd: NULL argument");
  if (H % 4 == 0)
    hipLaunchKernelGGL(k_pool_fwd<4>, dim3((unsigned)gnx_cdiv(B * (H / 4), 256)), dim3(256), 0, h->stream, x, ptr, B,
                       (int)H, (int)mode, out);
  else
    hipLaunchKernelGGL(k_pool_fwd<1>, dim3((unsigned)gnx_cdiv(B * H, 256)), dim3(256), 0, h->stream, x, ptr, B, (int)H,
                       (int)mode, out);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

extern "C" int32_t gnx_segment_pool_bwd(gnx_handle* h, const float* dout, const float* x, const float* out,
                                        const int32_t* ptr, int64_t B, int32_t H, int32_t mode, float* dx) {
  GNX_CHECK_ARG(h && H > 0 && B >= 0 && mode >= 0 && mode <= 2, "gnx_segment_pool_bwd: bad argument");
  if (B == 0) return GNX_OK;
  GNX_CHECK_ARG(dout && ptr && dx && (mode != GNX_POOL_MAX || (x && out)), "gnx_segment_pool_bwd: NULL argument");
  if (H % 4 == 0)
    hipLaunchKernelGGL(k_pool_bwd<4>, dim3((unsigned)gnx_cdiv(B * (H / 4), 256)), dim3(256), 0, h->stream, dout, x, out,
                       ptr, B, (int)H, (int)mode, dx);
  else
    hipLaunchKernelGGL(k_pool_bwd<1>, dim3((unsigned)gnx_cdiv(B * H, 256)), dim3(256), 0, h->stream, dout, x, out, ptr, B,
                       (int)H, (int)mode, dx);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

// Fused PNA edge pipeline: message assembly -> pre-layer 1 -> mean|min|max|std aggregate in ONE kernel per layer.
//
// Reference semantics: [3P] torch_geometric.nn.PNAConv.message + DegreeScalerAggregation as built at
// /root/reference/gnnepcsaft/train/models.py:445-457 with pre_layers = 2 (configs/default.py:44):
//     h1[p] = relu(P[dst[p]] + Q[src[p]] + Te[code[p]])      (pre-layer 0 folded to node level, gnx_edge_combine_fwd)
//     m[p]  = h1[p] W1^T + b1                                  (pre-layer 1, no activation: the last pre layer)
//     A[n]  = [mean | min | max | std] over the CSR row of n   (gnx_pna_aggregate_fwd)
// The unfused launch sequence moves h1 and m through HBM twice each (edge_combine 213 MB -> k_gemm_ws3 158 MB ->
// k_pna_agg_fwd 253 MB per cfg-2 layer by PMC).  Here a persistent workgroup walks EDGE TILES of whole destination
// rows: it gathers the three operand rows straight into the split-bf16 LDS images of the product (no h1 read), multiplies
// by the weights it holds in registers (the k_gemm_ws3 body), parks the fp32 result tile in LDS and reduces the CSR rows it
// already holds -- m is written once (backward needs it), never read back.
//
// Bit-exactness: h1 is computed with k_edge_combine_fwd's expression, the product with k_gemm_ws3's MFMA sequence and
// accumulator order, the aggregate with k_pna_agg_fwd's operation order over the same rows, so h1 / m / A are bit-identical
// to the unfused path's (tests/test_fused_gpu.py).
//
// Measured and rejected (round 3, same box, cfg-2's layer shape, 102 vs 100 us): a ROLE-SPECIALISED form of this kernel --
// waves 0-3 only multiply (64 rows x 32 columns each) and park the result tile, waves 4-7 gather / split the next tile and
// reduce / store the previous one, every common-path store unconditional (clamped duplicates) so that hipcc can count the
// stores behind an in-flight gather.  It was bit-identical and no faster: the phases of a tile are latency-bound, not
// throughput-bound (one wave per SIMD issues a dependent VALU chain at half the rate of two), so giving the reduction to
// half the waves doubled its length.  In-kernel stamps of THIS kernel (tools/ubench/edge_fwd_stamp.hip): issue of the next
// tile's 25 loads 19 %, MFMAs 15 %, result tile -> LDS 4 %, gather wait + split 11 %, stores + reduction 26 %, barriers 25 %.
//
// Also measured and rejected: issuing the next gather right after the image barrier, AHEAD of the reduction's stores (so
// that it does not queue behind them): 91 vs 99 us without the h1 / m stores, but 121 vs 102 us with them -- the stores then
// queue behind the gather and the next wait for the gather drains them too.
//
// Edge tiles (gnx_edge_tiles): tile j holds every node whose FIRST CSR position lies in [W j, W (j + 1)); with in-degrees
// <= maxdeg and W = 65 - maxdeg a tile never exceeds 64 message rows.  A violated bound (a degree above the hint) sets
// sticky range-flag bit 6 and the overflowing rows are dropped (no out-of-bounds access).
#include "gnx_common.hpp"
#include "gnx_split.hpp"

#include <cmath>

#define STD_VAR_MIN 1e-5f
#define STD_MASK_AT 0.0031622776601683794f

#define EF_BM 64
#define EF_LDC 132                            // floats per row of the fp32 message tile (33 x 16 B: odd)
#define EF_A_BYTES (2 * W3_BUF)               // two stages of three bf16 images [64][128 (+8)]
#define EF_C_BYTES (EF_BM * EF_LDC * 4)
#define EF_RP 128                             // CSR bounds of a tile's first 128 nodes are staged in LDS (two stages)
#define EF_RP_BYTES (2 * (EF_RP + 4) * 4)
#define EF_LDS (EF_A_BYTES + EF_C_BYTES + EF_RP_BYTES)  // 139 296 B: one workgroup per CU

struct edge_fwd_args {
  const float* P;
  const float* Q;
  const float* Te;
  const int* src;
  const int* dst;
  const int* code;
  const int* rowptr;
  const int* tile_info;  // (first node, first CSR position) per tile, ntiles + 1 entries
  int ntiles, T, F, H;
  int64_t N, E;
  const float* W[GNX_PNA_MAX_TOWERS];  // pre-layer 1 weight [F, F] ([out, in]) per tower
  const float* b[GNX_PNA_MAX_TOWERS];
  float* h1;  // [E, H] or NULL
  float* m;   // [E, H] or NULL
  float* A;   // [N, T, 4F]
  int* flag;
#ifdef EF_STAMP
  unsigned long long* stamps;  // diagnostic build only (tools/ubench/edge_fwd_stamp.hip): 8 phase sums per workgroup
#endif
};

// phase stamps of the diagnostic build: cycles since the previous stamp are added to phase `i` (thread 0 stores the sums)
#ifdef EF_STAMP
#define EF_AT(i)                                     \
  do {                                               \
    __builtin_amdgcn_sched_barrier(0);               \
    const unsigned long long tn_ = clock64();        \
    tacc[i] += tn_ - tprev;                          \
    tprev = tn_;                                     \
    __builtin_amdgcn_sched_barrier(0);               \
  } while (0)
#else
#define EF_AT(i)
#endif

__global__ void k_edge_tiles(const int* __restrict__ rowptr, int64_t N, int64_t E, int W, int ntiles,
                             int* __restrict__ info) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j > ntiles) return;
  if (j == ntiles) {
    info[2 * j] = (int)N;
    info[2 * j + 1] = (int)E;
    return;
  }
  const int64_t target = (int64_t)W * j;  // <= E by the choice of ntiles
  int lo = 0, hi = (int)N;                // first n in [0, N] with rowptr[n] >= target (rowptr[N] = E)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (rowptr[mid] >= target) hi = mid;
    else lo = mid + 1;
  }
  info[2 * j] = lo;
  info[2 * j + 1] = rowptr[lo];
}

extern "C" int32_t gnx_edge_tiles_count(int64_t E, int32_t tile_w) { return tile_w > 0 ? (int32_t)(E / tile_w) + 1 : 0; }

extern "C" int32_t gnx_edge_tiles(gnx_handle* h, const int32_t* rowptr, int64_t N, int64_t E, int32_t tile_w,
                                  int32_t* tile_info) {
  GNX_CHECK_ARG(h && rowptr && tile_info && N >= 0 && E >= 0 && tile_w >= 1 && tile_w <= EF_BM,
                "gnx_edge_tiles: bad argument (tile_w must be in [1, 64])");
  const int ntiles = gnx_edge_tiles_count(E, tile_w);
  hipLaunchKernelGGL(k_edge_tiles, dim3((unsigned)gnx_cdiv(ntiles + 1, 256)), dim3(256), 0, h->stream, rowptr, N, E,
                     (int)tile_w, ntiles, tile_info);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

#ifdef EF_STAMP
static unsigned long long* ef_stamp_buf = nullptr;
#endif

__global__ void __launch_bounds__(512, 1) k_pna_edge_fwd(edge_fwd_args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* Cs = reinterpret_cast<float*>(lds + EF_A_BYTES);
  int* rpl = reinterpret_cast<int*>(lds + EF_A_BYTES + EF_C_BYTES);  // [2][EF_RP + 4]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = (wave >> 2) * 32, wc = (wave & 3) * 32;
  const int li = lane & 31, lh = lane >> 5;
  const int F = g.F, H = g.H, T = g.T;
  const int nslab = (F + 15) >> 4;
  const int gc = wc + li;
  const int tw = blockIdx.x % T;       // this workgroup's tower (the grid is a multiple of T)
  const int tstride = gridDim.x / T;
  const int coff = tw * F;

  // loader / aggregation mapping: 32 lanes cover one 512-B row with consecutive float4, 16 rows per pass
  const int ar = tid >> 5, ak = (tid & 31) * 4;
  const bool ak_ok = ak < F;
  const int akc = ak_ok ? ak : 0;
  const int2* tinfo = reinterpret_cast<const int2*>(g.tile_info);
  const int last = g.ntiles - 1;
  const int Em1 = (int)(g.E - 1);

  struct bounds {
    int n0, e0, n1, e1;
  };
  // Tile bounds come out of a lane vector, not out of memory: lane l holds (n0, e0, n1, e1) of the workgroup's l-th tile
  // counted from `vbase` (refilled every 32 tiles), and v_readlane hands tile i's bounds to the scalar unit.  A uniform
  // global load here compiles to a VECTOR load + readfirstlane behind s_waitcnt vmcnt(0), i.e. every iteration would wait
  // for the gather it has just issued and for the previous tile's stores.
  int vb_n0, vb_e0, vb_n1, vb_e1;
  auto fill_bounds = [&](int jbase) {
    const int jt = jbase + lane * tstride;
    const int jj = jt < last ? jt : last;  // clamped: tiles past the end are never processed, their loads stay in range
    const int2 a = tinfo[jj], c = tinfo[jj + 1];
    vb_n0 = a.x;
    vb_e0 = a.y;
    vb_n1 = c.x;
    vb_e1 = c.y;
  };
  auto bounds_at = [&](int rel) {  // rel: wave-uniform, < 64
    bounds b = {__builtin_amdgcn_readlane(vb_n0, rel), __builtin_amdgcn_readlane(vb_e0, rel),
                __builtin_amdgcn_readlane(vb_n1, rel), __builtin_amdgcn_readlane(vb_e1, rel)};
    return b;
  };
  auto count_of = [&](const bounds& b) {
    const int c = b.e1 - b.e0;
    return c < EF_BM ? c : EF_BM;
  };

  // dst, src, code of rows ar + 16 i of a tile
  auto load_idx = [&](const bounds& b, int (&ix)[4][3]) {
    const int cm1 = count_of(b) - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = ar + 16 * i;
      int e = b.e0 + (r < cm1 ? r : (cm1 > 0 ? cm1 : 0));
      e = e < Em1 ? e : Em1;
      ix[i][0] = g.dst[e];
      ix[i][1] = g.src[e];
      ix[i][2] = g.code[e];
    }
  };
  int idx[4][3];  // ... of the tile whose gather is issued next
  f32x4 ga[4][3];
  auto issue_gather = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ga[i][0] = *reinterpret_cast<const f32x4*>(g.P + (int64_t)idx[i][0] * H + coff + akc);
      ga[i][1] = *reinterpret_cast<const f32x4*>(g.Q + (int64_t)idx[i][1] * H + coff + akc);
      ga[i][2] = *reinterpret_cast<const f32x4*>(g.Te + (int64_t)idx[i][2] * H + coff + akc);
    }
  };
  // CSR bounds of the tile's nodes: rowptr[n0 .. n0 + EF_RP] go through LDS (thread t < EF_RP + 1 carries entry t), so that
  // the reduction below waits on no vector-memory counter
  auto load_rp = [&](const bounds& b) {
    const int64_t node = (int64_t)b.n0 + (tid <= EF_RP ? tid : EF_RP);
    return g.rowptr[node < g.N ? node : g.N];
  };
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  // gathered rows -> h1 (k_edge_combine_fwd's expression) -> global h1 + three bf16 images of LDS stage `buf`
  auto consume_gather = [&](const bounds& b, unsigned char* buf) {
    const int cnt = count_of(b);
    f32x4 hv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = ak_ok && (ar + 16 * i) < cnt;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float r = (ga[i][0][j] + ga[i][1][j]) + ga[i][2][j];
        hv[i][j] = ok ? fmaxf(r, 0.f) : 0.f;
      }
      if (ok && g.h1 != nullptr)
        *reinterpret_cast<f32x4*>(g.h1 + (int64_t)(b.e0 + ar + 16 * i) * H + coff + ak) = hv[i];
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {  // rows ar + 32 hh and ar + 32 hh + 16: one split3 of 8 values
      float x[8];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) x[4 * q + j] = hv[2 * hh + q][j];
      bf16x8 pc[3];
      split3(x, pc[0], pc[1], pc[2]);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(&pc[p]);
        unsigned char* q = buf + p * W3_PIECE + (ar + 32 * hh) * W3_LDB + ak * 2;
        *reinterpret_cast<f32x2*>(q) = f32x2{w.x, w.y};
        *reinterpret_cast<f32x2*>(q + 16 * W3_LDB) = f32x2{w.z, w.w};
      }
    }
  };

  const float bv = (gc < F) ? g.b[tw][gc] : 0.f;
  int j = blockIdx.x / T;  // grid <= ntiles * T
  fill_bounds(j);
  int rel = 0;  // position of tile j in the lane vector
  bounds B0 = bounds_at(0), B1 = bounds_at(1), B2 = bounds_at(2);
  load_idx(B0, idx);
  issue_gather();  // in flight while the weight fragments are fetched and split
  if (tid <= EF_RP) rpl[tid] = load_rp(B0);

  // ---- this wave's pre-layer-1 weight fragments, split once: lane holds column gc, k = 16 s + 8 lh + jj
  bf16x8 b1[8], b2[8], b3[8];
  {
    const float* Wt = g.W[tw];
    const bool n_ok = gc < F;
    const int gcc = n_ok ? gc : 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float x[8];
      const int k0 = 16 * s + 8 * lh;
      const int ka = (k0 < F) ? k0 : 0, kb = (k0 + 4 < F) ? k0 + 4 : 0;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(Wt + (int64_t)gcc * F + ka);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(Wt + (int64_t)gcc * F + kb);
      const bool ok0 = n_ok && k0 < F, ok1 = n_ok && k0 + 4 < F;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        x[jj] = ok0 ? v0[jj] : 0.f;
        x[4 + jj] = ok1 ? v1[jj] : 0.f;
      }
      split3(x, b1[s], b2[s], b3[s]);
    }
  }
  consume_gather(B0, lds);
  load_idx(B1, idx);
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(idx[i][0]), "v"(idx[i][1]), "v"(idx[i][2]));  // arrived before the loop
  __syncthreads();

  int cur = 0;
#ifdef EF_STAMP
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = clock64();
#endif
  while (j < g.ntiles) {
    const int j1 = j + tstride;
    const bool has1 = j1 < g.ntiles;
    // loads of the next tiles, oldest first: CSR bounds of tile j1, indices of tile j + 2, then the gather of tile j1 (it
    // stays in flight under the MFMAs; the two small loads ahead of it are touched right after them)
    const int rpv = load_rp(B1);
    int idn[4][3];
    load_idx(B2, idn);  // (clamped bounds past the end: harmless loads)
    __builtin_amdgcn_sched_barrier(0);  // keep the small loads AHEAD of the gather in the vmcnt queue
    issue_gather();  // unconditional (past the end: clamped, harmless loads) so that hipcc can COUNT the loads behind idn
    const bounds B3 = bounds_at(rel + 3);
    if (tid == 0 && B0.e1 - B0.e0 > EF_BM) atomicOr(g.flag, 64);
    EF_AT(0);  // issue of the next tile's loads

    f32x16 acc, corr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[r] = 0.f;
      corr[r] = 0.f;
    }
    const unsigned char* ap = lds + cur * W3_BUF + (wr + li) * W3_LDB + 16 * lh;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < nslab) {
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(ap + 32 * s);
        const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(ap + 32 * s + W3_PIECE);
        const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(ap + 32 * s + 2 * W3_PIECE);
        corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3[s], corr, 0, 0, 0);
        corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1[s], corr, 0, 0, 0);
        corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2[s], corr, 0, 0, 0);
        corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2[s], corr, 0, 0, 0);
        corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1[s], corr, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1[s], acc, 0, 0, 0);
      }
    }
    EF_AT(1);  // fragment reads + MFMAs
    // gfx950 counts loads AND stores in one in-order vmcnt: a load result first touched behind the tile's (data-dependent
    // number of) stores makes hipcc wait vmcnt(0), i.e. for every store of the tile to be acknowledged (measured: 35 % of
    // the kernel's time).  Everything loaded at the top is touched HERE, where only the gather is outstanding behind it.
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(idn[i][0]), "v"(idn[i][1]), "v"(idn[i][2]));
    asm volatile("" ::"v"(rpv));
    __syncthreads();  // every wave has finished reducing the previous tile out of Cs
    EF_AT(2);
    {
      float* cw = Cs + (wr + 4 * lh) * EF_LDC + gc;
#pragma unroll
      for (int r = 0; r < 16; ++r) cw[((r & 3) + 8 * (r >> 2)) * EF_LDC] = (acc[r] + corr[r]) + bv;
    }
    if (tid <= EF_RP) rpl[(cur ^ 1) * (EF_RP + 4) + tid] = rpv;
    EF_AT(3);  // result tile -> LDS
    if (has1) consume_gather(B1, lds + (cur ^ 1) * W3_BUF);
    EF_AT(4);  // wait for the gather, h1, split, LDS stores
    __syncthreads();
    EF_AT(5);

    // ---- the tile's messages: written once, reduced per destination row in k_pna_agg_fwd's operation order
    {
      const int cnt = count_of(B0);
      if (ak_ok) {
        if (g.m != nullptr) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = ar + 16 * i;
            if (r < cnt)
              *reinterpret_cast<f32x4*>(g.m + (int64_t)(B0.e0 + r) * H + coff + ak) =
                  *reinterpret_cast<const f32x4*>(Cs + r * EF_LDC + ak);
          }
        }
        // one destination row: reduce message rows [p0, p1) (CSR positions) out of the LDS tile
        auto reduce_node = [&](int node, int p0, int p1) {
          const int d = p1 - p0;
          p0 -= B0.e0;
          p1 -= B0.e0;
          p1 = p1 < cnt ? p1 : cnt;
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, s2 = s;
          f32x4 mn = {INFINITY, INFINITY, INFINITY, INFINITY}, mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
          int p = p0;
#ifndef EF_NO_AGG_UNROLL
          for (; p + 1 < p1; p += 2) {  // two rows in flight (same operation order as one by one)
            const f32x4 a = *reinterpret_cast<const f32x4*>(Cs + p * EF_LDC + ak);
            const f32x4 b = *reinterpret_cast<const f32x4*>(Cs + (p + 1) * EF_LDC + ak);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              s[v] = __fadd_rn(__fadd_rn(s[v], a[v]), b[v]);
              s2[v] = __fadd_rn(__fadd_rn(s2[v], __fmul_rn(a[v], a[v])), __fmul_rn(b[v], b[v]));
              mn[v] = fminf(mn[v], fminf(a[v], b[v]));
              mx[v] = fmaxf(mx[v], fmaxf(a[v], b[v]));
            }
          }
#endif
          for (; p < p1; ++p) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(Cs + p * EF_LDC + ak);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              s[v] = __fadd_rn(s[v], a[v]);
              s2[v] = __fadd_rn(s2[v], __fmul_rn(a[v], a[v]));
              mn[v] = fminf(mn[v], a[v]);
              mx[v] = fmaxf(mx[v], a[v]);
            }
          }
          const float cntf = (float)(d > 0 ? d : 1);
          f32x4 mean, sd;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            mean[v] = __fdiv_rn(s[v], cntf);
            const float mean2 = __fdiv_rn(s2[v], cntf);
            const float var = __fsub_rn(mean2, __fmul_rn(mean[v], mean[v]));
            const float o = __fsqrt_rn(fmaxf(var, STD_VAR_MIN));
            sd[v] = (o <= STD_MASK_AT) ? 0.f : o;
            if (d == 0) {
              mn[v] = 0.f;
              mx[v] = 0.f;
            }
          }
          float* o = g.A + ((int64_t)node * T + tw) * (int64_t)(4 * F) + ak;
          *reinterpret_cast<f32x4*>(o) = mean;
          *reinterpret_cast<f32x4*>(o + F) = mn;
          *reinterpret_cast<f32x4*>(o + 2 * F) = mx;
          *reinterpret_cast<f32x4*>(o + 3 * F) = sd;
        };
        const int* rpc = rpl + cur * (EF_RP + 4);
        const int nn = B0.n1 - B0.n0;  // nodes of the tile (> EF_RP only with long runs of edge-less nodes)
        const int nl = nn < EF_RP ? nn : EF_RP;
        for (int ln = ar; ln < nl; ln += 16) reduce_node(B0.n0 + ln, rpc[ln], rpc[ln + 1]);  // no vector-memory wait here
        if (nn > EF_RP)  // wave-uniform, rare: the remaining nodes read their CSR bounds from global memory
          for (int ln = EF_RP + ar; ln < nn; ln += 16) reduce_node(B0.n0 + ln, g.rowptr[B0.n0 + ln], g.rowptr[B0.n0 + ln + 1]);
      }
    }
    EF_AT(6);  // message stores + aggregate
    // rotate the pipeline
    B0 = B1;
    B1 = B2;
    B2 = B3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      idx[i][0] = idn[i][0];
      idx[i][1] = idn[i][1];
      idx[i][2] = idn[i][2];
    }
    cur ^= 1;
    j = j1;
    if (++rel == 32) {  // refill the bounds vector from the tile that is now current (one full wait per 32 tiles)
      fill_bounds(j);
      asm volatile("" ::"v"(vb_n0), "v"(vb_e0), "v"(vb_n1), "v"(vb_e1));  // waited for here, not at the loop head
      rel = 0;
    }
  }
#ifdef EF_STAMP
  if (tid == 0 && g.stamps != nullptr)
    for (int i = 0; i < 8; ++i) g.stamps[blockIdx.x * 8 + i] = tacc[i];
#endif
}

// messages h1 W1^T + b1 and their aggregate from P / Q / Te (see the header of this file).  h1 / m may be NULL (not kept:
// inference).  W1 / b1: HOST arrays of T device pointers.  tile_info from gnx_edge_tiles(tile_w); F % 4 == 0, F <= 128.
extern "C" int32_t gnx_pna_edge_fwd(gnx_handle* h, const float* P, const float* Q, const float* Te, const int32_t* src,
                                    const int32_t* dst, const int32_t* code, const int32_t* rowptr,
                                    const int32_t* tile_info, int32_t tile_w, int64_t N, int64_t E, int32_t T, int32_t F,
                                    const float* const* W1, const float* const* b1, float* h1, float* m, float* A) {
  GNX_CHECK_ARG(h && T >= 1 && T <= GNX_PNA_MAX_TOWERS && F >= 4 && F <= 128 && F % 4 == 0 && N >= 0 && E >= 0,
                "gnx_pna_edge_fwd: bad shape T=%d F=%d (need F %% 4 == 0, F <= 128)", T, F);
  if (N == 0) return GNX_OK;
  GNX_CHECK_ARG(A && rowptr && W1 && b1, "gnx_pna_edge_fwd: NULL argument");
  if (E == 0) return gnx_fill(h, A, N * (int64_t)T * 4 * F, 0.f);  // empty rows: mean = min = max = std = 0
  GNX_CHECK_ARG(P && Q && Te && src && dst && code && tile_info && tile_w >= 1 && tile_w <= EF_BM,
                "gnx_pna_edge_fwd: NULL argument or bad tile width %d", tile_w);
  edge_fwd_args g;
  g.P = P;
  g.Q = Q;
  g.Te = Te;
  g.src = src;
  g.dst = dst;
  g.code = code;
  g.rowptr = rowptr;
  g.tile_info = tile_info;
  g.ntiles = gnx_edge_tiles_count(E, tile_w);
  g.T = T;
  g.F = F;
  g.H = T * F;
  g.N = N;
  g.E = E;
  for (int t = 0; t < GNX_PNA_MAX_TOWERS; ++t) {
    g.W[t] = t < T ? W1[t] : nullptr;
    g.b[t] = t < T ? b1[t] : nullptr;
    GNX_CHECK_ARG(t >= T || (g.W[t] && g.b[t]), "gnx_pna_edge_fwd: W1[%d] / b1[%d] is NULL", t, t);
  }
  g.h1 = h1;
  g.m = m;
  g.A = A;
  g.flag = h->d_flag;
#ifdef EF_STAMP
  g.stamps = ef_stamp_buf;
#endif
  static bool attr_set = false;
  if (!attr_set) {
    GNX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pna_edge_fwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(160 * 1024)));
    attr_set = true;
  }
  const double H = (double)T * F;
  // own algorithmic bytes: P and Q rows once (8NH), indices + CSR (16E + 4N), A (16NH), h1 / m if kept (4EH each)
  const double bytes = 24.0 * N * H + 16.0 * E + 4.0 * N + 4.0 * E * H * ((h1 ? 1 : 0) + (m ? 1 : 0));
  const double flops = 2.0 * E * (double)F * F * T;
  gnx_prof_scope prof(h, GNX_K_PNA_EDGE_FWD, bytes, flops, 6.0 * flops, true);
  int grid = h->num_cus > 0 ? h->num_cus : 256;
  const int64_t want = (int64_t)g.ntiles * T;
  if (grid > want) grid = (int)want;
  grid = grid / T * T;
  if (grid < T) grid = T;
  GNX_LAUNCH_TIMED(prof, k_pna_edge_fwd, dim3((unsigned)grid), dim3(512), (size_t)EF_LDS, h->stream, g);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Fused PNA edge pipeline, backward: pre-layer 1's input gradient with the ReLU mask of pre-layer 0, the destination
// sums and the bond-table gradient in ONE pass over the message gradient ("one read of the message
// gradient in backward" -- the three-launch sequence read it in the masked product, then re-read the product's result for
// dP (k_edge_combine_bwd) and again, gathered by bond code, for dTe (k_key_segment_sum)):
//     gh1[p] = (ge[p] W1_t) * (h1[p] > 0)          k_gemm_ws3<NN, mask>'s arithmetic (bit-identical)
//     dP[n]  = sum over the CSR row of n of gh1    k_edge_combine_bwd's order (bit-identical)
//     dTe[c] += sum over the positions with bond code c of gh1
// dQ (the sum over the edges LEAVING a node) is a gather through the by-source index and stays a pass of its own over gh1.
// The bond-code sums ride on the matrix pipe: the masked result block of a wave (32 rows x 32 columns, column on the lane,
// rows in the accumulator registers) is exactly the B operand of a product that sums over its ROW index, so
// dTe_block += onehot(code)[64 x 32 rows] * gh1_block with the one-hot operand generated in registers from the tile's codes
// and gh1 as three exact bf16 pieces: 12 MFMAs per wave and tile beside the 48 of the product, exact 0/1 products, fp32
// accumulators kept for the whole launch and flushed with one atomic add per (code, column) and workgroup.
// ---------------------------------------------------------------------------------------------------------------
#define EB_MASK_BYTES (2 * EF_BM * 32)  // ReLU mask of the h1 tile: one nibble-carrying byte per (row, 4 columns), two stages
#define EB_CODE_BYTES (2 * EF_BM * 4)
#define EB_LDS (EF_A_BYTES + EF_C_BYTES + EF_RP_BYTES + EB_MASK_BYTES + EB_CODE_BYTES)

struct edge_bwd_args {
  const float* ge;   // [E, H] message gradient (CSR order)
  const float* h1;   // [E, H] pre-layer 0's activation (mask)
  const int* code;
  const int* rowptr;
  const int* tile_info;
  int ntiles, T, F, H, R;
  int64_t N, E;
  const float* W[GNX_PNA_MAX_TOWERS];  // pre-layer 1 weight [F, F] ([out, in])
  float* gh1;  // [E, H]
  float* dP;   // [N, H]
  float* dTe;  // [R, H], accumulated
  int* flag;
};

__global__ void __launch_bounds__(512, 1) k_pna_edge_bwd(edge_bwd_args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* Cs = reinterpret_cast<float*>(lds + EF_A_BYTES);
  int* rpl = reinterpret_cast<int*>(lds + EF_A_BYTES + EF_C_BYTES);
  unsigned char* mk = lds + EF_A_BYTES + EF_C_BYTES + EF_RP_BYTES;                       // [2][64][32]
  int* cds = reinterpret_cast<int*>(lds + EF_A_BYTES + EF_C_BYTES + EF_RP_BYTES + EB_MASK_BYTES);  // [2][64]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = (wave >> 2) * 32, wc = (wave & 3) * 32;
  const int li = lane & 31, lh = lane >> 5;
  const int F = g.F, H = g.H, T = g.T;
  const int nslab = (F + 15) >> 4;
  const int gc = wc + li;
  const int tw = blockIdx.x % T;
  const int tstride = gridDim.x / T;
  const int coff = tw * F;
  const int ar = tid >> 5, ak = (tid & 31) * 4;
  const bool ak_ok = ak < F;
  const int akc = ak_ok ? ak : 0;
  const int2* tinfo = reinterpret_cast<const int2*>(g.tile_info);
  const int last = g.ntiles - 1;
  const int Em1 = (int)(g.E - 1);

  struct bounds {
    int n0, e0, n1, e1;
  };
  int vb_n0, vb_e0, vb_n1, vb_e1;  // lane l: bounds of the workgroup's l-th tile from the refill point (see k_pna_edge_fwd)
  auto fill_bounds = [&](int jbase) {
    const int jt = jbase + lane * tstride;
    const int jj = jt < last ? jt : last;
    const int2 a = tinfo[jj], c = tinfo[jj + 1];
    vb_n0 = a.x;
    vb_e0 = a.y;
    vb_n1 = c.x;
    vb_e1 = c.y;
  };
  auto bounds_at = [&](int rel) {
    bounds b = {__builtin_amdgcn_readlane(vb_n0, rel), __builtin_amdgcn_readlane(vb_e0, rel),
                __builtin_amdgcn_readlane(vb_n1, rel), __builtin_amdgcn_readlane(vb_e1, rel)};
    return b;
  };
  auto count_of = [&](const bounds& b) {
    const int c = b.e1 - b.e0;
    return c < EF_BM ? c : EF_BM;
  };

  // tile loads: 4 rows x float4 of ge and of h1 per thread (32 lanes cover one 512-B row), the tile's bond codes and CSR slice
  f32x4 rg[4], rh[4];
  int rcode = 0, rpv = 0;
  auto load_tile = [&](const bounds& b) {
    const int cm1 = count_of(b) - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = ar + 16 * i;
      int e = b.e0 + (r < cm1 ? r : (cm1 > 0 ? cm1 : 0));  // rows past the count: clamped (zeroed at LDS-store time)
      e = e < Em1 ? e : Em1;
      rg[i] = *reinterpret_cast<const f32x4*>(g.ge + (int64_t)e * H + coff + akc);
      rh[i] = *reinterpret_cast<const f32x4*>(g.h1 + (int64_t)e * H + coff + akc);
    }
    {
      const int r = tid & 63;
      int e = b.e0 + (r < cm1 ? r : (cm1 > 0 ? cm1 : 0));
      e = e < Em1 ? e : Em1;
      rcode = g.code[e];
      const int64_t node = (int64_t)b.n0 + (tid <= EF_RP ? tid : EF_RP);
      rpv = g.rowptr[node < g.N ? node : g.N];
    }
  };
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  auto store_tile = [&](const bounds& b, int st) {
    const int cnt = count_of(b);
    unsigned char* buf = lds + st * W3_BUF;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {  // rows ar + 32 hh and ar + 32 hh + 16: one split3 of 8 values
      float x[8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const bool ok = ak_ok && (ar + 32 * hh + 16 * q) < cnt;
#pragma unroll
        for (int c = 0; c < 4; ++c) x[4 * q + c] = ok ? rg[2 * hh + q][c] : 0.f;
      }
      bf16x8 pc[3];
      split3(x, pc[0], pc[1], pc[2]);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(&pc[p]);
        unsigned char* q = buf + p * W3_PIECE + (ar + 32 * hh) * W3_LDB + ak * 2;
        *reinterpret_cast<f32x2*>(q) = f32x2{w.x, w.y};
        *reinterpret_cast<f32x2*>(q + 16 * W3_LDB) = f32x2{w.z, w.w};
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // ReLU mask of pre-layer 0: bit c of the byte = h1[row][4 (tid & 31) + c] > 0
      const unsigned bits = (rh[i][0] > 0.f ? 1u : 0u) | (rh[i][1] > 0.f ? 2u : 0u) | (rh[i][2] > 0.f ? 4u : 0u) |
                            (rh[i][3] > 0.f ? 8u : 0u);
      mk[(st * EF_BM + ar + 16 * i) * 32 + (tid & 31)] = (unsigned char)(ak_ok ? bits : 0u);
    }
    if (tid < EF_BM) cds[st * EF_BM + tid] = rcode;
    if (tid <= EF_RP) rpl[st * (EF_RP + 4) + tid] = rpv;
  };

  int j = blockIdx.x / T;
  fill_bounds(j);
  int rel = 0;
  bounds B0 = bounds_at(0), B1 = bounds_at(1);
  load_tile(B0);

  // ---- this wave's weight fragments for the input-gradient orientation: B(k, n) = W1[k][n] (k = pre-layer 1's output
  //      index, n = its input index): the zero-filled 128 x 128 image is staged in LDS once (k_gemm_ws3<NN>'s scheme)
  bf16x8 b1[8], b2[8], b3[8];
  {
    float* wl = reinterpret_cast<float*>(lds);
    const float* Wt = g.W[tw];
    const int r = tid >> 5, c4 = (tid & 31) * 4;
    const bool c_ok = c4 < F;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rr = r + 16 * i;
      const f32x4 v = *reinterpret_cast<const f32x4*>(Wt + (int64_t)(rr < F ? rr : 0) * F + (c_ok ? c4 : 0));
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(wl + rr * 128 + c4) = (c_ok && rr < F) ? v : z;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float x[8];
      const int k0 = 16 * s + 8 * lh;
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = wl[(k0 + q) * 128 + gc];
      split3(x, b1[s], b2[s], b3[s]);
    }
    // consumed HERE: hipcc may otherwise sink the LDS reads above to below the barrier, where they race with the other waves'
    // stores of the first tile (seen in k_gemm_ws3<NN>'s pipelined form)
#pragma unroll
    for (int s = 0; s < 8; ++s) asm volatile("" ::"v"(b1[s]), "v"(b2[s]), "v"(b3[s]));
    __syncthreads();  // the image is overwritten by the first tile
  }
  store_tile(B0, 0);
  __syncthreads();

  f32x16 Y0, Y1;  // bond-table gradient of this wave's columns: codes 0..31 / 32..63 (C/D layout: code on the register index)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    Y0[r] = 0.f;
    Y1[r] = 0.f;
  }
  int cur = 0;
  while (j < g.ntiles) {
    const int j1 = j + tstride;
    const bool has1 = j1 < g.ntiles;
    load_tile(B1);  // tile j1 (clamped bounds past the end: harmless loads); in flight under the MFMAs
    if (tid == 0 && B0.e1 - B0.e0 > EF_BM) atomicOr(g.flag, 64);

    f32x16 acc, corr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[r] = 0.f;
      corr[r] = 0.f;
    }
    const unsigned char* ap = lds + cur * W3_BUF + (wr + li) * W3_LDB + 16 * lh;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < nslab) {
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(ap + 32 * s);
        const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(ap + 32 * s + W3_PIECE);
        const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(ap + 32 * s + 2 * W3_PIECE);
        corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3[s], corr, 0, 0, 0);
        corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1[s], corr, 0, 0, 0);
        corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2[s], corr, 0, 0, 0);
        corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2[s], corr, 0, 0, 0);
        corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1[s], corr, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1[s], acc, 0, 0, 0);
      }
    }
    // ---- ReLU mask, then the masked block as the B operand of the one-hot product (it sums over the block's ROW index)
    {
      const unsigned char* mrow = mk + (cur * EF_BM + wr + 4 * lh) * 32 + (gc >> 2);
      const int bit = gc & 3;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned m = mrow[((r & 3) + 8 * (r >> 2)) * 32];
        const float v = acc[r] + corr[r];
        acc[r] = ((m >> bit) & 1u) ? v : 0.f;
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float x[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = acc[8 * s + q];
      bf16x8 p1, p2, p3;
      split3(x, p1, p2, p3);
      // element q of lane half lh is block row 16 s + 8 (q >> 2) + 4 lh + (q & 3): its bond code decides the one-hot entry
      const int* cp = cds + cur * EF_BM + wr + 16 * s + 4 * lh;
      const int c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3], c4 = cp[8], c5 = cp[9], c6 = cp[10], c7 = cp[11];
      const int cc[8] = {c0, c1, c2, c3, c4, c5, c6, c7};
      split_u32x4 o0, o1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        o0[q] = (cc[2 * q] == li ? 0x3F80u : 0u) | (cc[2 * q + 1] == li ? 0x3F800000u : 0u);
        o1[q] = (cc[2 * q] == li + 32 ? 0x3F80u : 0u) | (cc[2 * q + 1] == li + 32 ? 0x3F800000u : 0u);
      }
      const bf16x8 h0 = __builtin_bit_cast(bf16x8, o0), h1v = __builtin_bit_cast(bf16x8, o1);
      Y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h0, p1, Y0, 0, 0, 0);
      Y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1v, p1, Y1, 0, 0, 0);
      Y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h0, p2, Y0, 0, 0, 0);
      Y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1v, p2, Y1, 0, 0, 0);
      Y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h0, p3, Y0, 0, 0, 0);
      Y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1v, p3, Y1, 0, 0, 0);
    }
    __syncthreads();  // every wave has finished reading the previous tile out of Cs
    {
      float* cw = Cs + (wr + 4 * lh) * EF_LDC + gc;
#pragma unroll
      for (int r = 0; r < 16; ++r) cw[((r & 3) + 8 * (r >> 2)) * EF_LDC] = acc[r];
    }
    if (has1) store_tile(B1, cur ^ 1);
    __syncthreads();

    // ---- the masked gradient rows: written once; their sum per destination row (k_edge_combine_bwd's order) -> dP
    {
      const int cnt = count_of(B0);
      if (ak_ok) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = ar + 16 * i;
          if (r < cnt)
            *reinterpret_cast<f32x4*>(g.gh1 + (int64_t)(B0.e0 + r) * H + coff + ak) =
                *reinterpret_cast<const f32x4*>(Cs + r * EF_LDC + ak);
        }
        auto sum_node = [&](int node, int p0, int p1) {
          p0 -= B0.e0;
          p1 -= B0.e0;
          p1 = p1 < cnt ? p1 : cnt;
          f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
          for (int p = p0; p < p1; ++p) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(Cs + p * EF_LDC + ak);
#pragma unroll
            for (int v = 0; v < 4; ++v) sacc[v] += a[v];
          }
          *reinterpret_cast<f32x4*>(g.dP + (int64_t)node * H + coff + ak) = sacc;
        };
        const int* rpc = rpl + cur * (EF_RP + 4);
        const int nn = B0.n1 - B0.n0;
        const int nl = nn < EF_RP ? nn : EF_RP;
        for (int ln = ar; ln < nl; ln += 16) sum_node(B0.n0 + ln, rpc[ln], rpc[ln + 1]);
        if (nn > EF_RP)
          for (int ln = EF_RP + ar; ln < nn; ln += 16) sum_node(B0.n0 + ln, g.rowptr[B0.n0 + ln], g.rowptr[B0.n0 + ln + 1]);
      }
    }
    B0 = B1;
    B1 = bounds_at(rel + 2);
    cur ^= 1;
    j = j1;
    if (++rel == 32) {
      fill_bounds(j);
      asm volatile("" ::"v"(vb_n0), "v"(vb_e0), "v"(vb_n1), "v"(vb_e1));
      rel = 0;
      B1 = bounds_at(1);
    }
  }
  // ---- flush the bond-table sums: register r of lane (li, lh) holds code (r & 3) + 8 (r >> 2) + 4 lh (+ 32), column gc
  if (gc < F) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (c < g.R) atomicAdd(g.dTe + (int64_t)c * H + coff + gc, Y0[r]);
      if (c + 32 < g.R) atomicAdd(g.dTe + (int64_t)(c + 32) * H + coff + gc, Y1[r]);
    }
  }
}

// gh1 = (ge W1) * (h1 > 0), dP = its sums over CSR rows, dTe += its sums by bond code (see above).  W1: HOST array of T device
// pointers.  R <= 64 bond codes; F % 4 == 0, F <= 128; tile_info from gnx_edge_tiles(tile_w); float operands 16-byte aligned.
extern "C" int32_t gnx_pna_edge_bwd(gnx_handle* h, const float* ge, const float* h1, const int32_t* code,
                                    const int32_t* rowptr, const int32_t* tile_info, int32_t tile_w, int64_t N, int64_t E,
                                    int32_t T, int32_t F, int32_t R, const float* const* W1, float* gh1, float* dP,
                                    float* dTe) {
  GNX_CHECK_ARG(h && T >= 1 && T <= GNX_PNA_MAX_TOWERS && F >= 4 && F <= 128 && F % 4 == 0 && R >= 1 && R <= 64 && N >= 0 &&
                    E >= 0, "gnx_pna_edge_bwd: bad shape T=%d F=%d R=%d (need F %% 4 == 0, F <= 128, R <= 64)", T, F, R);
  if (N == 0) return GNX_OK;
  GNX_CHECK_ARG(dP && rowptr && W1, "gnx_pna_edge_bwd: NULL argument");
  if (E == 0) return gnx_fill(h, dP, N * (int64_t)T * F, 0.f);
  GNX_CHECK_ARG(ge && h1 && code && tile_info && gh1 && dTe && tile_w >= 1 && tile_w <= EF_BM,
                "gnx_pna_edge_bwd: NULL argument or bad tile width %d", tile_w);
  edge_bwd_args g;
  g.ge = ge;
  g.h1 = h1;
  g.code = code;
  g.rowptr = rowptr;
  g.tile_info = tile_info;
  g.ntiles = gnx_edge_tiles_count(E, tile_w);
  g.T = T;
  g.F = F;
  g.H = T * F;
  g.R = R;
  g.N = N;
  g.E = E;
  for (int t = 0; t < GNX_PNA_MAX_TOWERS; ++t) {
    g.W[t] = t < T ? W1[t] : nullptr;
    GNX_CHECK_ARG(t >= T || g.W[t], "gnx_pna_edge_bwd: W1[%d] is NULL", t);
  }
  g.gh1 = gh1;
  g.dP = dP;
  g.dTe = dTe;
  g.flag = h->d_flag;
  static bool attr_set = false;
  if (!attr_set) {
    GNX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pna_edge_bwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(160 * 1024)));
    attr_set = true;
  }
  const double Hd = (double)T * F;
  // ge and h1 read once, gh1 and dP written once, codes + CSR
  const double bytes = 12.0 * E * Hd + 4.0 * N * Hd + 4.0 * E + 4.0 * N;
  const double flops = 2.0 * E * (double)F * F * T;
  gnx_prof_scope prof(h, GNX_K_PNA_EDGE_BWD, bytes, flops, 6.0 * flops + 3.0 * 2.0 * E * 64.0 * F * T, true);
  int grid = h->num_cus > 0 ? h->num_cus : 256;
  const int64_t want = (int64_t)g.ntiles * T;
  if (grid > want) grid = (int)want;
  grid = grid / T * T;
  if (grid < T) grid = T;
  GNX_LAUNCH_TIMED(prof, k_pna_edge_bwd, dim3((unsigned)grid), dim3(512), (size_t)EB_LDS, h->stream, g);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

// AtomEncoder / BondEncoder [3P ogb]: sum of K embedding rows, and the embedding_dense_backward scatter-add into
// tiny tables (174 / 13 / 60 rows) done in LDS-privatised form instead of 10^5 contending global atomics.
#include "gnx_common.hpp"

struct offs_t {
  int o[18];
};

template <int VEC>
__global__ void __launch_bounds__(256) k_embed_fwd(const int64_t* __restrict__ idx, int64_t N, int K, offs_t offs,
                                                   const float* __restrict__ table, int H, float* __restrict__ out,
                                                   int* __restrict__ flag) {
  const int G = H / VEC;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * G) return;
  int64_t n = t / G;
  int c = (int)(t % G) * VEC;
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  // four features at a time: their indices, then their table rows, are loaded together (one dependent index -> row pair
  // per feature was 18 serial round trips for the 9 atom features); summed in feature order as before
  for (int k0 = 0; k0 < K; k0 += 4) {
    int64_t f[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = idx[n * K + (k0 + j < K ? k0 + j : K - 1)];
    const float* row[4];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + j < K ? k0 + j : K - 1;
      const int rows = offs.o[k + 1] - offs.o[k];
      if (f[j] < 0 || f[j] >= rows) {
        bad = bad || (k0 + j < K);
        f[j] = 0;
      }
      row[j] = table + (int64_t)(offs.o[k] + (int)f[j]) * H + c;
    }
    if (bad && c == 0) atomicOr(flag, 16);
    if constexpr (VEC == 4) {
      f32x4 r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = *reinterpret_cast<const f32x4*>(row[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (k0 + j < K) {
          acc[0] += r[j].x;
          acc[1] += r[j].y;
          acc[2] += r[j].z;
          acc[3] += r[j].w;
        }
    } else {
      float r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = row[j][0];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (k0 + j < K) acc[0] += r[j];
    }
  }
  float* o = out + n * H + c;
  if constexpr (VEC == 4) {
    f32x4 r = {acc[0], acc[1], acc[2], acc[3]};
    *reinterpret_cast<f32x4*>(o) = r;
  } else {
    o[0] = acc[0];
  }
}

extern "C" int32_t gnx_embed_sum_fwd(gnx_handle* h, const int64_t* idx, int64_t N, int32_t K, const int32_t* offsets,
                                     const float* table, int32_t H, float* out) {
  GNX_CHECK_ARG(h && offsets && table && K > 0 && K <= 16 && H > 0 && N >= 0, "gnx_embed_sum_fwd: bad argument");
  GNX_CHECK_ARG(N == 0 || (idx && out), "gnx_embed_sum_fwd: NULL array with N>0");
  if (N == 0) return GNX_OK;
  gnx_prof_scope prof(h, GNX_K_EMBED, 8.0 * N * K + 4.0 * N * H);
  offs_t o;
  for (int k = 0; k <= 17; ++k) o.o[k] = offsets[k <= K ? k : K];
  if (H % 4 == 0) {
    int64_t threads = N * (H / 4);
    hipLaunchKernelGGL(k_embed_fwd<4>, dim3((unsigned)gnx_cdiv(threads, 256)), dim3(256), 0, h->stream, idx, N, (int)K,
                       o, table, (int)H, out, h->d_flag);
  } else {
    int64_t threads = N * H;
    hipLaunchKernelGGL(k_embed_fwd<1>, dim3((unsigned)gnx_cdiv(threads, 256)), dim3(256), 0, h->stream, idx, N, (int)K,
                       o, table, (int)H, out, h->d_flag);
  }
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// dtable[R, H] += sum over rows n, features k of dout[n, :] at row offs[k]+idx[n,k].
// grid = (row chunks, column slabs of CW).  Block: 256 threads = (256/CW) row lanes x CW columns.
// LDS table R x CW accumulated with ds_add_f32, flushed with one global atomic per touched element.
// ---------------------------------------------------------------------------------------------------------------
template <typename IdxT, int VEC>
__global__ void __launch_bounds__(256) k_table_scatter_add(const IdxT* __restrict__ idx, int64_t N, int K, offs_t offs,
                                                           int R, const float* __restrict__ dout, int H, int CW,
                                                           int64_t rows_per_block, float* __restrict__ dtable,
                                                           float* __restrict__ part) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < R * CW; i += 256) lds[i] = 0.f;
  __syncthreads();
  const int G = CW / VEC;        // threads per row
  const int cg = tid % G;
  const int rl = tid / G;
  const int RL = 256 / G;        // rows in flight per pass
  const int c = cg * VEC;        // column inside the slab
  const int col = blockIdx.y * CW + c;
  int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > N) r1 = N;
  if (col < H && r0 < r1) {
    // four rows in flight per thread; every load is unconditional (clamped row) and issued before any use, so one
    // round trip serves 4 rows x (1 gradient quad + 1 index) instead of a dependent chain per row
    constexpr int UN = (VEC == 1) ? 8 : 4;
    for (int64_t n = r0 + rl; n < r1; n += UN * RL) {
      int64_t nn[UN];
      bool ok[UN];
      float g[UN][VEC];
#pragma unroll
      for (int j = 0; j < UN; ++j) {
        const int64_t row = n + (int64_t)j * RL;
        ok[j] = row < r1;
        nn[j] = ok[j] ? row : r0;
      }
#pragma unroll
      for (int j = 0; j < UN; ++j) {
        if constexpr (VEC == 4) {
          f32x4 v = *reinterpret_cast<const f32x4*>(dout + nn[j] * H + col);
          g[j][0] = v.x;
          g[j][1] = v.y;
          g[j][2] = v.z;
          g[j][3] = v.w;
        } else {
          g[j][0] = dout[nn[j] * H + col];
        }
      }
      for (int k = 0; k < K; ++k) {
        const int rows = offs.o[k + 1] - offs.o[k];
        int64_t f[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) f[j] = (int64_t)idx[nn[j] * K + k];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
          if (ok[j] && f[j] >= 0 && f[j] < rows) {  // out-of-range indices were flagged in forward
            float* d = &lds[(offs.o[k] + (int)f[j]) * CW + c];
#pragma unroll
            for (int v = 0; v < VEC; ++v) atomicAdd(d + v, g[j][v]);
          }
        }
      }
    }
  }
  __syncthreads();
  if (part != nullptr) {
    // two-stage reduction: every block stores its private table; k_table_reduce folds them.  (Hundreds of blocks
    // atomically adding into the same few KB serialise at the memory-side atomic units: 100+ us for a 60-row table.)
    float* dst = part + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * (int64_t)R * CW;
    for (int i = tid; i < R * CW; i += 256) dst[i] = lds[i];
    return;
  }
  for (int i = tid; i < R * CW; i += 256) {
    int r = i / CW, cc = blockIdx.y * CW + (i % CW);
    float v = lds[i];
    if (cc < H && v != 0.f) atomicAdd(&dtable[(int64_t)r * H + cc], v);
  }
}

// dtable[r, col] += sum over chunks of part[chunk][slab][r][c]; grid = (elements / 256, chunk groups); each thread folds
// its chunk group in order and issues one atomic (<= 16 per address in total).
__global__ void __launch_bounds__(256) k_table_reduce(const float* __restrict__ part, int nchunks, int slabs, int R,
                                                      int CW, int H, int chunks_per_group, float* __restrict__ dtable) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;  // element of the [R, slabs*CW] padded table
  const int W = slabs * CW;
  if (e >= (int64_t)R * W) return;
  const int r = (int)(e / W), col = (int)(e % W);
  if (col >= H) return;
  const int slab = col / CW, c = col % CW;
  int c0 = blockIdx.y * chunks_per_group, c1 = c0 + chunks_per_group;
  if (c1 > nchunks) c1 = nchunks;
  float acc = 0.f;
  for (int ch = c0; ch < c1; ++ch) acc += part[(((int64_t)ch * slabs + slab) * R + r) * CW + c];
  if (acc != 0.f) atomicAdd(&dtable[(int64_t)r * H + col], acc);
}

static void table_scatter_geometry(int64_t N, int R, int H, int* CW, int* slabs, int64_t* rows_per_block,
                                   int64_t* chunks) {
  int cw = 64;
  while ((size_t)R * cw * sizeof(float) > 64 * 1024 && cw > 8) cw >>= 1;
  *CW = cw;
  *slabs = (int)gnx_cdiv(H, cw);
  // ~768 blocks (3 per CU) in total; at least 256 rows each so the per-block table traffic stays a small fraction
  int64_t rpb = gnx_cdiv(N, gnx_cdiv(768, *slabs));
  if (rpb < 256) rpb = 256;
  *rows_per_block = rpb;
  *chunks = gnx_cdiv(N, rpb);
}

size_t gnx_table_scatter_ws_bytes(int64_t N, int R, int H) {
  if (N <= 0) return 0;
  int CW, slabs;
  int64_t rpb, chunks;
  table_scatter_geometry(N, R, H, &CW, &slabs, &rpb, &chunks);
  return sizeof(float) * (size_t)chunks * slabs * R * CW;
}

template <typename IdxT>
static int32_t launch_table_scatter_add(gnx_handle* h, const IdxT* idx, int64_t N, int K, const int32_t* offsets, int R,
                                        const float* dout, int H, float* dtable, void* ws, size_t ws_bytes) {
  offs_t o;
  for (int k = 0; k <= 17; ++k) o.o[k] = offsets[k <= K ? k : K];
  int CW, slabs;
  int64_t rows_per_block, chunks;
  table_scatter_geometry(N, R, H, &CW, &slabs, &rows_per_block, &chunks);
  GNX_CHECK_ARG((size_t)R * CW * sizeof(float) <= 64 * 1024, "table scatter-add: %d rows do not fit the LDS tile", R);
  float* part = nullptr;
  if (ws != nullptr && chunks > 8) {
    if (ws_bytes < gnx_table_scatter_ws_bytes(N, R, H)) {
      gnx_set_error("table scatter-add: workspace %zu < %zu", ws_bytes, gnx_table_scatter_ws_bytes(N, R, H));
      return GNX_E_WORKSPACE;
    }
    part = reinterpret_cast<float*>(ws);
  }
  // One column per lane: a wave's ds_add_f32 then touches 64 consecutive floats (2 lanes per bank, the minimum).  The
  // float4-per-thread layout put every lane of an instruction on 8 banks (8-way conflict on the LDS atomic unit:
  // 106 us for a 163840 x 128 gradient instead of ~25 us), so it is not used even when H % 4 == 0.
  if (false)
    hipLaunchKernelGGL((k_table_scatter_add<IdxT, 4>), dim3((unsigned)chunks, (unsigned)slabs), dim3(256),
                       (size_t)R * CW * sizeof(float), h->stream, idx, N, K, o, R, dout, H, CW, rows_per_block, dtable,
                       part);
  else
    hipLaunchKernelGGL((k_table_scatter_add<IdxT, 1>), dim3((unsigned)chunks, (unsigned)slabs), dim3(256),
                       (size_t)R * CW * sizeof(float), h->stream, idx, N, K, o, R, dout, H, CW, rows_per_block, dtable,
                       part);
  GNX_LAUNCH_CHECK();
  if (part != nullptr) {
    const int groups = (int)(chunks < 16 ? chunks : 16);
    const int cpg = (int)gnx_cdiv(chunks, groups);
    hipLaunchKernelGGL(k_table_reduce, dim3((unsigned)gnx_cdiv((int64_t)R * slabs * CW, 256), (unsigned)groups),
                       dim3(256), 0, h->stream, part, (int)chunks, slabs, R, CW, H, cpg, dtable);
    GNX_LAUNCH_CHECK();
  }
  return GNX_OK;
}

int32_t gnx_embed_bwd_mfma(gnx_handle* h, const int64_t* idx, int64_t N, int K, const int32_t* offsets, int R,
                           const float* dout, int H, float* dtable);  // gnx_gemm.hip

extern "C" size_t gnx_table_scatter_workspace_bytes(int64_t rows, int32_t R, int32_t H) {
  return gnx_table_scatter_ws_bytes(rows, R, H);
}

extern "C" int32_t gnx_embed_sum_bwd(gnx_handle* h, const int64_t* idx, int64_t N, int32_t K, const int32_t* offsets,
                                     int32_t R, const float* dout, int32_t H, float* dtable, void* ws, size_t ws_bytes) {
  GNX_CHECK_ARG(h && offsets && dtable && K > 0 && K <= 16 && H > 0 && R > 0 && N >= 0, "gnx_embed_sum_bwd: bad argument");
  GNX_CHECK_ARG(N == 0 || (idx && dout), "gnx_embed_sum_bwd: NULL array with N>0");
  GNX_CHECK_ARG(offsets[K] == R, "gnx_embed_sum_bwd: offsets[K]=%d != R=%d", offsets[K], R);
  if (N == 0) return GNX_OK;
  gnx_prof_scope prof(h, GNX_K_EMBED, 8.0 * N * K + 4.0 * N * H);
  {
    // large batches: the one-hot x gradient product on the MFMA (exact); small ones: LDS-privatised table adds
    const int32_t st = gnx_embed_bwd_mfma(h, idx, N, K, offsets, R, dout, H, dtable);
    if (st <= 0) return st;
  }
  return launch_table_scatter_add<int64_t>(h, idx, N, K, offsets, R, dout, H, dtable, ws, ws_bytes);
}

// used by gnx_edge_combine_bwd / gnx_gine_aggregate_bwd (int32 codes, one table)
int32_t gnx_code_scatter_add(gnx_handle* h, const int32_t* code, int64_t E, int R, const float* g, int H,
                             float* dtable, void* ws, size_t ws_bytes) {
  if (E == 0) return GNX_OK;
  int32_t offs[2] = {0, R};
  return launch_table_scatter_add<int32_t>(h, code, E, 1, offs, R, g, H, dtable, ws, ws_bytes);
}

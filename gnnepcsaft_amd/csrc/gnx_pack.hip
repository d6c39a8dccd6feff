// Input packer: PyG-style COO batch (int64) -> dst-sorted CSR + by-source index (int32), bond codes, graph_ptr,
// PNA degree scalers.  Integer work, bit-exact, deterministic (stable inside every row).
#include "gnx_common.hpp"

// ---------------------------------------------------------------------------------------------------------------
// exclusive scan of int32 counts (n elements) -> out[0..n] (out[n] = total). 1024 elements per block.
// ---------------------------------------------------------------------------------------------------------------
#define SCAN_ITEMS 4
#define SCAN_BLOCK 256
#define SCAN_TILE (SCAN_ITEMS * SCAN_BLOCK)

__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_tile(const int* __restrict__ in, int* __restrict__ out,
                                                          int* __restrict__ sums, int64_t n) {
  __shared__ int s[SCAN_BLOCK];
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int t = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = (base + i < n) ? in[base + i] : 0;
    t += v[i];
  }
  s[threadIdx.x] = t;
  __syncthreads();
  // Hillis-Steele inclusive scan over the 256 thread totals
  for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
    int add = (threadIdx.x >= off) ? s[threadIdx.x - off] : 0;
    __syncthreads();
    s[threadIdx.x] += add;
    __syncthreads();
  }
  int excl = s[threadIdx.x] - t;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) out[base + i] = excl;
    excl += v[i];
  }
  if (threadIdx.x == SCAN_BLOCK - 1) sums[blockIdx.x] = s[SCAN_BLOCK - 1];
}

__global__ void k_scan_add(int* __restrict__ out, const int* __restrict__ offs, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x;
  int o = offs[blockIdx.x];
  for (int k = 0; k < SCAN_ITEMS; ++k, i += SCAN_BLOCK)
    if (i < n) out[i] += o;
}

__global__ void k_set_total(int* __restrict__ out, const int* __restrict__ in, int64_t n) {
  // out[n] = out[n-1] + in[n-1]
  if (threadIdx.x == 0 && blockIdx.x == 0) out[n] = (n > 0) ? out[n - 1] + in[n - 1] : 0;
}

// out must hold n+1 ints when write_total, n otherwise.  ws: scan_ws_ints(n) ints.
static int32_t exclusive_scan(gnx_handle* h, const int* in, int* out, int64_t n, int* ws, bool write_total) {
  if (n > 0) {
    int64_t blocks = gnx_cdiv(n, SCAN_TILE);
    hipLaunchKernelGGL(k_scan_tile, dim3((unsigned)blocks), dim3(SCAN_BLOCK), 0, h->stream, in, out, ws, n);
    GNX_LAUNCH_CHECK();
    if (blocks > 1) {
      int* sums_scanned = ws + blocks + 1;
      // scan the block sums in place into a second array (recursive), then add
      int32_t st = exclusive_scan(h, ws, sums_scanned, blocks, sums_scanned + blocks + 1, false);
      if (st != GNX_OK) return st;
      hipLaunchKernelGGL(k_scan_add, dim3((unsigned)blocks), dim3(SCAN_BLOCK), 0, h->stream, out, sums_scanned, n);
      GNX_LAUNCH_CHECK();
    }
  }
  if (write_total) {
    hipLaunchKernelGGL(k_set_total, dim3(1), dim3(64), 0, h->stream, out, in, n);
    GNX_LAUNCH_CHECK();
  }
  return GNX_OK;
}

static size_t scan_ws_ints_total(int64_t n) {
  // recursion above uses, per level, blocks+1 (sums) followed by the next level's arrays
  size_t tot = 0;
  int64_t m = n;
  while (m > 1) {
    int64_t b = gnx_cdiv(m, SCAN_TILE);
    tot += 2 * ((size_t)b + 1);
    m = b;
  }
  return tot + 16;
}

// ---------------------------------------------------------------------------------------------------------------
__global__ void k_convert_edges(const int64_t* __restrict__ ei, int64_t E, int64_t N, int* __restrict__ src0,
                                int* __restrict__ dst0, int* __restrict__ flag) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t s = ei[e], d = ei[E + e];
  bool bad = (s < 0) | (s >= N) | (d < 0) | (d >= N);
  if (bad) {
    atomicOr(flag, 1);
    s = 0;
    d = 0;
  }
  src0[e] = (int)s;
  dst0[e] = (int)d;
}

__global__ void k_count_keys(const int* __restrict__ key, int64_t E, int* __restrict__ counts) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < E) atomicAdd(&counts[key[e]], 1);
}

__global__ void k_fill_groups(const int* __restrict__ key, int64_t E, const int* __restrict__ ptr,
                              int* __restrict__ cursor, int* __restrict__ items) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int k = key[e];
  int pos = ptr[k] + atomicAdd(&cursor[k], 1);
  items[pos] = (int)e;
}

// one thread per group: ascending insertion sort of its items (degrees are tiny for molecules; O(d^2) worst case)
__global__ void k_sort_groups(const int* __restrict__ ptr, int64_t N, int* __restrict__ items) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int b = ptr[i], e = ptr[i + 1];
  for (int a = b + 1; a < e; ++a) {
    int v = items[a];
    int c = a - 1;
    while (c >= b && items[c] > v) {
      items[c + 1] = items[c];
      --c;
    }
    items[c + 1] = v;
  }
}

__global__ void k_gather_endpoints(const int* __restrict__ perm, const int* __restrict__ src0,
                                   const int* __restrict__ dst0, int64_t E, int* __restrict__ src,
                                   int* __restrict__ dst) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= E) return;
  int e = perm[p];
  src[p] = src0[e];
  dst[p] = dst0[e];
}

static int32_t group_by_key(gnx_handle* h, const int* key, int64_t E, int64_t N, int* ptr, int* items, int* cursor,
                            int* scan_ws) {
  gnx_zero_ints(h, cursor, N > 0 ? N : 1);
  if (E > 0) {
    hipLaunchKernelGGL(k_count_keys, dim3((unsigned)gnx_cdiv(E, 256)), dim3(256), 0, h->stream, key, E, cursor);
    GNX_LAUNCH_CHECK();
  }
  int32_t st = exclusive_scan(h, cursor, ptr, N, scan_ws, true);
  if (st != GNX_OK) return st;
  gnx_zero_ints(h, cursor, N > 0 ? N : 1);
  if (E > 0) {
    hipLaunchKernelGGL(k_fill_groups, dim3((unsigned)gnx_cdiv(E, 256)), dim3(256), 0, h->stream, key, E, ptr, cursor,
                       items);
    GNX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sort_groups, dim3((unsigned)gnx_cdiv(N, 256)), dim3(256), 0, h->stream, ptr, N, items);
    GNX_LAUNCH_CHECK();
  }
  return GNX_OK;
}

extern "C" size_t gnx_pack_csr_workspace_bytes(int64_t N, int64_t E) {
  if (N < 0) N = 0;
  if (E < 0) E = 0;
  size_t ints = 2 * (size_t)E + (size_t)N + 1 + scan_ws_ints_total(N) + 64;
  return ints * sizeof(int);
}

extern "C" int32_t gnx_pack_csr(gnx_handle* h, const int64_t* edge_index, int64_t E, int64_t N, int32_t* rowptr,
                                int32_t* perm, int32_t* src, int32_t* dst, int32_t* colptr, int32_t* cpos, void* ws,
                                size_t ws_bytes) {
  GNX_CHECK_ARG(h != nullptr, "gnx_pack_csr: handle is NULL");
  GNX_CHECK_ARG(N >= 0 && E >= 0 && N < (1ll << 31) - 1 && E < (1ll << 31) - 1, "gnx_pack_csr: N=%lld E=%lld out of int32 range",
                (long long)N, (long long)E);
  GNX_CHECK_ARG(rowptr && colptr, "gnx_pack_csr: rowptr/colptr NULL");
  GNX_CHECK_ARG(E == 0 || (edge_index && perm && src && dst && cpos), "gnx_pack_csr: NULL edge array with E>0");
  if (ws_bytes < gnx_pack_csr_workspace_bytes(N, E) || (!ws && ws_bytes)) {
    gnx_set_error("gnx_pack_csr: workspace %zu < %zu", ws_bytes, gnx_pack_csr_workspace_bytes(N, E));
    return GNX_E_WORKSPACE;
  }
  GNX_CHECK_ARG(ws != nullptr, "gnx_pack_csr: workspace is NULL");
  int* w = reinterpret_cast<int*>(ws);
  int* src0 = w;
  int* dst0 = src0 + E;
  int* cursor = dst0 + E;
  int* scan_ws = cursor + N + 1;
  if (E > 0) {
    hipLaunchKernelGGL(k_convert_edges, dim3((unsigned)gnx_cdiv(E, 256)), dim3(256), 0, h->stream, edge_index, E, N,
                       src0, dst0, h->d_flag);
    GNX_LAUNCH_CHECK();
  }
  int32_t st = group_by_key(h, dst0, E, N, rowptr, perm, cursor, scan_ws);
  if (st != GNX_OK) return st;
  if (E > 0) {
    hipLaunchKernelGGL(k_gather_endpoints, dim3((unsigned)gnx_cdiv(E, 256)), dim3(256), 0, h->stream, perm, src0,
                       dst0, E, src, dst);
    GNX_LAUNCH_CHECK();
  }
  return group_by_key(h, src, E, N, colptr, cpos, cursor, scan_ws);
}

// ---------------------------------------------------------------------------------------------------------------
struct dims_t {
  int d[16];
};

__global__ void k_feature_code(const int64_t* __restrict__ feat, int64_t rows, int K, dims_t dims,
                               const int* __restrict__ perm, int* __restrict__ code, int* __restrict__ flag) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= rows) return;
  int64_t r = perm ? perm[p] : p;
  int c = 0;
  bool bad = false;
  for (int k = 0; k < K; ++k) {
    int64_t f = feat[r * K + k];
    if (f < 0 || f >= dims.d[k]) {
      bad = true;
      f = 0;
    }
    c = c * dims.d[k] + (int)f;
  }
  if (bad) atomicOr(flag, 2);
  code[p] = c;
}

extern "C" int32_t gnx_feature_code(gnx_handle* h, const int64_t* feat, int64_t rows, int32_t K, const int32_t* dims,
                                    const int32_t* perm, int32_t* code, void* ws, size_t ws_bytes) {
  (void)ws;
  (void)ws_bytes;
  GNX_CHECK_ARG(h && dims && K > 0 && K <= 16 && rows >= 0, "gnx_feature_code: bad argument");
  GNX_CHECK_ARG(rows == 0 || (feat && code), "gnx_feature_code: NULL array with rows>0");
  dims_t d;
  int64_t prod = 1;
  for (int k = 0; k < 16; ++k) d.d[k] = (k < K) ? dims[k] : 1;
  for (int k = 0; k < K; ++k) {
    GNX_CHECK_ARG(dims[k] > 0, "gnx_feature_code: dims[%d] <= 0", k);
    prod *= dims[k];
  }
  GNX_CHECK_ARG(prod < (1ll << 31), "gnx_feature_code: code space overflows int32");
  if (rows > 0) {
    hipLaunchKernelGGL(k_feature_code, dim3((unsigned)gnx_cdiv(rows, 256)), dim3(256), 0, h->stream, feat, rows, (int)K,
                       d, perm, code, h->d_flag);
    GNX_LAUNCH_CHECK();
  }
  return GNX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// batch (non-decreasing) -> graph_ptr: ptr[g] = first n with batch[n] >= g ; ptr[B] = N
__global__ void k_graph_ptr(const int64_t* __restrict__ batch, int64_t N, int64_t B, int* __restrict__ ptr,
                            int* __restrict__ flag) {
  int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n > N) return;
  int64_t prev = (n == 0) ? -1 : batch[n - 1];
  int64_t cur = (n == N) ? B : batch[n];
  if (n < N && (cur < 0 || cur >= B)) {
    atomicOr(flag, 4);
    return;
  }
  if (cur < prev) {
    atomicOr(flag, 8);
    return;
  }
  for (int64_t g = prev + 1; g <= cur; ++g) ptr[g] = (int)n;
}

extern "C" int32_t gnx_graph_ptr(gnx_handle* h, const int64_t* batch, int64_t N, int64_t B, int32_t* graph_ptr,
                                 void* ws, size_t ws_bytes) {
  (void)ws;
  (void)ws_bytes;
  GNX_CHECK_ARG(h && graph_ptr && N >= 0 && B >= 0 && (batch || N == 0), "gnx_graph_ptr: bad argument");
  hipLaunchKernelGGL(k_graph_ptr, dim3((unsigned)gnx_cdiv(N + 1, 256)), dim3(256), 0, h->stream, batch, N, B,
                     graph_ptr, h->d_flag);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
__global__ void k_degree_scalers(const int* __restrict__ rowptr, int64_t N, float avg_log, float* __restrict__ amp,
                                 float* __restrict__ att) {
  int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float d = (float)(rowptr[n + 1] - rowptr[n]);
  amp[n] = logf(d + 1.0f) / avg_log;
  att[n] = avg_log / logf(fmaxf(d, 1.0f) + 1.0f);
}

extern "C" int32_t gnx_degree_scalers(gnx_handle* h, const int32_t* rowptr, int64_t N, float avg_deg_log, float* amp,
                                      float* att) {
  GNX_CHECK_ARG(h && N >= 0 && (N == 0 || (rowptr && amp && att)), "gnx_degree_scalers: bad argument");
  if (N == 0) return GNX_OK;
  hipLaunchKernelGGL(k_degree_scalers, dim3((unsigned)gnx_cdiv(N, 256)), dim3(256), 0, h->stream, rowptr, N,
                     avg_deg_log, amp, att);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// In-degree classes: nodes stably sorted by in-degree (a deterministic counting sort: per-block histograms -> scan ->
// stable ranks), and tile tables that never straddle two classes.  Lets PNA's post-layer 0 use one effective weight
// W_eff(d) = W1 + amp(d) W2 + att(d) W3 per class instead of the 12F-wide scaled operand (amp/att depend on d only).
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_degree_max(const int* __restrict__ rowptr, int64_t N, int* __restrict__ out) {
  int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int d = (n < N) ? rowptr[n + 1] - rowptr[n] : 0;
  // wave max, then one atomic per wave
  for (int off = 32; off > 0; off >>= 1) d = max(d, __shfl_xor(d, off));
  if ((threadIdx.x & 63) == 0) atomicMax(out, d);
}

extern "C" int32_t gnx_degree_max(gnx_handle* h, const int32_t* rowptr, int64_t N, int32_t* max_degree_host) {
  GNX_CHECK_ARG(h && max_degree_host && N >= 0 && (N == 0 || rowptr), "gnx_degree_max: bad argument");
  int* d_out = h->d_flag + 8;  // scratch word inside the handle's flag block
  gnx_zero_ints(h, d_out, 1);
  if (N > 0) {
    hipLaunchKernelGGL(k_degree_max, dim3((unsigned)gnx_cdiv(N, 256)), dim3(256), 0, h->stream, rowptr, N, d_out);
    GNX_LAUNCH_CHECK();
  }
  int v = 0;
  GNX_HIP(hipMemcpyAsync(&v, d_out, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  GNX_HIP(hipStreamSynchronize(h->stream));
  *max_degree_host = v;
  return GNX_OK;
}

#define DC_BLOCK 256
#define DC_MAX_CLASSES 64

// blockhist[d * nblocks + b] = #nodes of block b with in-degree d
// keys != NULL: key of item n = keys[n]; else key = in-degree rowptr[n+1]-rowptr[n]
__global__ void __launch_bounds__(DC_BLOCK) k_dc_hist(const int* __restrict__ rowptr, const int* __restrict__ keys,
                                                       int64_t N, int D, int nblocks, int* __restrict__ blockhist,
                                                       int* __restrict__ flag) {
  __shared__ int hist[DC_MAX_CLASSES];
  if (threadIdx.x < DC_MAX_CLASSES) hist[threadIdx.x] = 0;
  __syncthreads();
  int64_t n = (int64_t)blockIdx.x * DC_BLOCK + threadIdx.x;
  if (n < N) {
    int d = keys ? keys[n] : rowptr[n + 1] - rowptr[n];
    if (d < 0) d = 0;
    if (d >= D) {  // only possible when D came from a caller's hint: reported through the sticky flag (bit 5)
      atomicOr(flag, 32);
      d = D - 1;
    }
    atomicAdd(&hist[d], 1);
  }
  __syncthreads();
  if (threadIdx.x < D) blockhist[threadIdx.x * nblocks + blockIdx.x] = hist[threadIdx.x];
}

// stable position of every node: scanned block offset of its (class, block) + rank among earlier same-class threads
__global__ void __launch_bounds__(DC_BLOCK) k_dc_fill(const int* __restrict__ rowptr, const int* __restrict__ keys,
                                                       int64_t N, int D, int nblocks,
                                                       const int* __restrict__ blockoff, int* __restrict__ dperm,
                                                       int* __restrict__ cls_ptr) {
  __shared__ int degs[DC_BLOCK];
  int64_t n = (int64_t)blockIdx.x * DC_BLOCK + threadIdx.x;
  int d = -1;
  if (n < N) {
    d = keys ? keys[n] : rowptr[n + 1] - rowptr[n];
    if (d < 0) d = 0;
    if (d >= D) d = D - 1;
  }
  degs[threadIdx.x] = d;
  __syncthreads();
  if (n < N) {
    int rank = 0;
    for (int t = 0; t < (int)threadIdx.x; ++t) rank += (degs[t] == d) ? 1 : 0;
    dperm[blockoff[d * nblocks + blockIdx.x] + rank] = (int)n;
  }
  if (blockIdx.x == 0 && (int)threadIdx.x <= D)
    cls_ptr[threadIdx.x] = ((int)threadIdx.x == D) ? (int)N : blockoff[threadIdx.x * nblocks];
}

extern "C" size_t gnx_degree_classes_workspace_bytes(int64_t N, int32_t D) {
  if (N < 0) N = 0;
  int64_t nblocks = gnx_cdiv(N > 0 ? N : 1, DC_BLOCK);
  size_t ints = 2 * (size_t)(D * nblocks + 1) + scan_ws_ints_total(D * nblocks) + 64;
  return ints * sizeof(int);
}

extern "C" int32_t gnx_degree_classes(gnx_handle* h, const int32_t* rowptr, int64_t N, int32_t D, int32_t* dperm,
                                      int32_t* cls_ptr, void* ws, size_t ws_bytes) {
  GNX_CHECK_ARG(h && rowptr && cls_ptr && N >= 0 && D >= 1 && D <= DC_MAX_CLASSES && (N == 0 || dperm),
                "gnx_degree_classes: bad argument (D must be in [1,%d])", DC_MAX_CLASSES);
  if (ws_bytes < gnx_degree_classes_workspace_bytes(N, D) || !ws) {
    gnx_set_error("gnx_degree_classes: workspace %zu < %zu", ws_bytes, gnx_degree_classes_workspace_bytes(N, D));
    return GNX_E_WORKSPACE;
  }
  int nblocks = (int)gnx_cdiv(N > 0 ? N : 1, DC_BLOCK);
  int* blockhist = reinterpret_cast<int*>(ws);
  int* blockoff = blockhist + (size_t)D * nblocks + 1;
  int* scan_ws = blockoff + (size_t)D * nblocks + 1;
  hipLaunchKernelGGL(k_dc_hist, dim3(nblocks), dim3(DC_BLOCK), 0, h->stream, rowptr, (const int*)nullptr, N, (int)D,
                     nblocks, blockhist, h->d_flag);
  GNX_LAUNCH_CHECK();
  int32_t st = exclusive_scan(h, blockhist, blockoff, (int64_t)D * nblocks, scan_ws, false);
  if (st != GNX_OK) return st;
  hipLaunchKernelGGL(k_dc_fill, dim3(nblocks), dim3(DC_BLOCK), 0, h->stream, rowptr, (const int*)nullptr, N, (int)D,
                     nblocks, blockoff, dperm, cls_ptr);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

// Stable grouping of E items by a small integer key (0 <= key < R <= 64): pos int32[E] = item ids sorted by key
// (ascending id inside a key), ptr int32[R+1].  Used for the per-edge bond code: the inverted index lets the bond-table
// gradient be a gather-sum over contiguous index runs instead of 10^7 atomics.  ws: gnx_degree_classes_workspace_bytes(E, R).
extern "C" int32_t gnx_group_by_small_key(gnx_handle* h, const int32_t* keys, int64_t E, int32_t R, int32_t* pos,
                                          int32_t* ptr, void* ws, size_t ws_bytes) {
  GNX_CHECK_ARG(h && ptr && E >= 0 && R >= 1 && R <= DC_MAX_CLASSES && (E == 0 || (keys && pos)),
                "gnx_group_by_small_key: bad argument (R must be in [1,%d])", DC_MAX_CLASSES);
  if (ws_bytes < gnx_degree_classes_workspace_bytes(E, R) || !ws) {
    gnx_set_error("gnx_group_by_small_key: workspace %zu < %zu", ws_bytes, gnx_degree_classes_workspace_bytes(E, R));
    return GNX_E_WORKSPACE;
  }
  int nblocks = (int)gnx_cdiv(E > 0 ? E : 1, DC_BLOCK);
  int* blockhist = reinterpret_cast<int*>(ws);
  int* blockoff = blockhist + (size_t)R * nblocks + 1;
  int* scan_ws = blockoff + (size_t)R * nblocks + 1;
  hipLaunchKernelGGL(k_dc_hist, dim3(nblocks), dim3(DC_BLOCK), 0, h->stream, (const int*)nullptr, keys, E, (int)R,
                     nblocks, blockhist, h->d_flag);
  GNX_LAUNCH_CHECK();
  int32_t st = exclusive_scan(h, blockhist, blockoff, (int64_t)R * nblocks, scan_ws, false);
  if (st != GNX_OK) return st;
  hipLaunchKernelGGL(k_dc_fill, dim3(nblocks), dim3(DC_BLOCK), 0, h->stream, (const int*)nullptr, keys, E, (int)R,
                     nblocks, blockoff, pos, ptr);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

// tile table: for class c, ceil(n_c / tile_rows) tiles of (first position in dperm, #rows, class); ntiles[0] = count.
// One thread per tile (grid-stride): the class of tile t is found by walking the D <= 4096 class sizes (a serial loop of one
// thread over ~1000 tiles took 15-28 us inside every step's packing phase).
__global__ void __launch_bounds__(256) k_class_tiles(const int* __restrict__ cls_ptr, int D, int tile_rows,
                                                     int* __restrict__ tile_info, int* __restrict__ ntiles) {
  int total = 0;
  for (int c = 0; c < D; ++c) total += (cls_ptr[c + 1] - cls_ptr[c] + tile_rows - 1) / tile_rows;
  if (blockIdx.x == 0 && threadIdx.x == 0) ntiles[0] = total;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    int first = 0, c = 0;
    for (; c < D; ++c) {
      const int nt = (cls_ptr[c + 1] - cls_ptr[c] + tile_rows - 1) / tile_rows;
      if (t < first + nt) break;
      first += nt;
    }
    const int r = cls_ptr[c] + (t - first) * tile_rows;
    const int rows = cls_ptr[c + 1] - r;
    tile_info[3 * t + 0] = r;
    tile_info[3 * t + 1] = rows < tile_rows ? rows : tile_rows;
    tile_info[3 * t + 2] = c;
  }
}

extern "C" int32_t gnx_class_tiles(gnx_handle* h, const int32_t* cls_ptr, int32_t D, int32_t tile_rows,
                                   int32_t* tile_info, int32_t* ntiles) {
  GNX_CHECK_ARG(h && cls_ptr && tile_info && ntiles && D >= 1 && tile_rows > 0, "gnx_class_tiles: bad argument");
  hipLaunchKernelGGL(k_class_tiles, dim3(16), dim3(256), 0, h->stream, cls_ptr, (int)D, (int)tile_rows, tile_info, ntiles);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

/*
 * gnx.h — C ABI of libgnnepcsaft_hip.so: the MI355X (gfx950) kernels behind gnnepcsaft's GNN forward/backward.
 *
 * The reference has no FFI for this path: the arithmetic sits in torch_geometric / ogb Python modules that
 * /root/reference/gnnepcsaft/train/models.py only wires together.  Each entry point below therefore cites the
 * reference call site (ref: = /root/reference/gnnepcsaft/...) and the third-party op [3P] it stands in for.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no torch types.
 *  - every function returns int32 status: 0 = ok, <0 = error (GNX_E_*); text via gnx_last_error() (thread-local).
 *  - the library never allocates or frees tensor memory: all buffers (inputs, outputs, workspaces) are caller-owned
 *    DEVICE pointers (hipMalloc / torch tensors' data_ptr()); workspace sizes are queried with *_workspace_bytes.
 *    (The handle itself owns 4.25 KiB of device memory for the sticky range flag, allocated in gnx_create.)
 *  - all launches go on the handle's stream (gnx_set_stream; default = the NULL stream) and are asynchronous.
 *    Only gnx_check_range (integer input validation read-back) and gnx_prof_end synchronise that stream.
 *  - all float tensors are fp32 row-major; "ld" = leading dimension in elements.  Index tensors handed over by the
 *    caller in the reference's layout are int64; everything the library produces is int32.
 *  - NO gnx_comm_* entry points (SURVEY.md §8b lists gnx_comm_init / gnx_allreduce_f32 as the stand-in for Lightning
 *    DDP's gradient exchange, ref: train/train.py:85-88): the exchange is torch.distributed's "nccl" backend = the RCCL
 *    that PyTorch-ROCm bundles, driven by gnnepcsaft_amd/dp.py on this library's side stream (gnx_side_stream).  A second
 *    RCCL linked into this library would live beside torch's copy in the same process; gnx_scale (below) is the only
 *    arithmetic of the exchange (sum -> average).
 *  - no hidden global state except the opaque gnx_handle (device id, stream, options, profiling events).  The GNX_*
 *    environment variables are read ONCE, in gnx_create, as initial option values; no entry point reads the
 *    environment afterwards (gnx_set_option changes an option at run time).
 */
#ifndef GNX_H_
#define GNX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNX_ABI_VERSION 3

enum {
  GNX_OK = 0,
  GNX_E_INVALID = -1,   /* bad argument (shape, alignment, null pointer) */
  GNX_E_HIP = -2,       /* HIP runtime error; see gnx_last_error() */
  GNX_E_RANGE = -3,     /* integer input out of range (node id >= N, feature >= vocab, unsorted batch) */
  GNX_E_WORKSPACE = -4  /* workspace too small */
};

typedef struct gnx_handle gnx_handle;

/* ---- lifecycle --------------------------------------------------------------------------------------------- */
int32_t gnx_create(gnx_handle** out, int32_t device);
int32_t gnx_destroy(gnx_handle* h);
int32_t gnx_set_stream(gnx_handle* h, void* hip_stream);
const char* gnx_last_error(void);
int32_t gnx_abi_version(void);

/* A/B switches of the handle (kernel selection only; every setting computes the same function).  Initial value = the
 * environment variable of the same name with the GNX_ prefix (e.g. GNX_GEMM_SPLIT=0), read once in gnx_create. */
enum {
  GNX_OPT_GEMM_SPLIT = 0,        /* 1: products with >= 4096 rows as three-bf16-piece split products; 0: fp32 MFMA */
  GNX_OPT_GEMM_WS = 1,           /* 1: weights-stationary kernel for one segment with K, N <= 128, M >= 8192 */
  GNX_OPT_GEMM_VEC = 2,          /* 0: element-wise loaders everywhere (debug) */
  GNX_OPT_WGRAD_VEC = 3,         /* 0: element-wise weight-gradient loaders (debug) */
  GNX_OPT_WGRAD_WGS = 4,         /* > 0: target workgroup count of the weight-gradient kernels (default: #CUs) */
  GNX_OPT_AGG_BWD_RECOMPUTE = 5, /* 1: PNA aggregate backward recomputes mean/min/max/std from the messages */
  GNX_OPT_EMBED_BWD_MFMA = 6,    /* atom-embedding gradient as a one-hot MFMA product for N >= 256: 1 = scattered bf16 one-hot x three-piece gradient (tables of <= 192 rows), 2 = computed fp32 one-hot on the fp32 matrix pipe, 0 = LDS atomics */
  GNX_OPT_STD_BWD_CENTERED = 7,  /* 1: std gradient divides by the centred two-pass std (see gnx_pna_aggregate_bwd) */
  GNX_OPT_GEMM_PIPE = 8,         /* 1: tiled split products with >= 8 K-tiles per tile take the software-pipelined kernel (bit-identical results) */
  GNX_OPT_WGRAD_PIPE = 9,        /* 1: split weight gradients of 16-byte aligned operands through the software-pipelined kernel */
  GNX_OPT_EDGE_FUSED = 10,       /* 1: gnx_pna_conv_fwd / _bwd take the fused edge kernels (gnx_pna_edge_fwd / gnx_pna_edge_bwd) when eligible (bit-identical messages, h1, aggregate, gh1, dP); 2: forward only; 0: three launches each */
  GNX_OPT_SIDE_CUS = 11,         /* > 0: side stream 0 (weight gradients) is created with a CU mask of that many CUs (read when the stream is first used) */
  GNX_OPT_GEMM_AS = 12,          /* 1: split products with one segment, 96 < K <= 128 and N >= 256 take the activation-stationary kernel (the row tile is split once for all column tiles; bit-identical results) */
  GNX_OPT_GEMM_WS_FAST = 13,     /* the weights-stationary split kernel's predicate-free form (quad-transposed 16-byte stores, exact waits) for N = 128, aligned operands, no accumulate: 2 = as two 4-wave workgroups per CU on 32-row tiles (default), 1 = one 8-wave workgroup per CU on 64-row tiles, 0 = the predicated kernel; bit-identical results */
  GNX_OPT_GEMM_TILE_ROWS = 14,   /* 96 / 128: row-tile height of the pipelined tiled product (0: chosen per launch; bit-identical results) */
  GNX_OPT_GEMM_MID = 15,         /* 1: products with M < 4096 rows and <= 48 tiles of 128 x 128 run on 16 x 16 patches (k_gemm_mid) instead of the latency-bound tiled kernel */
  GNX_OPT_SPLIT_AHEAD = 16,      /* gnx_pna_conv_fwd / _bwd split the weight images of their tiled products on side stream 2 at entry (GNX_GEMM_SPLIT_ONLY) instead of in front of each product: 1 = for layers with two or more towers (cfg-5: -0.44 ms per step; one tower: the fork / join costs what it saves), 2 = always, 0 = never */
  GNX_OPT_COUNT = 17
};
int32_t gnx_set_option(gnx_handle* h, int32_t opt, int32_t value);
int32_t gnx_get_option(gnx_handle* h, int32_t opt, int32_t* value);

/* ---- profiling hook (bench.py's live per-kernel timing; HIP events on the handle's stream) ------------------- */
/* kernel ids (groups of kernels with one roofline) */
enum {
  GNX_K_NONE = 0,
  GNX_K_PNA_AGG_FWD = 1,        /* scatter-aggregate forward (the metric's roofline kernel) */
  GNX_K_PNA_AGG_BWD = 2,
  GNX_K_GEMM_WS = 3,            /* weights-stationary products (K, N <= 128) */
  GNX_K_GEMM_WGRAD = 4,         /* single / per-degree-class weight gradients */
  GNX_K_GINE_AGG_FWD = 5,
  GNX_K_GINE_AGG_BWD = 6,
  GNX_K_EDGE_COMBINE_FWD = 7,
  GNX_K_EDGE_COMBINE_BWD = 8,
  GNX_K_BN_FWD = 9,
  GNX_K_BN_BWD = 10,
  GNX_K_GEMM_TILED = 11,        /* tiled multi-segment / degree-class-grouped products (incl. their weight split) */
  GNX_K_GEMM_SMALL = 12,        /* M <= 256 products (60-row bond-table chain, merged H x H weights) */
  GNX_K_GEMM_WGRAD_BATCHED = 13,/* a layer's weight gradients in one launch */
  GNX_K_KEY_SEGMENT_SUM = 14,   /* bond-table gradient through the inverted index */
  GNX_K_EMBED = 15,             /* embedding sums, forward and backward */
  GNX_K_PNA_EDGE_FWD = 16,      /* fused message assembly -> pre-layer 1 -> scatter-aggregate (gnx_pna_edge_fwd) */
  GNX_K_PNA_EDGE_BWD = 17,      /* fused masked input gradient of pre-layer 1 + destination sums + bond-table sums (gnx_pna_edge_bwd) */
  GNX_K_COUNT = 18
};
/* start recording a HIP event pair around every launch of the kernels whose id bit is set in kernel_mask
 * (bit k = GNX_K_* id k).  Events go on the handle's stream, i.e. the stream the kernels run on. */
int32_t gnx_prof_begin(gnx_handle* h, uint32_t kernel_mask);
/* synchronise the stream; #launches of kernel group `kid` seen since gnx_prof_begin, their summed duration (ms), and
 * the sums of what the launches were given to do: ALGORITHMIC bytes (operands read + results written once, fp32 /
 * int32), algorithmic FLOPs (2 M N K of the products) and the bf16-MFMA FLOPs actually executed (6 x the
 * algorithmic ones on the split-operand kernels, 0 on fp32-MFMA kernels).  Any of the three may be NULL. */
int32_t gnx_prof_read(gnx_handle* h, int32_t kid, int64_t* launches, double* total_ms, double* alg_bytes,
                      double* alg_flops, double* mfma_bf16_flops);
/* stop recording and drop the recorded events. */
int32_t gnx_prof_end(gnx_handle* h);

/* ---- input packer: PyG-style COO batch -> dst-sorted CSR (+ by-source index) ------------------------------- */
/* Replaces what PNAConv/GINEConv's MessagePassing.propagate [3P] does implicitly with edge_index on every call
 * (ref: train/models.py:212-214), and utils.degree [3P] (ref: train/models.py:445-457 via DegreeScalerAggregation).
 *   edge_index int64[2,E] (row 0 = source j, row 1 = target i), N nodes.
 *   rowptr  int32[N+1]  CSR by target;  perm int32[E]: CSR position p -> original edge id, STABLE (ascending inside a
 *   row, i.e. the order scatter_add_ visits them);  src int32[E], dst int32[E]: endpoints in CSR order;
 *   colptr int32[N+1], cpos int32[E]: for every source node the ascending list of CSR positions leaving it.
 * Bit-exact integer work.  Out-of-range node ids are clamped and set the handle's sticky range flag (bit 0), which
 * gnx_check_range reports (the only synchronising call), so a training loop can validate once per step. */
size_t gnx_pack_csr_workspace_bytes(int64_t N, int64_t E);
int32_t gnx_pack_csr(gnx_handle* h, const int64_t* edge_index, int64_t E, int64_t N, int32_t* rowptr, int32_t* perm,
                     int32_t* src, int32_t* dst, int32_t* colptr, int32_t* cpos, void* ws, size_t ws_bytes);
/* mixed-radix code of the K integer features of every row, gathered through perm (perm may be NULL = identity):
 *   code[p] = ((f[perm[p],0]*dims[1] + f[perm[p],1])*dims[2] + ...) ; dims is a HOST array of K vocab sizes.
 * Used for edge_attr int64[E,3] -> bond code in [0,60) (vocab ref: data/ogb_utils.py:24-33). Range-checked lazily (sticky flag bit 1).
 * ws: unused (may be NULL). */
int32_t gnx_feature_code(gnx_handle* h, const int64_t* feat, int64_t rows, int32_t K, const int32_t* dims,
                         const int32_t* perm, int32_t* code, void* ws, size_t ws_bytes);
/* batch int64[N] (non-decreasing graph id per node, PyG Batch.batch) -> graph_ptr int32[B+1]. Range/sortedness
 * checked lazily (sticky flag bits 2/3). ws: unused (may be NULL). */
int32_t gnx_graph_ptr(gnx_handle* h, const int64_t* batch, int64_t N, int64_t B, int32_t* graph_ptr, void* ws,
                      size_t ws_bytes);
/* PNA degree scalers [3P DegreeScalerAggregation]: amp[n] = log(d+1)/avg_deg_log, att[n] = avg_deg_log/log(max(d,1)+1),
 * d = rowptr[n+1]-rowptr[n]. */
int32_t gnx_degree_scalers(gnx_handle* h, const int32_t* rowptr, int64_t N, float avg_deg_log, float* amp, float* att);

/* ---- embeddings: AtomEncoder / BondEncoder [3P ogb] (ref: train/models.py:175-176, 205-206) ----------------- */
/* out[n,:] = sum_{k<K} table[offsets[k] + idx[n,k], :], summed left to right in fp32 (bit-exact with the CPU path).
 * idx int64[N,K]; table fp32[R,H] = the K embedding tables stacked; offsets: HOST int32[K+1] (row offsets, last = R).
 * Range-checked lazily: an out-of-range index poisons nothing (it is clamped) and sets the handle's sticky range
 * flag, reported by the next gnx_pack_* / gnx_check_range call. */
int32_t gnx_embed_sum_fwd(gnx_handle* h, const int64_t* idx, int64_t N, int32_t K, const int32_t* offsets,
                          const float* table, int32_t H, float* out);
/* dtable[R,H] += scatter-add of dout[N,H] (embedding_dense_backward): per-workgroup LDS tables (<= 256 rows), then a
 * two-stage fold through the caller's workspace (gnx_table_scatter_workspace_bytes(N, R, H); ws may be NULL = direct
 * atomics, slower: hundreds of workgroups adding into the same few KB serialise at the memory-side atomic units). */
size_t gnx_table_scatter_workspace_bytes(int64_t rows, int32_t R, int32_t H);
int32_t gnx_embed_sum_bwd(gnx_handle* h, const int64_t* idx, int64_t N, int32_t K, const int32_t* offsets,
                          int32_t R, const float* dout, int32_t H, float* dtable, void* ws, size_t ws_bytes);
int32_t gnx_check_range(gnx_handle* h); /* sync; GNX_E_RANGE if any lazy check tripped since the last call */

/* ---- dense contractions on the matrix cores ------------------------------------------------------------------
 * Arithmetic: products with >= 4096 rows evaluate every fp32 operand as an exact sum of three bf16 pieces and the
 * product as six v_mfma_f32_32x32x16_bf16 with fp32 accumulation (fp32-faithful: dropped terms <= 2^-25 |a||b|;
 * DESIGN.md §2); smaller products, row-scaled segments and GNX_OPT_GEMM_SPLIT = 0 use v_mfma_f32_32x32x2_f32. */
/* One A-operand segment of a K-concatenated product.  The product is
 *     C[m,n] (+)= act( sum_s  sum_{k<K_s} (rs_s[m] * A_s[m,k]) * B_s(k,n)  + bias[n] )        m<M, n<N
 *   b_trans = 1 ("NT", forward Linear):   B_s(k,n) = B_s[n*ldb + k]   (weight [out,in] row-major, PyG/torch layout)
 *   b_trans = 0 ("NN", input gradient):   B_s(k,n) = B_s[k*ldb + n]
 * rowscale may be NULL (=1).  Segments let the caller feed torch.cat([...], dim=-1) operands without materialising
 * them: PNAConv's [x_i || x_j || e] pre-layer and the 13F-wide [x || A || amp*A || att*A] post-layer
 * ([3P] PNAConv.message / DegreeScalerAggregation; ref: train/models.py:445-457). */
typedef struct {
  const float* a;        /* [M, k] with leading dimension lda */
  int64_t lda;
  const float* rowscale; /* [M] or NULL */
  const float* b;        /* segment's slice of the weight */
  int64_t ldb;
  int32_t k;
  int32_t _pad;
} gnx_gemm_seg;

enum {
  GNX_GEMM_RELU = 1,       /* C = max(.,0) */
  GNX_GEMM_ACCUMULATE = 2, /* C += (applied before relu; relu+accumulate is rejected) */
  GNX_GEMM_B_TRANS = 4,    /* NT */
  GNX_GEMM_SPLIT_ONLY = 8, /* only write the split weight images this call would use into ws (nothing else is launched; a call that
                              needs no images does nothing): lets the caller run the split ahead of the product, on another stream */
  GNX_GEMM_PRESPLIT = 16   /* ws already holds those images (a GNX_GEMM_SPLIT_ONLY call with the same arguments) */
};
/* mask (optional, may be NULL): multiply the result by (mask[m*ldmask+n] > 0) — ReLU backward fused in dgrad.
 * ws / ws_bytes: caller-owned device scratch for the split weight images of the tiled split-operand kernel, size from
 * gnx_gemm_workspace_bytes with the same arguments (0 = this call needs none).  ws == NULL is allowed: the product
 * then runs on the fp32-MFMA kernel.  A non-NULL workspace that is too small returns GNX_E_WORKSPACE.  The scratch is
 * free for reuse as soon as the launches of this call have run (stream order). */
size_t gnx_gemm_workspace_bytes(gnx_handle* h, int32_t nseg, const gnx_gemm_seg* segs, const int64_t* cls_strides,
                                int32_t num_classes, int64_t M, int32_t N, const float* mask, int32_t flags,
                                int32_t grouped);
int32_t gnx_gemm(gnx_handle* h, int32_t nseg, const gnx_gemm_seg* segs, int64_t M, int32_t N, const float* bias,
                 const float* mask, int64_t ldmask, float* C, int64_t ldc, int32_t flags, void* ws, size_t ws_bytes);
/* weight gradient ("TN"): dW[n, k] += sum_m dC[m,n] * (rs[m] * A[m,k]),  n<N, k<K.  Split over M across workgroups,
 * fp32 atomics into dW (caller zeroes or accumulates).  dbias (optional) += column sums of dC. */
int32_t gnx_gemm_wgrad(gnx_handle* h, const float* dC, int64_t lddc, const float* A, int64_t lda,
                       const float* rowscale, int64_t M, int32_t N, int32_t K, float* dW, int64_t lddw, float* dbias);

/* several independent weight gradients in ONE launch (a layer's same-shaped dW += dC^T (rs*A) products share the grid:
 * every workgroup gets an nprob x longer row range, so the atomic flush per problem shrinks nprob x).  nprob <= 8;
 * every operand 16-byte aligned with leading dimensions, N and K multiples of 4. */
typedef struct {
  const float* dC;       /* [M, N], ld lddc */
  int64_t lddc;
  const float* A;        /* [M, K], ld lda */
  int64_t lda;
  const float* rowscale; /* [M] or NULL */
  float* dW;             /* [N, K], ld lddw (accumulated) */
  int64_t lddw;
  float* dbias;          /* [N] or NULL (accumulated) */
  int64_t M;
  int32_t N, K;
} gnx_wgrad_prob;
int32_t gnx_gemm_wgrad_batched(gnx_handle* h, int32_t nprob, const gnx_wgrad_prob* probs);

/* several independent SMALL products (M, N <= a few hundred rows: the 60-row bond-table chain, the merged H x H weights
 * and their gradients) in one launch -- for all layers of a model at once instead of ~9 launches of ~8 us per layer:
 *     C (+)= act( op(A) op(B) + bias )     op(A) = A [M,K] or, with GNX_SB_A_TRANS, A stored [K,M] (a TN weight gradient);
 *                                          op(B): GNX_SB_B_TRANS = B stored [N,K] (torch weight layout), else [K,N].
 * GNX_SB_ACCUMULATE: plain read-modify-write (outputs of the launch's problems must then be disjoint);
 * GNX_SB_ATOMIC: fp32 atomic adds (several problems may add into one output; the output must be initialised). */
enum { GNX_SB_A_TRANS = 1, GNX_SB_B_TRANS = 2, GNX_SB_ACCUMULATE = 4, GNX_SB_ATOMIC = 8, GNX_SB_RELU = 16 };
#define GNX_SMALL_BATCH 32
typedef struct {
  const float* A;
  int64_t lda;
  const float* B;
  int64_t ldb;
  const float* bias; /* [N] or NULL */
  float* C;
  int64_t ldc;
  int32_t M, N, K;
  int32_t flags;
} gnx_small_prob;
int32_t gnx_gemm_small_batched(gnx_handle* h, int32_t nprob, const gnx_small_prob* probs); /* probs: HOST array */

/* ---- in-degree classes: PNA post-layer 0 with one effective weight per degree ------------------------------- */
/* amp/att of [3P] DegreeScalerAggregation depend on the in-degree d only, so
 *     [x | A | amp*A | att*A] W^T  =  x W0^T + A (W1 + amp(d) W2 + att(d) W3)^T  =  x W0^T + A Weff(d)^T
 * with rows grouped by d: 26 N F^2 -> 10 N F^2 FLOPs for the widest product of the layer (ref: train/models.py:445-457).
 * gnx_degree_max: max in-degree (synchronises; the host needs D = max+1 to size buffers; D <= 64 for grouping).
 * gnx_degree_classes: dperm int32[N] = node ids stably sorted by in-degree, cls_ptr int32[D+1] = class boundaries.
 * gnx_class_tiles: tile_info int32[3*(N/tile_rows + D)] = (first position in dperm, #rows, class) per tile, tiles
 * never straddle classes; ntiles int32[1] (device) = number of tiles. */
int32_t gnx_degree_max(gnx_handle* h, const int32_t* rowptr, int64_t N, int32_t* max_degree_host);
size_t gnx_degree_classes_workspace_bytes(int64_t N, int32_t D);
int32_t gnx_degree_classes(gnx_handle* h, const int32_t* rowptr, int64_t N, int32_t D, int32_t* dperm,
                           int32_t* cls_ptr, void* ws, size_t ws_bytes);
int32_t gnx_class_tiles(gnx_handle* h, const int32_t* cls_ptr, int32_t D, int32_t tile_rows, int32_t* tile_info,
                        int32_t* ntiles);
/* gnx_gemm over class tiles: workgroup t handles rows row_index[tile_info[3t] .. +tile_info[3t+1]) (gathered A rows,
 * scattered C rows) and reads segment s's B at b + class * cls_strides[s] (cls_strides: HOST int64[nseg], 0 = shared).
 * num_classes = number of weight sets behind every non-zero stride (class ids in tile_info are < num_classes).
 * Launches max_tiles workgroups; those >= *ntiles exit. */
int32_t gnx_gemm_grouped(gnx_handle* h, int32_t nseg, const gnx_gemm_seg* segs, const int64_t* cls_strides,
                         int32_t num_classes, int64_t M, int32_t N, const float* bias, const float* mask,
                         int64_t ldmask, float* C, int64_t ldc, int32_t flags, const int32_t* row_index,
                         const int32_t* tile_info, const int32_t* ntiles, int64_t max_tiles, void* ws, size_t ws_bytes);
/* The same over a tile table built with tile_rows = 96 or 128 (gnx_gemm_grouped assumes 128).  gnx_gemm_tile_rows: the
 * height to build it with for M rows x N columns -- 96 where 128-row tiles would leave the last round of the persistent
 * workgroups mostly idle (cfg-2: 640 tiles on 512 workgroup slots), else 128; results are bit-identical either way. */
int32_t gnx_gemm_tile_rows(gnx_handle* h, int64_t M, int32_t N);
int32_t gnx_gemm_grouped_rows(gnx_handle* h, int32_t nseg, const gnx_gemm_seg* segs, const int64_t* cls_strides,
                              int32_t num_classes, int64_t M, int32_t N, const float* bias, const float* mask,
                              int64_t ldmask, float* C, int64_t ldc, int32_t flags, const int32_t* row_index,
                              const int32_t* tile_info, const int32_t* ntiles, int64_t max_tiles, void* ws,
                              size_t ws_bytes, int32_t tile_rows);
/* per-class weight gradient: dW_cls[c] (stride dw_cls_stride) += sum over the rows of class c of dC[row]^T A[row]. */
int32_t gnx_gemm_wgrad_grouped(gnx_handle* h, const float* dC, int64_t lddc, const float* A, int64_t lda, int64_t M,
                               int32_t N, int32_t K, float* dW_cls, int64_t lddw, int64_t dw_cls_stride,
                               const int32_t* row_index, const int32_t* chunk_info, const int32_t* nchunks,
                               int64_t max_chunks);
/* Weff[d][o][j] = W[o][F+j] + amp(d) W[o][5F+j] + att(d) W[o][9F+j]  (W = post_nns[t][0].weight [F,13F], ld ldw);
 * gnx_pna_weff_bwd: dW[:,F:5F] += sum_d dWeff[d]; dW[:,5F:9F] += sum_d amp(d) dWeff[d]; dW[:,9F:13F] += sum_d att(d).. */
int32_t gnx_pna_weff(gnx_handle* h, const float* W, int64_t ldw, int32_t F, int32_t D, float avg_deg_log, float* Weff);
int32_t gnx_pna_weff_bwd(gnx_handle* h, const float* dWeff, int32_t F, int32_t D, float avg_deg_log, float* dW,
                         int64_t lddw);
/* gnx_pna_weff for n (layer, tower) pairs in one launch: W / Weff HOST arrays of n device pointers, avg_deg_log HOST [n] */
int32_t gnx_pna_weff_batched(gnx_handle* h, int32_t n, const float* const* W, int64_t ldw, int32_t F, int32_t D,
                             const float* avg_deg_log, float* const* Weff);
int32_t gnx_pna_weff_bwd_batched(gnx_handle* h, int32_t n, const float* const* dWeff, int32_t F, int32_t D,
                                 const float* avg_deg_log, float* const* dW, int64_t lddw);

/* ---- PNA message assembly (first pre-layer folded to node level) ------------------------------------------- */
/* h1[p,:] = relu(P[dst[p],:] + Q[src[p],:] + Te[code[p],:])  for CSR position p;  width = H.
 * ([3P] PNAConv.message: Linear(3F->F) on cat([x_i, x_j, e]) == W_i x_i + W_j x_j + (W_e e + b), then ReLU.) */
int32_t gnx_edge_combine_fwd(gnx_handle* h, const float* P, const float* Q, const float* Te, const int32_t* src,
                             const int32_t* dst, const int32_t* code, int64_t E, int32_t H, int32_t relu, float* h1);
/* given g[E,H] (already masked by relu'):  dP[i,:] = sum_{p in row i} g[p,:] ;  dQ[j,:] = sum_{p in cpos(j)} g[p,:] ;
 * dTe[R,H] += scatter-add by code (LDS-privatised, ws = gnx_table_scatter_workspace_bytes(E, R, H) or NULL).
 * dP/dQ are overwritten. */
int32_t gnx_edge_combine_bwd(gnx_handle* h, const float* g, const int32_t* rowptr, const int32_t* colptr,
                             const int32_t* cpos, const int32_t* code, int64_t N, int64_t E, int32_t H, int32_t R,
                             float* dP, float* dQ, float* dTe, void* ws, size_t ws_bytes);

/* inverted index by a small integer key (bond code): pos int32[E] = item ids stably grouped by key, ptr int32[R+1];
 * ws: gnx_degree_classes_workspace_bytes(E, R); R <= 64.  gnx_key_segment_sum: dtable[r,:] += sum_{items of key r} g[item,:]
 * as a gather-sum over contiguous index runs (no LDS atomics; (#chunks + R) x H global atomics). */
int32_t gnx_group_by_small_key(gnx_handle* h, const int32_t* keys, int64_t E, int32_t R, int32_t* pos, int32_t* ptr,
                               void* ws, size_t ws_bytes);
int32_t gnx_key_segment_sum(gnx_handle* h, const float* g, const int32_t* pos, const int32_t* key, int64_t E, int32_t H,
                            float* dtable);

/* ---- the scatter-aggregate: PNA mean|min|max|std over target nodes  (HBM-bound; the roofline kernel) ------- */
/* [3P] MultiAggregation([Mean,Min,Max,Std], mode='cat') as used by DegreeScalerAggregation
 * (aggregator list ref: train/models.py:443).  m fp32[E, T*F] in CSR order, rowptr int32[N+1].
 * A fp32[N, T, 4F]: per tower [mean || min || max || std].  Empty rows -> 0.  mean = (sequential sum in CSR order)/max(d,1);
 * std = sqrt(max(mean(x*x) - mean^2, 1e-5)) masked to 0 when <= sqrt(1e-5). */
int32_t gnx_pna_aggregate_fwd(gnx_handle* h, const float* m, const int32_t* rowptr, int64_t N, int64_t E, int32_t T,
                              int32_t F, float* A);
/* dm[E,T*F] from dA[N,T,4F]; min/max gradients split evenly over ties (scatter_reduce amin/amax backward);
 * std gradient 0 where the forward masked (the mask decision is the forward's, bit for bit).  Where it is not masked
 * the gradient is dstd (m - mean) / (n std): with GNX_OPT_STD_BWD_CENTERED (default) `std` is re-evaluated as
 * sqrt(sum (m - mean)^2 / n), which is accurate to fp32 rounding, instead of the forward's mean(x^2) - mean(x)^2 whose
 * cancellation error (~ eps mean(x^2) / (2 var) relative, up to 3e-5 just above the mask) the CPU path carries into
 * its gradient; 0 = divide by the forward's value like the CPU path does. */
int32_t gnx_pna_aggregate_bwd(gnx_handle* h, const float* dA, const float* m, const float* A, const int32_t* rowptr,
                              int64_t N, int64_t E, int32_t T, int32_t F, float* dm);

/* ---- fused PNA edge pipeline: message assembly -> pre-layer 1 -> scatter-aggregate in one kernel ---------------- */
/* [3P] PNAConv.message + DegreeScalerAggregation for pre_layers = 2 (ref: train/models.py:445-457, configs/default.py:44):
 *   h1[p] = relu(P[dst[p]] + Q[src[p]] + Te[code[p]]);  m[p] = h1[p] W1_t^T + b1_t per tower;  A[n] = mean|min|max|std of
 *   the CSR row of n -- the work of gnx_edge_combine_fwd + gnx_gemm + gnx_pna_aggregate_fwd without reading h1 or m back
 *   from HBM; h1, m and A come out bit-identical to that sequence.  h1 / m (fp32[E, T*F]) may be NULL (not kept).
 * Edge tiles: gnx_edge_tiles groups the destination nodes so that tile j = the nodes whose first CSR position lies in
 *   [tile_w j, tile_w (j+1)); tile_info int32[2 (count + 1)] = (first node, first CSR position) per tile, count =
 *   gnx_edge_tiles_count(E, tile_w).  With in-degrees <= maxdeg, tile_w = 65 - maxdeg keeps every tile within the kernel's
 *   64 message rows; a tile that exceeds them (a degree above the bound) drops rows and sets sticky range-flag bit 6
 *   (gnx_check_range).  W1 / b1: HOST arrays of T device pointers ([F,F] row-major [out,in], [F]).  F % 4 == 0, F <= 128,
 *   float operands 16-byte aligned. */
int32_t gnx_edge_tiles_count(int64_t E, int32_t tile_w);
int32_t gnx_edge_tiles(gnx_handle* h, const int32_t* rowptr, int64_t N, int64_t E, int32_t tile_w, int32_t* tile_info);
/* backward of the same stage in one pass over the message gradient ge [E, T*F] (CSR order):
 *   gh1[p] = (ge[p] W1_t) * (h1[p] > 0);  dP[n] = sum of gh1 over the CSR row of n;  dTe[c] += sum of gh1 over the positions
 *   with bond code c  (R <= 64 codes; dTe [R, T*F] is ACCUMULATED with one atomic add per (code, column) and workgroup).
 *   gh1 and dP bit-identical to gnx_gemm(mask = h1) + gnx_edge_combine_bwd; dQ stays gnx_edge_combine_bwd(dP = NULL). */
int32_t gnx_pna_edge_bwd(gnx_handle* h, const float* ge, const float* h1, const int32_t* code, const int32_t* rowptr,
                         const int32_t* tile_info, int32_t tile_w, int64_t N, int64_t E, int32_t T, int32_t F, int32_t R,
                         const float* const* W1, float* gh1, float* dP, float* dTe);
int32_t gnx_pna_edge_fwd(gnx_handle* h, const float* P, const float* Q, const float* Te, const int32_t* src,
                         const int32_t* dst, const int32_t* code, const int32_t* rowptr, const int32_t* tile_info,
                         int32_t tile_w, int64_t N, int64_t E, int32_t T, int32_t F, const float* const* W1,
                         const float* const* b1, float* h1, float* m, float* A);

/* ---- GINE message + sum aggregate (fused gather, no materialised messages) --------------------------------- */
/* [3P] GINEConv (ref: train/models.py:529-538): out[i,:] = (1+eps) x[i,:] + sum_{p in row i} relu(x[src[p],:] + Le[code[p],:]) */
int32_t gnx_gine_aggregate_fwd(gnx_handle* h, const float* x, const float* Le, const int32_t* rowptr,
                               const int32_t* src, const int32_t* code, int64_t N, int64_t E, int32_t H, float eps,
                               float* out);
/* dx[j,:] = (1+eps) dout[j,:] + sum_{p in cpos(j)} dout[dst[p],:] * (x[j,:] + Le[code[p],:] > 0);
 * dLe[R,H] (optional) += sum_p [code[p]==r] dout[dst[p],:] * (x[src[p],:] + Le[r,:] > 0); code_pos (optional) =
 * CSR positions stably grouped by code (gnx_group_by_small_key): register sums per key run instead of LDS atomics. */
int32_t gnx_gine_aggregate_bwd(gnx_handle* h, const float* dout, const float* x, const float* Le,
                               const int32_t* colptr, const int32_t* cpos, const int32_t* src, const int32_t* dst,
                               const int32_t* code, const int32_t* code_pos, int64_t N, int64_t E, int32_t H, int32_t R,
                               float eps, float* dx, float* dLe);

/* dLe only (what gnx_gine_aggregate_bwd computes when dLe != NULL), as its own call so that it can run on a side
 * stream beside the dx chain; dLe is ACCUMULATED (+=). */
int32_t gnx_gine_dle(gnx_handle* h, const float* dout, const float* x, const float* Le, const int32_t* src,
                     const int32_t* dst, const int32_t* code, const int32_t* code_pos, int64_t E, int32_t H, int32_t R,
                     float* dLe);

/* ---- contiguous segment reduce: global pool (ref: train/models.py:218-225, 587-595) ------------------------ */
enum { GNX_POOL_ADD = 0, GNX_POOL_MEAN = 1, GNX_POOL_MAX = 2 };
int32_t gnx_segment_pool_fwd(gnx_handle* h, const float* x, const int32_t* ptr, int64_t B, int32_t H, int32_t mode,
                             float* out);
int32_t gnx_segment_pool_bwd(gnx_handle* h, const float* dout, const float* x, const float* out, const int32_t* ptr,
                             int64_t B, int32_t H, int32_t mode, float* dx);

/* ---- BatchNorm1d (+ReLU) over rows (ref: train/models.py:184,212-214 and the readout mlp :186-194) ---------- */
size_t gnx_batchnorm_workspace_bytes(int64_t M, int32_t H);
/* training: batch statistics (biased var to normalise, unbiased into running_var, momentum), saves mean/rstd[H].
 * eval (training=0): uses running stats. y = relu?(gamma*(x-mean)*rstd + beta).  num_batches_tracked (device int64[1],
 * may be NULL) is incremented in training mode, as torch.nn.BatchNorm1d does before the call. */
int32_t gnx_batchnorm_fwd(gnx_handle* h, const float* x, int64_t M, int32_t H, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                          float eps, int32_t training, int32_t relu, float* y, float* save_mean, float* save_rstd,
                          void* ws, size_t ws_bytes);
/* dy masked by (y>0) when relu; dgamma/dbeta are ACCUMULATED (+=). */
int32_t gnx_batchnorm_bwd(gnx_handle* h, const float* dy, const float* x, const float* y, int64_t M, int32_t H,
                          const float* gamma, const float* save_mean, const float* save_rstd, int32_t relu,
                          float* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes);

/* ---- dropout (ref: train/models.py:177, 209; p = 0.25 in configs/pna_msigmae_7.py:40) --------------------------- */
/* y[i] = keep(i) ? x[i] / (1 - p) : 0 with keep(i) decided by Philox4x32-10 on the counter (i / 4, offset) under the
 * key `seed` (uniform u in [0,1) from 24 bits; keep iff u >= p).  Stateless: calling it again with the same
 * (seed, offset) on the upstream gradient IS the backward pass (the mask is recomputed, never stored).  In place
 * (y == x) is allowed.  torch's RNG stream cannot be reproduced; parity for p > 0 is statistical. */
int32_t gnx_dropout(gnx_handle* h, const float* x, int64_t n, float p, uint64_t seed, uint64_t offset, float* y);

/* ---- loss: APE-Huber (ref: train/models.py:89-91) + MAPE metric (:92) ---------------------------------------- */
/* out[0] = mean huber((pred-t)/t, delta), out[1] = mean |pred-t|/max(|t|,1.17e-6); dpred (optional) = d out[0]/d pred. */
int32_t gnx_huber_ape(gnx_handle* h, const float* pred, const float* target, int64_t count, float delta, float* out2,
                      float* dpred);

/* ---- optimizer step on device (ref: train/models.py:47-63) ----------------------------------- */
/* torch.optim.AdamW(amsgrad=True) single-tensor arithmetic over flat fp32 buffers of n elements; step counts from 1. */
int32_t gnx_adamw_amsgrad(gnx_handle* h, float* p, const float* g, float* m, float* v, float* vmax, int64_t n, float lr,
                          float beta1, float beta2, float eps, float weight_decay, int64_t step);
/* torch.optim.SGD(momentum=0, weight_decay=0, nesterov=False): p -= lr * g */
int32_t gnx_sgd(gnx_handle* h, float* p, const float* g, int64_t n, float lr);

/* ---- one conv layer's backward as ONE call (native launch sequence) ------------------------------------------- */
/* The backward of PNAConv (ref: train/models.py:445-457 via autograd) is ~35 launches; issued one by one from Python
 * they cost 0.3 ms of host time per layer.  gnx_pna_conv_bwd issues the same launches in the same order on the same
 * streams (bound stream; side stream 0 = weight gradients; side stream 1 = bond-table chain) from one call.  Every
 * pointer is caller-owned device memory except `params` / `grads`, HOST arrays of 4 + T*2*(pre+post) device pointers in
 * the order: edge_encoder.{weight,bias}, lin.{weight,bias}, then per tower pre_nns (w, b) x pre_layers and post_nns
 * (w, b) x post_layers.  All gradients are ACCUMULATED (+=) into `grads` (persistent in-place sinks).  Requires the
 * degree-class form of post-layer 0 (D classes, tables from gnx_degree_classes / gnx_class_tiles with 128-row tiles
 * and the weight-gradient chunks) and the by-code inverted index code_pos (gnx_group_by_small_key). */
#define GNX_PNA_MAX_LAYERS 8
#define GNX_PNA_MAX_TOWERS 8
typedef struct {
  int64_t N, E;
  int32_t T, F, pre_layers, post_layers, R, D;
  float avg_deg_log;
  int32_t merged;            /* 1: lin o last post layer was evaluated as one product with Wm (post_layers > 1) */
  int32_t acc_first;         /* 1: this layer's backward is the first of the pass: clear acc_buf before adding */
  int32_t use_side_streams;  /* 0: everything on the bound stream */
  int32_t n_h, n_z;          /* number of saved pre activations hs[] (= pre_layers) and post activations zs[] */
  /* graph structure (gnx_pack_csr, gnx_degree_classes, gnx_class_tiles, gnx_group_by_small_key) */
  const int32_t *rowptr, *colptr, *cpos, *code, *code_pos, *dperm, *tiles, *ntiles, *chunks, *nchunks;
  int64_t max_tiles, max_chunks;
  /* saved by the forward */
  const float *x, *BE, *EE, *A;               /* [N,H], [R,H], [R,F], [N,T*4F] */
  const float* hs[GNX_PNA_MAX_LAYERS];        /* [E,H] each; hs[n_h-1] = the messages */
  const float* zs[GNX_PNA_MAX_LAYERS];        /* [N,H] each */
  const float* weff[GNX_PNA_MAX_TOWERS];      /* [D,F,4F] per tower */
  const float* Wm;                            /* [H,H] merged weight (merged = 1) */
  const float* const* params;                 /* HOST array of device pointers */
  float* const* grads;                        /* HOST array of device pointers (accumulated) */
  const float* dout;                          /* [N,H] upstream gradient */
  /* temporaries, caller-allocated, contents undefined on entry */
  float* gbuf[GNX_PNA_MAX_LAYERS];            /* [N,H] each: post_layers of them (every level of the gradient chain
                                                 keeps its own buffer: queued weight gradients read them later) */
  float* dA;                                  /* [N,T*4F] */
  float* gebuf[GNX_PNA_MAX_LAYERS];           /* [E,H] each: pre_layers of them */
  float *dP, *dQ;                             /* [N,H] */
  float *dTe, *dEE;                           /* [R,H], [R,F] */
  float *dWm, *dbm;                           /* [H,H], [H] (merged = 1) */
  float* dWeff;                               /* [T,D,F,4F] */
  void* ws;                                   /* split weight images: >= gnx_pna_conv_bwd_workspace_bytes */
  size_t ws_bytes;
  float* acc_buf;                             /* [R,H] bond-embedding gradient accumulator shared by the model's layers */
  float* dx;                                  /* [N,H] out: gradient w.r.t. the layer input */
  int32_t defer_small;                        /* bit 1 (value 2): this is the LAST conv backward of the pass (nothing follows on the
                                                 main stream: its weight gradients may take every CU); bit 2 (value 4): ... and the ones queued before the
                                                 edge backward are launched there instead of at the end; bit 3 (value 8): the per-class weight
                                                 gradient of post-layer 0 is forked BEHIND the aggregate backward (a memory-bound kernel it
                                                 slows down a lot: 135 us beside it, 55 alone) instead of in front of it; bit 4 (value 16): the
                                                 batched weight gradients start in front of the dx product (matrix-bound) instead of behind it
                                                 (beside the next layer's BatchNorm backward).  bit 0 (value 1):
                                                 dTe / dEE / dWm / dbm / dWeff were ZEROED by the caller and stay alive until
                                                 gnx_pna_stack_finish, which runs every layer's 60-row bond-table chain, its
                                                 lin o last-post un-merge and its Weff gradient in a few batched launches */
  int32_t etile_w;                            /* tile width of etile_info */
  const int32_t* etile_info;                  /* gnx_edge_tiles table (NULL: three-launch edge backward); with two pre layers
                                                 the masked input gradient, dP and dTe then come from gnx_pna_edge_bwd */
} gnx_pna_bwd_args;
size_t gnx_pna_conv_bwd_workspace_bytes(int32_t T, int32_t F, int32_t D);
int32_t gnx_pna_conv_bwd(gnx_handle* h, const gnx_pna_bwd_args* args);

/* What gnx_pna_conv_bwd(defer_small = 1) left out, for ALL L layers of a model in ~6 launches (instead of ~14 per layer):
 * side stream 1: pre-layer-0 edge-slice / edge_encoder gradients and the bond-embedding gradient (atomic adds into acc_buf,
 * which the caller zeroed) from every layer's dTe; side stream 0 (behind the layers' weight gradients): lin / last-post
 * gradients from dWm / dbm, and post-layer 0's A-block gradients from dWeff.  Pointer arrays are HOST arrays; params /
 * grads hold L x (4 + T 2 (pre + post)) device pointers in gnx_pna_bwd_args' order; ones: device fp32[>= R] of 1.0. */
typedef struct {
  int32_t L, T, F, pre_layers, post_layers, R, D, merged, use_side_streams, _pad;
  const float* BE;
  float* acc_buf;
  const float* ones;
  const float* avg_deg_log;        /* HOST [L] */
  const float* const* params;      /* HOST [L * np] */
  float* const* grads;             /* HOST [L * np] */
  const float* const* EE;          /* HOST [L]: [R,F] */
  float* const* dTe;               /* HOST [L]: [R,H]  (by-code segment sums, complete on side stream 1) */
  float* const* dEE;               /* HOST [L]: [R,F]  zero-initialised */
  const float* const* dWm;         /* HOST [L]: [H,H]  (merged = 1) */
  const float* const* dbm;         /* HOST [L]: [H] */
  const float* const* dWeff;       /* HOST [L * T]: [D,F,4F] */
} gnx_pna_finish_args;
int32_t gnx_pna_stack_finish(gnx_handle* h, const gnx_pna_finish_args* args);
/* gnx_pna_weight_only for all L layers of a model in 3 launches (+ 1 per tower beyond the second): avg_deg_log HOST [L]; params HOST [L * np]; EE / Te / Wm /
 * bm HOST [L]; weff HOST [L * T] (D > 0). */
int32_t gnx_pna_weight_only_all(gnx_handle* h, int32_t L, const float* BE, int32_t R, int32_t T, int32_t F,
                                int32_t pre_layers, int32_t post_layers, int32_t D, const float* avg_deg_log,
                                const float* const* params, int32_t merged, float* const* EE, float* const* Te,
                                float* const* weff, float* const* Wm, float* const* bm);

/* the weight-only part of a PNAConv forward in one call: EE [R,F], Te [R,H], Weff(d) [D,F,4F] per tower (D > 0) and, with
 * merged = 1 (post_layers > 1), Wm [H,H] = lin_w @ blockdiag(W_last_t), bm [H] = lin_w b_last + lin_b.  `params` as in
 * gnx_pna_bwd_args; `weff` a HOST array of T device pointers. */
int32_t gnx_pna_weight_only(gnx_handle* h, const float* BE, int32_t R, int32_t T, int32_t F, int32_t pre_layers,
                            int32_t post_layers, int32_t D, float avg_deg_log, const float* const* params, int32_t merged,
                            float* EE, float* Te, float* const* weff, float* Wm, float* bm);
/* the rest of the forward in one call.  hs[i] ([E,H], pre_layers of them) and zs[i] ([N,H]: post_layers - merged of
 * them) are the activations the backward needs; P, Q [N,H] and A [N,T*4F] likewise caller-owned. */
typedef struct {
  int64_t N, E;
  int32_t T, F, pre_layers, post_layers, D, merged;
  const int32_t *rowptr, *src, *dst, *code, *dperm, *tiles, *ntiles;
  int64_t max_tiles;
  const float *x, *Te;                         /* [N,H], [R,H] */
  const float* weff[GNX_PNA_MAX_TOWERS];
  const float *Wm, *bm;                        /* merged = 1 */
  const float* const* params;                  /* HOST array of device pointers */
  float *P, *Q, *A;
  float* hs[GNX_PNA_MAX_LAYERS];
  float* zs[GNX_PNA_MAX_LAYERS];
  void* ws;                                    /* >= gnx_pna_conv_bwd_workspace_bytes(T, F, D) */
  size_t ws_bytes;
  float* out;                                  /* [N,H] */
  const int32_t* etile_info;                   /* gnx_edge_tiles table (NULL: unfused edge pipeline) */
  int32_t etile_w;                             /* its tile width */
  int32_t tile_rows;                           /* height `tiles` was built with: 96 or 128 (0 = 128); gnx_gemm_tile_rows */
} gnx_pna_fwd_args;
int32_t gnx_pna_conv_fwd(gnx_handle* h, const gnx_pna_fwd_args* args);

/* ---- small elementwise helpers used by the host module ----------------------------------------------------- */
int32_t gnx_fill(gnx_handle* h, float* p, int64_t n, float v);
/* p[i] *= v : the 1/world factor of the gradient average after the all-reduce (sum) of the flat gradient buffer
 * (DDP's averaging; ref: train/train.py:85-88 strategy="auto"). */
int32_t gnx_scale(gnx_handle* h, float* p, int64_t n, float v);
/* y[i] += a * x[i] : bias gradient of ``lin`` from the merged (lin o last post layer) product of PNAConv's backward. */
int32_t gnx_axpy(gnx_handle* h, float* y, const float* x, int64_t n, float a);

/* Second stream for work that is independent of the caller's stream (the weight gradients of a layer's backward;
 * stands in for what the reference gets from autograd's per-op CUDA streams under DDP, ref: train/train.py:85-88).
 * gnx_side_begin: the handle's own side stream waits for everything queued so far on the bound stream, and every
 *   launch until gnx_side_end goes to the side stream.  gnx_side_join: the bound stream waits for the side stream.
 * Buffers touched by side-stream launches must stay allocated until the join (the caller keeps them alive). */
int32_t gnx_side_begin(gnx_handle* h);
/* the side stream itself (created on first use), so that the caller can order OTHER work behind the weight-gradient
 * launches without a host sync: the data-parallel exchange issues a layer's slice of the gradient all-reduce on it
 * (torch.cuda.ExternalStream) while the input-gradient chain of the remaining layers keeps the main stream busy. */
int32_t gnx_side_stream(gnx_handle* h, void** hip_stream);
int32_t gnx_side_end(gnx_handle* h);
int32_t gnx_side_join(gnx_handle* h);
/* the same with an explicit side-stream index (0, 1 or 2 -- 2 is used by gnx_pna_conv_fwd / _bwd themselves; gnx_side_begin / gnx_side_join are index 0).  Stream 1 carries
 * the bond-table gradient chain of a conv layer's backward (by-code segment sum -> 60-row products), which feeds only
 * parameter gradients and the bond-embedding gradient consumed at the very end of backward. */
int32_t gnx_side_stream_n(gnx_handle* h, int32_t which, void** hip_stream);
int32_t gnx_side_begin_n(gnx_handle* h, int32_t which);
int32_t gnx_side_join_n(gnx_handle* h, int32_t which);
/* y[m,:] = clip(x[m,:], lo[:], hi[:]), NaN propagates like Tensor.clip  (pred_with_bounds, ref: train/models.py:246-253) */
int32_t gnx_clip_rows(gnx_handle* h, const float* x, int64_t M, int32_t P, const float* lo, const float* hi, float* y);

#ifdef __cplusplus
}
#endif
#endif /* GNX_H_ */

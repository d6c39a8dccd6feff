// Dense feature x weight contractions on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD).
//
// Shapes on this path are tall-skinny: M = #edges or #nodes (1e5..1e6), N = 32..512, K = 32..13*F.  Three kernels:
//   k_gemm       tiled 128 x 128 x 32, multi-segment (K-concatenated operands with per-row scale), optional row
//                gather/scatter over degree-class tiles with per-class weights.
//   k_gemm_ws    weights-stationary persistent kernel for ONE segment with K, N <= 128 (the most frequent shape):
//                the weight image stays in LDS for the workgroup's lifetime, 64-row A tiles stream through a double
//                buffer, one barrier per tile.
//   k_gemm_wgrad dW += dC^T (rs * A): contraction over the rows, split over M, fp32 atomics.
//
// Lesson baked into the structure (measured, tools/gemm_diag.py + the ISA): any runtime-conditional global load in the
// epilogue ("if (mask) v *= mask[...]", "if (accumulate) v += C[...]") makes hipcc put `s_waitcnt vmcnt(0)` in front of
// EVERY store (3 per store in the first version of this file), serialising the 64 stores of a thread and draining the
// next tile's prefetch.  The epilogue kind is therefore a template parameter, its loads are issued unconditionally
// (clamped addresses) in one batch before they are used, and there are no loads between stores.
//
// LDS images:
//   "row-k" image  T[row][k]  (A always; B when the weight is [n][k], i.e. NT): row stride 36 (tiled) / 132 (ws) floats,
//        i.e. an odd number of 16-B slots.  A lane (i = l&31, h = l>>5) fetches k = 8*kk + 4*h .. +3 with ONE ds_read_b128
//        and feeds element t to MFMA step t, i.e. MFMA step (kk,t) contracts k in {8kk+t, 8kk+4+t}: the k order inside
//        a tile is permuted identically for A and B, which only reorders an exact-fp32 sum; the odd slot stride makes
//        the 16 lanes of every ds_read_b128 group hit 16 distinct 16-byte slots (MI355X_MICROARCH.md, LDS table).
//   "k-row" image  T[k][col]  (B when the weight is [k][n], i.e. NN; both operands of the weight gradient):
//        row stride 128 floats, read with ds_read_b32 (32 consecutive columns per half-wave: conflict-free).
#include "gnx_common.hpp"

#include <cstdlib>
#include <type_traits>

#define BM 128
#define BN 128
#define BK 32
#define LDK 36    // row stride of the row-k image
#define LDN 128   // row stride of the k-row image
#define MAX_SEGS 4

enum { EPI_PLAIN = 0, EPI_ACCUM = 1, EPI_MASK = 2 };

struct seg_dev {
  const float* a;
  const float* rs;
  const float* b;
  int64_t lda;
  int64_t ldb;
  int64_t cls_stride;  // grouped mode: B of class c starts at b + c * cls_stride (0 = class-independent)
  int k;
  int vec_a;  // 1: float4 loads allowed on A (16B-aligned rows)
  int vec_b;
};

struct gemm_args {
  seg_dev seg[MAX_SEGS];
  int nseg;
  int64_t M;
  int N;
  const float* bias;
  const float* mask;
  int64_t ldmask;
  float* C;
  int64_t ldc;
  int relu;
  // grouped mode (degree classes): workgroup b handles rows row_index[tile_info[3b] .. +tile_info[3b+1]) with the
  // class-tile_info[3b+2] weights; workgroups >= *ntiles exit.  NULL tile_info = plain row tiles.
  const int* row_index;
  const int* tile_info;
  const int* ntiles;
  int ny;  // column tiles (k_gemm3's 1-D grid)
  const __bf16* bsplit;  // k_gemm3: split weight images (k_split_weights)
  int Npad, Kpad;
  int koff[MAX_SEGS];  // padded k offset of every segment inside the images
  int steps;           // K-tiles per output tile
};

__device__ __forceinline__ f32x4 ld4(const float* p, bool vec, int valid) {
  // valid = number of in-range elements (0..4) starting at p
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (valid >= 4 && vec) {
    v = *reinterpret_cast<const f32x4*>(p);
  } else {
    if (valid > 0) v.x = p[0];
    if (valid > 1) v.y = p[1];
    if (valid > 2) v.z = p[2];
    if (valid > 3) v.w = p[3];
  }
  return v;
}

// Epilogue of one 32x32 accumulator tile (C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
// rows[r] = global row of accumulator register r (-1 = padding); every lane of a half-wave shares it.
template <int EPI>
__device__ __forceinline__ void epilogue_tile(const f32x16& acc, const int (&rows)[16], int gc, int N, float bv,
                                              int relu, const float* __restrict__ mask, int64_t ldmask,
                                              float* __restrict__ C, int64_t ldc) {
  const bool col_ok = gc < N;
  const int gcc = col_ok ? gc : 0;
  float extra[16];
  if constexpr (EPI != EPI_PLAIN) {
    // one batch of unconditional loads (clamped rows), one wait, then the stores
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t gr = rows[r] < 0 ? 0 : rows[r];
      extra[r] = (EPI == EPI_MASK) ? mask[gr * ldmask + gcc] : C[gr * ldc + gcc];
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = acc[r] + bv;
    if constexpr (EPI == EPI_ACCUM) v += extra[r];
    v = relu ? fmaxf(v, 0.f) : v;
    if constexpr (EPI == EPI_MASK) v = (extra[r] > 0.f) ? v : 0.f;
    if (col_ok && rows[r] >= 0) C[(int64_t)rows[r] * ldc + gc] = v;
  }
}

// Row scale of segments that have none (pointer-select, no branch).  A __device__ (global address space) variable on
// purpose: selecting between a kernel-argument pointer and a __constant__ address yields a GENERIC pointer, i.e. a
// flat_load, and one outstanding flat load turns every counted s_waitcnt vmcnt(N) of the loop into vmcnt(0).
__device__ float c_one = 1.0f;
__device__ float c_zero = 0.0f;
__device__ __attribute__((aligned(16))) float c_zero4[4] = {0.f, 0.f, 0.f, 0.f};

// VEC = every operand is 16-B aligned with leading dimensions / k / class strides multiples of 4 (always true for the
// model's shapes): loads are unconditional float4 from clamped addresses + a select, so all 8 loads of a K-tile are in
// flight together.  VEC = false keeps the generic element-wise path (odd shapes; correctness only).
template <bool B_TRANS, int EPI, bool VEC>
__global__ void __launch_bounds__(256, 2) k_gemm(gemm_args g) {
  __shared__ __attribute__((aligned(16))) float As[BM * LDK];
  __shared__ __attribute__((aligned(16))) float Bs[B_TRANS ? BN * LDK : BK * LDN];
  __shared__ int rid[BM];  // global row of every tile row (-1 = padding)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * BN;
  int cls = 0;
  if (g.tile_info != nullptr) {
    if ((int)blockIdx.x >= g.ntiles[0]) return;
    const int p0 = g.tile_info[3 * blockIdx.x], pr = g.tile_info[3 * blockIdx.x + 1];
    cls = g.tile_info[3 * blockIdx.x + 2];
    if (tid < BM) rid[tid] = (tid < pr) ? g.row_index[p0 + tid] : -1;
  } else {
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    if (tid < BM) rid[tid] = (m0 + tid < g.M) ? (int)(m0 + tid) : -1;
  }
  __syncthreads();

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging registers
  f32x4 ra[4], rb[4];

  // loader geometry for the row-k image: 8 lanes cover one 128-B row, 32 rows per pass, 4 passes
  const int lr = tid >> 3;        // 0..31
  const int lk = (tid & 7) * 4;   // 0..28
  // loader geometry for the k-row image (NN B): 32 lanes cover 128 columns, 8 k-rows per pass, 4 passes
  const int br = tid >> 5;        // 0..7
  const int bc = (tid & 31) * 4;  // 0..124
  int grow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) grow[i] = rid[lr + 32 * i];
  int okmask = 0;   // VEC: validity bits of ra[0..3] (bits 0-3) and rb[0..3] (bits 4-7) of the tile in flight
  float rsv[4] = {1.f, 1.f, 1.f, 1.f};

  auto load_tile = [&](const seg_dev& s, int k0) {
    const float* sb = s.b + (int64_t)cls * s.cls_stride;
    if constexpr (VEC) {
      // raw loads only: the select / row scale are applied in store_tile, so nothing consumes the loaded registers
      // before the MFMAs of the current tile have been issued (a select here makes hipcc wait for the loads at once)
      const int kcol = k0 + lk;
      const bool k_ok = kcol < s.k;
      const int kc = k_ok ? kcol : 0;
      okmask = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row = grow[i] >= 0 ? grow[i] : 0;
        ra[i] = *reinterpret_cast<const f32x4*>(s.a + row * s.lda + kc);
        rsv[i] = *(s.rs != nullptr ? s.rs + row : &c_one);
        okmask |= (k_ok && grow[i] >= 0) ? (1 << i) : 0;
      }
      if (B_TRANS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int gn = n0 + lr + 32 * i;
          rb[i] = *reinterpret_cast<const f32x4*>(sb + (int64_t)(gn < g.N ? gn : 0) * s.ldb + kc);
          okmask |= (k_ok && gn < g.N) ? (16 << i) : 0;
        }
      } else {
        const int nn = n0 + bc;
        const bool n_ok = nn < g.N;
        const int nc = n_ok ? nn : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int kr = k0 + br + 8 * i;
          rb[i] = *reinterpret_cast<const f32x4*>(sb + (int64_t)(kr < s.k ? kr : 0) * s.ldb + nc);
          okmask |= (n_ok && kr < s.k) ? (16 << i) : 0;
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int64_t gm = grow[i];
      int kv = s.k - (k0 + lk);
      kv = gm >= 0 ? kv : 0;
      f32x4 v = ld4(s.a + gm * s.lda + k0 + lk, s.vec_a, kv);
      if (s.rs != nullptr && kv > 0) {
        float sc = s.rs[gm];
        v.x *= sc;
        v.y *= sc;
        v.z *= sc;
        v.w *= sc;
      }
      ra[i] = v;
    }
    if (B_TRANS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int r = lr + 32 * i;
        int gn = n0 + r;
        int kv = s.k - (k0 + lk);
        kv = gn < g.N ? kv : 0;
        rb[i] = ld4(sb + (int64_t)gn * s.ldb + k0 + lk, s.vec_b, kv);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int kr = br + 8 * i;
        int nv = g.N - (n0 + bc);
        nv = (k0 + kr) < s.k ? nv : 0;
        rb[i] = ld4(sb + (int64_t)(k0 + kr) * s.ldb + n0 + bc, s.vec_b, nv);
      }
    }
  };

  auto store_tile = [&]() {
    if constexpr (VEC) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = ra[i];
        v.x *= rsv[i];
        v.y *= rsv[i];
        v.z *= rsv[i];
        v.w *= rsv[i];
        ra[i] = ((okmask >> i) & 1) ? v : z;
        rb[i] = ((okmask >> (4 + i)) & 1) ? rb[i] : z;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int r = lr + 32 * i;
      *reinterpret_cast<f32x4*>(&As[r * LDK + lk]) = ra[i];
    }
    if (B_TRANS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int r = lr + 32 * i;
        *reinterpret_cast<f32x4*>(&Bs[r * LDK + lk]) = rb[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int kr = br + 8 * i;
        *reinterpret_cast<f32x4*>(&Bs[kr * LDN + bc]) = rb[i];
      }
    }
  };

  // flattened (segment, k0) tile list
  int s_idx = 0, k0 = 0;
  load_tile(g.seg[0], 0);
  bool more = true;
  while (more) {
    __syncthreads();  // previous compute finished reading LDS
    store_tile();
    __syncthreads();
    // advance and prefetch the next tile into registers
    k0 += BK;
    if (k0 >= g.seg[s_idx].k) {
      ++s_idx;
      k0 = 0;
    }
    more = s_idx < g.nseg;
    if (more) load_tile(g.seg[s_idx], k0);

#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 a[2], b[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        a[mi] = *reinterpret_cast<const f32x4*>(&As[(wm * 64 + mi * 32 + li) * LDK + kk * 8 + 4 * lh]);
      if (B_TRANS) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          b[ni] = *reinterpret_cast<const f32x4*>(&Bs[(wn * 64 + ni * 32 + li) * LDK + kk * 8 + 4 * lh]);
      } else {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const float* p = &Bs[(kk * 8 + 4 * lh) * LDN + wn * 64 + ni * 32 + li];
          b[ni].x = p[0];
          b[ni].y = p[LDN];
          b[ni].z = p[2 * LDN];
          b[ni].w = p[3 * LDN];
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][t], b[ni][t], acc[mi][ni], 0, 0, 0);
    }
  }

#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    int rows[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rows[r] = rid[wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int gc = n0 + wn * 64 + ni * 32 + li;
      const float bv = (g.bias != nullptr && gc < g.N) ? g.bias[gc] : 0.f;
      epilogue_tile<EPI>(acc[mi][ni], rows, gc, g.N, bv, g.relu, g.mask, g.ldmask, g.C, g.ldc);
    }
  }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---------------------------------------------------------------------------------------------------------------
// Weights-stationary persistent GEMM: ONE segment, K <= 128, N <= 128, M = #nodes or #edges.
//   LDS: B image [128][132] (NT) or [128][128] (NN) + 2 x A image [64][132]  = 135 KB (dynamic), 1 workgroup / CU.
//   512 threads: wave w owns the 32x32 MFMA tile at rows (w>>2)*32, columns (w&3)*32 (16 accumulators); two waves per
//   SIMD cover each other's LDS-read latency and epilogue.
// One workgroup per CU keeps the whole weight image in LDS and streams 64-row A tiles through a double buffer: the next
// tile's global loads and the previous tile's stores are in flight while the MFMAs of the current tile run, with ONE
// barrier per tile and none inside the K loop.
// ---------------------------------------------------------------------------------------------------------------
#define WS_BM 64
#define WS_LD 132  // 33 x 16 B: odd -> the 16 lanes of every ds_read_b128 group hit distinct 16-B slots

struct ws_args {
  const float* A;
  int64_t lda;
  const float* B;
  int64_t ldb;
  int64_t M;
  int N, K;
  const float* bias;
  const float* mask;
  int64_t ldmask;
  float* C;
  int64_t ldc;
  int relu;
  int ntiles;
};

// full-tile epilogue of one 32x32 accumulator tile: no row checks, one base pointer, constant row strides
template <int EPI>
__device__ __forceinline__ void epilogue_tile_full(const f32x16& acc, int64_t row0, int gc, int N, float bv, int relu,
                                                   const float* __restrict__ mask, int64_t ldmask,
                                                   float* __restrict__ C, int64_t ldc) {
  if (gc >= N) return;
  float* cp = C + row0 * ldc + gc;
  float extra[16];
  if constexpr (EPI != EPI_PLAIN) {
    const float* ep = (EPI == EPI_MASK) ? mask + row0 * ldmask + gc : cp;
    const int64_t lde = (EPI == EPI_MASK) ? ldmask : ldc;
#pragma unroll
    for (int r = 0; r < 16; ++r) extra[r] = ep[((r & 3) + 8 * (r >> 2)) * lde];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = acc[r] + bv;
    if constexpr (EPI == EPI_ACCUM) v += extra[r];
    v = relu ? fmaxf(v, 0.f) : v;
    if constexpr (EPI == EPI_MASK) v = (extra[r] > 0.f) ? v : 0.f;
    cp[((r & 3) + 8 * (r >> 2)) * ldc] = v;
  }
}

template <bool B_TRANS, int EPI>
__global__ void __launch_bounds__(512, 1) k_gemm_ws(ws_args g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Bs = lds;
  float* As = lds + 128 * WS_LD;  // two buffers of WS_BM * WS_LD
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;       // 8 waves: two per SIMD hide each other's LDS / epilogue latency
  const int wr = (wave >> 2) * 32, wc = (wave & 3) * 32;  // one 32x32 MFMA tile per wave
  const int li = lane & 31, lh = lane >> 5;
  const int kgroups = (g.K + 7) >> 3;

  // ---- weight image, once per workgroup (zero-filled to 128 x 128); addresses clamped, values selected
  {
    const int r = tid >> 5, c4 = (tid & 31) * 4;  // 16 rows per pass, 8 passes
    const int rmax = B_TRANS ? g.N : g.K, cmax = B_TRANS ? g.K : g.N;
    const bool c_ok = c4 < cmax;
    const int cc = c_ok ? c4 : 0;
#pragma unroll 4
    for (int i = 0; i < 8; ++i) {
      const int rr = r + 16 * i;
      const bool ok = c_ok && rr < rmax;
      f32x4 v = *reinterpret_cast<const f32x4*>(g.B + (int64_t)(rr < rmax ? rr : 0) * g.ldb + cc);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      v = ok ? v : z;
      *reinterpret_cast<f32x4*>(&Bs[rr * (B_TRANS ? WS_LD : 128) + c4]) = v;
    }
  }

  // ---- A tile loader: 32 lanes cover one 512-B row, 16 rows per pass, 4 passes; clamped addresses, no branches
  const int ar = tid >> 5, ak = (tid & 31) * 4;
  const bool ak_ok = ak < g.K;
  const int akc = ak_ok ? ak : 0;
  f32x4 ra[4];
  int okmask = 0;  // validity of ra[] of the tile in flight; applied at LDS-store time so the loads stay in flight
  auto load_a = [&](int tile) {
    const int64_t m0 = (int64_t)tile * WS_BM;
    okmask = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t gm = m0 + ar + 16 * i;
      ra[i] = *reinterpret_cast<const f32x4*>(g.A + (gm < g.M ? gm : g.M - 1) * g.lda + akc);
      okmask |= (ak_ok && gm < g.M) ? (1 << i) : 0;
    }
  };
  auto store_a = [&](float* buf) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(&buf[(ar + 16 * i) * WS_LD + ak]) = ((okmask >> i) & 1) ? ra[i] : z;
  };

  const int gc = wc + li;
  const float bv = (g.bias != nullptr && gc < g.N) ? g.bias[gc] : 0.f;

  int tile = blockIdx.x;  // grid <= ntiles
  int cur = 0;
  load_a(tile);
  store_a(As);
  __syncthreads();

  while (tile < g.ntiles) {
    const int next = tile + gridDim.x;
    const bool has_next = next < g.ntiles;
    if (has_next) load_a(next);  // wave-uniform branch; loads stay in flight during the MFMAs below
    const int64_t m0 = (int64_t)tile * WS_BM;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* Ap = As + cur * (WS_BM * WS_LD) + (wr + li) * WS_LD + 4 * lh;
    const float* Bp = B_TRANS ? Bs + (wc + li) * WS_LD + 4 * lh : Bs + (4 * lh) * 128 + wc + li;
    auto frag_b = [&](int kk) {
      f32x4 b;
      if (B_TRANS) {
        b = *reinterpret_cast<const f32x4*>(Bp + kk * 8);
      } else {
        const float* p = Bp + kk * 8 * 128;
        b.x = p[0];
        b.y = p[128];
        b.z = p[256];
        b.w = p[384];
      }
      return b;
    };
    // software-pipelined fragment reads: the reads of group kk+1 are issued before the MFMAs of group kk
    f32x4 a0 = *reinterpret_cast<const f32x4*>(Ap), b0 = frag_b(0);
    for (int kk = 0; kk < kgroups; kk += 2) {
      const int k1 = (kk + 1 < kgroups) ? kk + 1 : kk;
      f32x4 a1 = *reinterpret_cast<const f32x4*>(Ap + k1 * 8), b1 = frag_b(k1);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc, 0, 0, 0);
      const int k2 = (kk + 2 < kgroups) ? kk + 2 : kk;
      a0 = *reinterpret_cast<const f32x4*>(Ap + k2 * 8);
      b0 = frag_b(k2);
      if (kk + 1 < kgroups) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc, 0, 0, 0);
      }
    }

    // epilogue (stores stay in flight under the next tile's MFMAs)
    if (m0 + WS_BM <= g.M) {
      epilogue_tile_full<EPI>(acc, m0 + wr + 4 * lh, gc, g.N, bv, g.relu, g.mask, g.ldmask, g.C, g.ldc);
    } else {
      int rows[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t gr = m0 + wr + (r & 3) + 8 * (r >> 2) + 4 * lh;
        rows[r] = gr < g.M ? (int)gr : -1;
      }
      epilogue_tile<EPI>(acc, rows, gc, g.N, bv, g.relu, g.mask, g.ldmask, g.C, g.ldc);
    }

    if (has_next) store_a(As + (cur ^ 1) * (WS_BM * WS_LD));
    __syncthreads();
    cur ^= 1;
    tile = next;
  }
}

template <bool BT, int EPI>
static hipError_t ws_launch_one(gnx_handle* h, const ws_args& g, int grid, size_t lds_bytes) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_ws<BT, EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((k_gemm_ws<BT, EPI>), dim3(grid), dim3(512), lds_bytes, h->stream, g);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Split-operand weights-stationary GEMM (the default for the shapes k_gemm_ws covers).
//
// fp32 MFMA runs at 1/16 of the bf16 MFMA rate on CDNA4, so the exact-fp32 kernel above is matrix-core bound at ~55 %
// of the fp32 peak while HBM idles.  Here every fp32 operand is written EXACTLY as a sum of three bf16 values
// (x = x1 + x2 + x3: round-to-nearest 8-bit pieces of a 24-bit significand, remainders exact) and the product is
//     a.b ~= a1.b1 + (a1.b2 + a2.b1 + a2.b2 + a1.b3 + a3.b1)          six v_mfma_f32_32x32x16_bf16 = 6/16 of the
// fp32-MFMA time.  Every bf16 x bf16 product is exact in fp32 and the accumulation is fp32; the three dropped terms
// (a2.b3, a3.b2, a3.b3) are <= 2^-25 |a||b|, i.e. below the rounding of the fp32 accumulation itself.  The leading
// term and the five correction terms use separate accumulators (added once at the end) so the corrections are not
// absorbed by the rounding of the large partial sum.  Inf operands give NaN (inf - inf in the remainder).
//
//   * B (<= 128 x 128 weights) lives in REGISTERS: wave w owns output columns (w&3)*32.., its B fragments for all
//     eight 16-deep k-slabs are 3 x 8 x 4 = 96 VGPRs, split once per launch.  No LDS traffic for B at all.
//   * A: 64-row tiles, split when staged: LDS holds three bf16 images [64][128 (+8 pad)] per buffer, two buffers
//     (102 KB).  Row stride 272 B = 17 x 16 B (odd) -> conflict-free ds_read_b128 fragments.
//   * per 16-deep slab a wave issues 3 ds_read_b128 and 6 MFMAs; the kernel is HBM-bound (A in, C out): the marginal
//     cost of 64 more rows per CU is 4.1 us at every size from 2 to 20 tiles per CU (= 4 TB/s of mixed read + write
//     traffic), and neither a second tile of loads in flight (hand-counted asm loads) nor software-pipelined fragment
//     reads moved it; a launch has ~8 us of fixed cost (weight fragments, first tile, drain) on top.
//     Round 2 (same-box A/B, tools/ws3_scaling.py): two register stages of A loads (tiles t+G and t+2G in flight, 64 KB
//     per CU) and a branch-free 8-slab loop for K = 128 (the tile's 48 MFMAs and 24 fragment reads in ONE basic block
//     instead of eight) each changed nothing (25.1 vs 25.2-26.4 us at 81 920 rows, 42-44 vs 40-43 us at 163 840): the
//     kernel moves 64 KB per 64-row tile and CU in 4.1-4.5 us = 14-16 GB/s per CU of mixed read + write traffic, and
//     that -- not latency, issue order or the matrix pipe (96 MFMAs per SIMD and tile = 1.3-1.7 us) -- is its bound.
//     Also measured and rejected: the same kernel as TWO 256-thread workgroups per CU (wave = 64 rows x 32 columns, one LDS
//     stage each, two barriers per tile; built on the idea that the eight waves of this workgroup are barrier-locked into
//     the same phase): bit-identical, 26.7 vs 25.2 us at 81 920 rows, 91.5 vs 87.7 us at 327 680.  With the C stores
//     removed this kernel needs 3.3 us per tile (67 vs 86 us at 327 680 rows), i.e. the on-CU work of a tile -- 96 MFMAs
//     per SIMD (1.3-1.7 us) + the split of 2 x 16 floats per lane (0.6 us) + 48 KB of LDS stores + 192 KB of fragment
//     reads -- adds up whoever issues it: tools/ubench/wave_specialised_overlap.hip shows the same for different waves of one
//     SIMD (0.91 us of MFMAs + 0.84 us of split and LDS stores -> 1.41-1.46 us together).  The split-operand formulation
//     itself, not this kernel's structure, sets the per-tile time; what would remove work is a producer that writes the
//     three bf16 images instead of fp32 (no split, half the staging LDS traffic in every consumer).
// ---------------------------------------------------------------------------------------------------------------
#include "gnx_split.hpp"

//
// Round 3, measured and rejected: the split of tile t + G issued BETWEEN the MFMA slabs of tile t (two register stages of
// loads, the k_gemm3p scheme; bit-identical): 27.4 vs 27.2 us at 81 920 rows, 42.3 vs 43.6 at 163 840, +1.5 us at 8 192
// rows, step 6.87 vs 6.91 ms with / without the predicate-free form either way -- the split is not what bounds a tile.
#ifdef WS_STAMP  // diagnostic build only (tools/ubench/gemm_ws3_stamp.hip): phase cycle sums of wave 0 per workgroup
__device__ unsigned long long* ws_stamp_buf = nullptr;
#define WS_AT(i)                                     \
  do {                                               \
    __builtin_amdgcn_sched_barrier(0);               \
    const unsigned long long tn_ = clock64();        \
    tacc[i] += tn_ - tprev;                          \
    tprev = tn_;                                     \
    __builtin_amdgcn_sched_barrier(0);               \
  } while (0)
#else
#define WS_AT(i)
#endif
//
// FAST (N == 128, 16-byte aligned C / mask, no accumulate): a predicate-free tile loop.  The result tile is transposed
// 4 x 4 inside every quad of lanes (two DPP exchange stages: lane = column, four rows per register group -> lane = row,
// four columns) and leaves as four 16-byte stores per lane, 8 full 128-byte lines per wave instruction; a partial last
// tile repeats row M - 1 (same inputs -> same value to the same address) and the next tile's loads are always issued
// (past the end: the current tile again), so no branch sits between the loads of tile t + G and their first use and the
// wait in front of the split is an exact vmcnt(stores issued since) -- with the predicated 4-byte epilogue hipcc could
// not count the stores and drained ALL of them (write acknowledgements, ~1 us) before every split.
// NW = waves per workgroup: 8 (64-row tiles, one workgroup per CU) or 4 (32-row tiles, TWO workgroups per CU with their
// own barriers, so that one can issue loads / stores while the other multiplies: in-kernel stamps of the 8-wave form give
// a tile 6.2 k cycles = loads 0.8 k + fragment reads and 48 MFMAs 2.2 k + epilogue 1.2 k + split 0.7 k + barrier 1.3 k, all
// eight waves in the same phase -- tools/ubench/gemm_ws3_stamp.hip).  The same arithmetic per row either way.
template <bool B_TRANS, int EPI, bool FAST, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) k_gemm_ws3(ws_args g) {
  constexpr int RS = 2 * NW;               // rows per loader pass (32 lanes per row)
  constexpr int TM = 4 * RS;               // rows per tile: 64 / 32
  constexpr int PIECE = TM * W3_LDB;       // one bf16 image of a tile
  constexpr int BUF = 3 * PIECE;           // the three images
  extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = (wave >> 2) * 32, wc = (wave & 3) * 32;
  const int li = lane & 31, lh = lane >> 5;
  const int nslab = (g.K + 15) >> 4;
  const int gc = wc + li;

  // ---- A tile loader: 32 lanes cover one 512-B row with consecutive float4 (whole cache lines per wave instruction:
  // a wave load touches 2 rows = 8 lines; 16 lanes x 2 float4 per row touched 16 half-used lines per instruction and
  // was bound by line operations in the texture path), 16 rows per pass, 4 passes
  const int ar = tid >> 5, ak = (tid & 31) * 4;
  const bool ak_ok = ak < g.K;
  const int akc = ak_ok ? ak : 0;
  f32x4 ra[4];
  int okmask = 0;
  auto load_a = [&](int tile) {
    const int64_t m0 = (int64_t)tile * TM;
    okmask = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t gm = m0 + ar + RS * i;
      ra[i] = *reinterpret_cast<const f32x4*>(g.A + (gm < g.M ? gm : g.M - 1) * g.lda + akc);
      okmask |= (ak_ok && (FAST || gm < g.M)) ? (1 << i) : 0;  // FAST: rows past the end repeat row M - 1
    }
  };
  typedef __attribute__((ext_vector_type(2))) float f32x2;  // 8-byte LDS word (four bf16)
  auto store_a = [&](unsigned char* buf) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // rows ar + 2 RS h and ar + 2 RS h + RS: four k each -> one split3 of 8 values
      float x[8];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) x[4 * q + j] = ((okmask >> (2 * h + q)) & 1) ? ra[2 * h + q][j] : 0.f;
      bf16x8 pc[3];
      split3(x, pc[0], pc[1], pc[2]);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(&pc[p]);
        unsigned char* q = buf + p * PIECE + (ar + 2 * RS * h) * W3_LDB + ak * 2;
        *reinterpret_cast<f32x2*>(q) = f32x2{w.x, w.y};
        *reinterpret_cast<f32x2*>(q + RS * W3_LDB) = f32x2{w.z, w.w};
      }
    }
  };

  const float bv = (g.bias != nullptr && gc < g.N) ? g.bias[gc] : 0.f;
  int tile = blockIdx.x;  // grid <= ntiles
  int cur = 0;
  load_a(tile);  // in flight while the weight fragments are fetched and split

  // ---- this wave's B fragments, split once: lane holds column gc, k = 16 s + 8 lh + j
  bf16x8 b1[8], b2[8], b3[8];
  {
    const bool n_ok = gc < g.N;
    const int gcc = n_ok ? gc : 0;
    if (!B_TRANS) {
      // W[k][n]: stage the zero-filled 128 x 128 image in LDS with coalesced float4 loads (the A buffers are not in
      // use yet); the fragment below is then a conflict-free column read instead of 64 strided global loads per lane
      float* wl = reinterpret_cast<float*>(lds3);
      const int r = tid >> 5, c4 = (tid & 31) * 4;
      const bool c_ok = c4 < g.N;
#pragma unroll
      for (int i = 0; i < 128 / RS; ++i) {
        const int rr = r + RS * i;
        f32x4 v = *reinterpret_cast<const f32x4*>(g.B + (int64_t)(rr < g.K ? rr : 0) * g.ldb + (c_ok ? c4 : 0));
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(wl + rr * 128 + c4) = (c_ok && rr < g.K) ? v : z;
      }
      __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float x[8];
      const int k0 = 16 * s + 8 * lh;
      if (B_TRANS) {  // W[n][k]: 8 consecutive floats of row gc
        const int ka = (k0 < g.K) ? k0 : 0, kb = (k0 + 4 < g.K) ? k0 + 4 : 0;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(g.B + (int64_t)gcc * g.ldb + ka);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(g.B + (int64_t)gcc * g.ldb + kb);
        const bool ok0 = n_ok && k0 < g.K, ok1 = n_ok && k0 + 4 < g.K;  // K % 4 == 0
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          x[j] = ok0 ? v0[j] : 0.f;
          x[4 + j] = ok1 ? v1[j] : 0.f;
        }
      } else {  // W[k][n]: column gc of 8 consecutive rows of the staged image
        const float* wl = reinterpret_cast<const float*>(lds3);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = wl[(k0 + j) * 128 + gc];
      }
      split3(x, b1[s], b2[s], b3[s]);
    }
    if (!B_TRANS) {
      // the image is overwritten by the first A tile.  The fragments are CONSUMED here: hipcc may otherwise sink the LDS
      // reads above to below this barrier (seen in the ISA of a software-pipelined form of this kernel: both barriers
      // back to back, the reads among the first tile's split), where they race with the other waves' stores of that tile
#pragma unroll
      for (int s = 0; s < 8; ++s) asm volatile("" ::"v"(b1[s]), "v"(b2[s]), "v"(b3[s]));
      __syncthreads();
    }
  }

  store_a(lds3);
  __syncthreads();

  // FAST epilogue geometry (see above)
  const int qj = li & 3, colq = wc + (li >> 2) * 4;
  f32x4 bv4 = {0.f, 0.f, 0.f, 0.f};
  if (FAST && g.bias != nullptr)
#pragma unroll
    for (int e = 0; e < 4; ++e) bv4[e] = g.bias[colq + e];
  const float floor_v = g.relu ? 0.f : -__builtin_inff();
  auto quad_swap = [&](float& x, float& y, auto CTRL, bool hi) {
    const float send = hi ? x : y;
    const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), decltype(CTRL)::value, 0xf, 0xf, true));
    x = hi ? got : x;
    y = hi ? y : got;
  };
  using XOR1 = std::integral_constant<int, 0xB1>;  // quad_perm [1,0,3,2]
  using XOR2 = std::integral_constant<int, 0x4E>;  // quad_perm [2,3,0,1]

#ifdef WS_STAMP
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long tprev = clock64();
  int ntile_done = 0;
#endif
  while (tile < g.ntiles) {
    const int next = tile + gridDim.x;
    const bool has_next = next < g.ntiles;
    if (FAST) {
      load_a(has_next ? next : tile);
      __builtin_amdgcn_sched_barrier(0);  // (hipcc sinks these loads below the MFMA slabs otherwise)
    } else if (has_next) {
      load_a(next);
    }
    const int64_t m0 = (int64_t)tile * TM;

    f32x16 acc, corr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[r] = 0.f;
      corr[r] = 0.f;
    }
    const unsigned char* ap = lds3 + cur * BUF + (wr + li) * W3_LDB + 16 * lh;
    WS_AT(0);  // issue of the next tile's loads
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < nslab) {
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(ap + 32 * s);
        const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(ap + 32 * s + PIECE);
        const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(ap + 32 * s + 2 * PIECE);
        corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3[s], corr, 0, 0, 0);
        corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1[s], corr, 0, 0, 0);
        corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2[s], corr, 0, 0, 0);
        corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2[s], corr, 0, 0, 0);
        corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1[s], corr, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1[s], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += corr[r];
    WS_AT(1);  // fragment reads + 48 MFMAs

    if constexpr (FAST) {
      const int64_t rbase = m0 + wr + 4 * lh + qj, rlast = g.M - 1;
      f32x4 ex[4];
      if constexpr (EPI == EPI_MASK) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int64_t gr = rbase + 8 * gq < rlast ? rbase + 8 * gq : rlast;
          ex[gq] = *reinterpret_cast<const f32x4*>(g.mask + gr * g.ldmask + colq);
        }
      }
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        float c0 = acc[4 * gq], c1 = acc[4 * gq + 1], c2 = acc[4 * gq + 2], c3 = acc[4 * gq + 3];
        quad_swap(c0, c1, XOR1{}, (qj & 1) != 0);
        quad_swap(c2, c3, XOR1{}, (qj & 1) != 0);
        quad_swap(c0, c2, XOR2{}, (qj & 2) != 0);
        quad_swap(c1, c3, XOR2{}, (qj & 2) != 0);
        f32x4 v = {c0 + bv4.x, c1 + bv4.y, c2 + bv4.z, c3 + bv4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], floor_v);
        if constexpr (EPI == EPI_MASK)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (ex[gq][e] > 0.f) ? v[e] : 0.f;
        const int64_t gr = rbase + 8 * gq < rlast ? rbase + 8 * gq : rlast;
        *reinterpret_cast<f32x4*>(g.C + gr * g.ldc + colq) = v;
      }
    } else if (m0 + TM <= g.M) {
      epilogue_tile_full<EPI>(acc, m0 + wr + 4 * lh, gc, g.N, bv, g.relu, g.mask, g.ldmask, g.C, g.ldc);
    } else {
      int rows[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t gr = m0 + wr + (r & 3) + 8 * (r >> 2) + 4 * lh;
        rows[r] = gr < g.M ? (int)gr : -1;
      }
      epilogue_tile<EPI>(acc, rows, gc, g.N, bv, g.relu, g.mask, g.ldmask, g.C, g.ldc);
    }

    WS_AT(2);  // epilogue (transpose + stores)
    if (has_next) store_a(lds3 + (cur ^ 1) * BUF);
    WS_AT(3);  // wait for the next tile's loads + split + LDS stores
    __syncthreads();
    WS_AT(4);  // barrier
    cur ^= 1;
    tile = next;
#ifdef WS_STAMP
    ++ntile_done;
#endif
  }
#ifdef WS_STAMP
  if (tid == 0 && ws_stamp_buf != nullptr) {
    for (int i = 0; i < 5; ++i) ws_stamp_buf[(size_t)blockIdx.x * 8 + i] = tacc[i];
    ws_stamp_buf[(size_t)blockIdx.x * 8 + 5] = (unsigned long long)ntile_done;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Split-operand tiled GEMM (same contract as k_gemm<.., VEC = true>: multi-segment, row scale, grouped rows; see the
// split-operand notes at k_gemm_ws3).
//   * weights: split once per call by k_split_weights into zero-padded bf16 images [class][3][n][k] (either source
//     layout), so the B tiles are plain 16-B copies global(L2) -> LDS;
//   * activations: split when a K-tile is staged (the only conversion work left in the loop, ~6 VALU ops / element);
//   * LDS  A3[3][128][32 (+8 pad)] bf16, B3[3][128 n][32 k (+8)] bf16, row stride 80 B = 5 x 16 B (odd -> conflict-free
//     ds_read_b128), 60 KB per workgroup, two workgroups per CU.
// Per 16-deep slab a wave (64 x 64 outputs) reads 12 fragments and issues 24 bf16 MFMAs (fp32: 64 MFMAs, 2x the time
// each): 6/16 of the matrix-core time of k_gemm.
// ---------------------------------------------------------------------------------------------------------------
#define G3_LDB 80                   // bytes per image row
#define G3_PIECE (128 * G3_LDB)     // 10240 B: one bf16 image of a 128 x 32 K-tile
#define G3_OP (3 * G3_PIECE)        // the three images of one operand

// Weight images for k_gemm3: every segment's B (either layout, any class) is split ONCE per call into three bf16
// images, zero padded (n to 128, every segment's k to 32, segments concatenated) and stored TILE-MAJOR
// [class][n tile][k tile][piece][128][32]: a workgroup copies the 24 KB of a K-tile global -> LDS with fully coalesced
// 16-byte loads (8 cache lines per wave instruction), no conversion work and no bounds checks.
struct split_args {
  seg_dev seg[MAX_SEGS];
  int koff[MAX_SEGS + 1];  // padded k offset of every segment; koff[nseg] = Kpad
  int nseg;
  int N, Npad, Kpad, D;
  int frag;  // 1: fragment-major blocks [slab][piece][n / 32][lane][8] (k_gemm3p), 0: [piece][128 n][32 k] (k_gemm3)
  __bf16* out;
};

template <bool B_TRANS>
__global__ void __launch_bounds__(256) k_split_weights(split_args g) {
  const int kgroups = g.Kpad >> 3;
  const int64_t total = (int64_t)g.D * g.Npad * kgroups;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  int c, n, kg;
  if (B_TRANS) {  // W[n][k]: k fastest
    kg = (int)(idx % kgroups);
    n = (int)((idx / kgroups) % g.Npad);
    c = (int)(idx / ((int64_t)kgroups * g.Npad));
  } else {  // W[k][n]: n fastest
    n = (int)(idx % g.Npad);
    kg = (int)((idx / g.Npad) % kgroups);
    c = (int)(idx / ((int64_t)kgroups * g.Npad));
  }
  const int kk = kg * 8;
  int si = 0;
#pragma unroll
  for (int q = 1; q < MAX_SEGS; ++q) si += (q < g.nseg && kk >= g.koff[q]) ? 1 : 0;
  const seg_dev& sg = g.seg[si];
  const int kl = kk - g.koff[si];
  const float* sb = sg.b + (int64_t)c * sg.cls_stride;
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool ok = n < g.N && kl + j < sg.k;
    const int64_t off = B_TRANS ? (int64_t)n * sg.ldb + kl + j : (int64_t)(kl + j) * sg.ldb + n;
    x[j] = ok ? sb[ok ? off : 0] : 0.f;
  }
  bf16x8 p1, p2, p3;
  split3(x, p1, p2, p3);
  // tile-major: [class][n tile][k tile][piece][128 n][32 k] -> the 24 KB of one (n tile, k tile) are contiguous
  const int NT = g.Npad / BN, KT = g.Kpad / BK;
  __bf16* blk = g.out + ((((int64_t)c * NT + n / BN) * KT + kk / BK) * 3) * (BN * BK);
  if (g.frag) {
    // MFMA B operand of column n, k = 16 slab + 8 half .. +7: lane = 32 half + n % 32 of the (slab, piece, n / 32) fragment
    const int nl = n % BN, kl2 = kk % BK;
    const int64_t lane_off = ((int64_t)(nl >> 5) * 64 + ((kl2 >> 3) & 1) * 32 + (nl & 31)) * 8;
    const int64_t pstride = 4 * 64 * 8, sbase = (int64_t)(kl2 >> 4) * 3 * pstride;
    *reinterpret_cast<bf16x8*>(blk + sbase + lane_off) = p1;
    *reinterpret_cast<bf16x8*>(blk + sbase + pstride + lane_off) = p2;
    *reinterpret_cast<bf16x8*>(blk + sbase + 2 * pstride + lane_off) = p3;
    return;
  }
  __bf16* o = blk + (n % BN) * BK + (kk % BK);
  *reinterpret_cast<bf16x8*>(o) = p1;
  *reinterpret_cast<bf16x8*>(o + BN * BK) = p2;
  *reinterpret_cast<bf16x8*>(o + 2 * BN * BK) = p3;
}

// Persistent workgroups (two per CU): workgroup w walks the virtual tile ids w, w + G, w + 2G, ...  A virtual id v maps
// to (row tile, column tile) so that the column tiles of one row tile get ids 8 apart = the same XCD (ids are dealt
// round-robin to the 8 XCDs and G % 8 == 0): the shared A rows hit that XCD's L2.  Row tiles grow with v, so the first
// invalid tile ends the walk.  The loads of the next tile (its row ids, then its first K-tile) are issued while the
// current tile is still multiplying, so a workgroup waits on a cold pipeline only once.
//
// Where a workgroup's time goes (s_memtime stamps around the phases, post-layer-0 shape, 25 K-tiles per workgroup on
// average): issuing the 10 loads of a K-tile 20 % (the waves stall AT the load instructions: 40 KB per K-tile through
// the CU's 64 B/clk vector-memory path, shared by both resident workgroups), split + LDS store incl. the wait for the
// data 22 %, LDS reads + MFMA issue 17 % + 7 % waiting for the other waves' MFMAs at the barrier, epilogue 6 %, tile
// switch / prologue the rest.  The phases of one workgroup are serial; only the two resident workgroups overlap.
//
// Measured dead ends on MI355X (kept out of the code, recorded here): a 512-thread double-buffered variant with
// hand-counted inline-asm loads, the same with the loads interleaved between MFMAs, and a loader/multiply
// wave-specialised variant were all correct and all 3-20 % SLOWER than this two-barrier form; their phase ablations
// were additive (load + split + MFMA time), i.e. the phases did not overlap inside one workgroup whatever the
// structure, while two independent workgroups per CU do overlap each other.
//
// Round 2 measured the pieces one by one (tools/ubench/*.hip, tools/gemm3_kslope.py, ablations of a software-pipelined
// rewrite that is NOT in the tree):
//   * the matrix pipe: 48 bf16 MFMAs take 0.65 us per wave at 2.4 GHz, 0.84-0.87 us with random operands on all 256 CUs
//     (the power management holds the clock near 1.8 GHz; <= 128 active CUs or constant operands are not throttled);
//     that is the floor of a K-tile, against 1.5-1.7 us (A inside the Infinity Cache) and 1.85-2.1 us (A streamed from
//     HBM) that a K-tile costs a CU in this kernel.  A lone workgroup finishes a tile in 34 us, two sharing a CU need
//     75 us for their two tiles: the second workgroup per CU hides epilogues and tile switches, not the K loop;
//   * the same wave can overlap its split (VALU) with its MFMAs: 48 MFMAs + the split of 32 floats take 0.78 us where
//     the parts take 0.65 + 0.60 us;
//   * a rewrite with ONE workgroup per CU, one barrier per K-tile, the split / LDS stores of K-tile j+2 and the global
//     loads of K-tile j+4 issued inside the MFMA block of K-tile j, B fragments fetched straight from a fragment-major
//     weight image into registers and a three-stage LDS ring for A reached 1.15 us per K-tile with A in the Infinity
//     Cache (MFMAs alone 0.75, + fragment reads and loads 1.06, + split and LDS stores 1.15) but 1.8-1.9 us from HBM
//     and ~20 us more per launch in prologue / epilogues that no second workgroup hides: not faster (111 vs 113-122 us
//     on post-layer 0).  Pinning the order of MFMA groups, fragment reads and LDS stores with sched_barrier and a
//     1 x 4 instead of a 2 x 2 wave grid (B read once per workgroup: L2 serves ~70 GB/s per CU, the Infinity Cache
//     ~33, HBM ~24) did not move these numbers;
//   * the same pipeline with TWO workgroups per CU (two-stage ring, no cross-barrier fragment prefetch, 218 VGPRs) is
//     k_gemm3p below: 0.9 us per K-tile and CU once both workgroups are in their K loops (the MFMA floor), 1.1 / 1.4 us
//     per K-tile over whole launches (Infinity Cache / HBM) against 1.75 / 2.0 us here.  It takes the products with
//     >= 12 K-tiles per tile; this kernel keeps the shorter ones (equal speed at 4 and 8 K-tiles, shorter prologue).
template <int EPI>
__global__ void __launch_bounds__(256, 2) k_gemm3(gemm_args g) {
  __shared__ __attribute__((aligned(16))) unsigned char A3[G3_OP];
  __shared__ __attribute__((aligned(16))) unsigned char B3[G3_OP];
  __shared__ int rid[BM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  // loader geometry.  A: 8 lanes cover one 128-byte row segment (one cache line), 32 rows per pass, 4 passes: every
  // wave instruction touches 8 whole lines (two lanes per row = 32 lines per instruction, each line hit by four
  // instructions, was texture-path bound).  B: the 24 KB K-tile of the tile-major image is read linearly.
  const int lrow = tid >> 3, lk4 = (tid & 7) * 4;
  const int NT = g.Npad / BN, KT = g.Kpad / BK;

  const int ny = g.ny, G = gridDim.x;
  const bool grouped = g.tile_info != nullptr;
  const int nrt = grouped ? g.ntiles[0] : (int)((g.M + BM - 1) / BM);
  auto row_tile_of = [&](int v) { return (v / (8 * ny)) * 8 + ((v % (8 * ny)) & 7); };
  auto n0_of = [&](int v) { return ((v % (8 * ny)) >> 3) * BN; };

  // tile context: cur = tile being multiplied, nxt = the following one (row ids in flight), ti2 = tile_info of the
  // one after that (two-level dependency tile_info -> row_index, each level fetched one tile ahead)
  int v_cur = blockIdx.x;
  if (row_tile_of(v_cur) >= nrt) return;
  int grow[4], cls, n0, grow_n[4] = {-1, -1, -1, -1}, ridt_n = -1, cls_n = 0, n0_n = 0, ti2_p0 = 0, ti2_pr = 0, ti2_cls = 0;
  bool valid_n;
  auto fetch_ti = [&](int v, int& p0, int& pr, int& c) {  // tile_info triple (clamped; callers check validity)
    const int rt = row_tile_of(v);
    const int rtc = rt < nrt ? rt : 0;
    p0 = g.tile_info[3 * rtc];
    pr = g.tile_info[3 * rtc + 1];
    c = g.tile_info[3 * rtc + 2];
  };
  auto fetch_rows = [&](int v, int p0, int pr, int c, int (&gr)[4], int& rt_id, int& cl) {
    const int t = tid < BM ? tid : 0;
    if (grouped) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = lrow + 32 * q;
        gr[q] = g.row_index[p0 + (r < pr ? r : 0)];
        gr[q] = r < pr ? gr[q] : -1;
      }
      rt_id = g.row_index[p0 + (t < pr ? t : 0)];
      rt_id = t < pr ? rt_id : -1;
      cl = c;
    } else {
      const int64_t m0 = (int64_t)row_tile_of(v) * BM;
#pragma unroll
      for (int q = 0; q < 4; ++q) gr[q] = (m0 + lrow + 32 * q < g.M) ? (int)(m0 + lrow + 32 * q) : -1;
      rt_id = (m0 + t < g.M) ? (int)(m0 + t) : -1;
      cl = 0;
    }
  };
  {
    int p0 = 0, pr = 0, c = 0, rt_id;
    if (grouped) fetch_ti(v_cur, p0, pr, c);
    fetch_rows(v_cur, p0, pr, c, grow, rt_id, cls);
    n0 = n0_of(v_cur);
    if (tid < BM) rid[tid] = rt_id;
    valid_n = row_tile_of(v_cur + G) < nrt;
    if (valid_n) {
      if (grouped) fetch_ti(v_cur + G, p0, pr, c);
      fetch_rows(v_cur + G, p0, pr, c, grow_n, ridt_n, cls_n);
      n0_n = n0_of(v_cur + G);
    }
    if (grouped) fetch_ti(v_cur + 2 * G, ti2_p0, ti2_pr, ti2_cls);
  }
  __syncthreads();

  // one accumulator per 32x32 tile (the register budget of a 128x128 tile does not allow k_gemm_ws3's separate
  // correction accumulators): 6 roundings per 16 k instead of the 16 of the fp32 MFMA chain
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[4];   // A K-tile of the next stage (fp32, split at store time): rows lrow + 32 q, k lk4..+3
  f32x4 rb[6];   // B K-tile of the next stage: six 16-byte words of the contiguous [3][128][32] bf16 block
  int kvalid_a = 0;  // per-pass validity bits of ra, applied at store time

  // Load cursor state kept in registers: indexing g.seg[] / g.koff[] with a run-time index is a kernarg load + wait,
  // and the row byte offsets are 64-bit multiplies -- both are paid when the segment (or tile) changes, not per K-tile
  // (measured with s_memtime: the "issue the loads" phase was 24 % of the workgroup's lifetime before this).
  const float* rowp[4];   // this thread's four A rows in the current segment (k = 0)
  int rowv = 0;           // validity bits of the four rows
  int segK = 0;
  const __bf16* bptr = nullptr;  // this thread's 16-byte slot in the B block of the cursor's K-tile
  auto enter_segment = [&](int s_i, const int (&gr)[4]) {
    const seg_dev& s = g.seg[s_i];
    rowv = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      rowp[q] = s.a + (int64_t)(gr[q] >= 0 ? gr[q] : 0) * s.lda;
      rowv |= gr[q] >= 0 ? (1 << q) : 0;
    }
    segK = s.k;
  };
  auto enter_tile = [&](const int (&gr)[4], int cl, int nn0) {
    enter_segment(0, gr);
    bptr = g.bsplit + (((int64_t)cl * NT + nn0 / BN) * KT * 3) * (BN * BK) + tid * 8;
  };
  auto load_ab = [&](int k0) {  // K-tile at (current segment, k0); the B blocks of a tile are consecutive
    const int k = k0 + lk4;
    const bool k_ok = k < segK;
    const int kc = k_ok ? k : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) ra[q] = *reinterpret_cast<const f32x4*>(rowp[q] + kc);
    kvalid_a = k_ok ? rowv : 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) rb[j] = *reinterpret_cast<const f32x4*>(bptr + j * 2048);
    bptr += 3 * (BN * BK);
  };

  typedef __attribute__((ext_vector_type(2))) float f32x2;  // 8-byte LDS word (four bf16)
  auto store_tile = [&]() {
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // rows (lrow + 64 h, lrow + 64 h + 32): four k each -> one split3 of 8 values
      float xa[8];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) xa[4 * q + j] = ((kvalid_a >> (2 * h + q)) & 1) ? ra[2 * h + q][j] : 0.f;
      bf16x8 pc[3];
      split3(xa, pc[0], pc[1], pc[2]);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(&pc[p]);
        unsigned char* qa = A3 + p * G3_PIECE + (lrow + 64 * h) * G3_LDB + lk4 * 2;
        *reinterpret_cast<f32x2*>(qa) = f32x2{w.x, w.y};
        *reinterpret_cast<f32x2*>(qa + 32 * G3_LDB) = f32x2{w.z, w.w};
      }
    }
    // B: chunk c = j * 256 + tid of the contiguous [piece][128][32] block -> piece j / 2, row (j & 1) * 64 + tid / 4
#pragma unroll
    for (int j = 0; j < 6; ++j)
      *reinterpret_cast<f32x4*>(B3 + (j >> 1) * G3_PIECE + ((j & 1) * 64 + (tid >> 2)) * G3_LDB + (tid & 3) * 16) = rb[j];
  };

  auto compute = [&]() {
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          a[mi][p] = *reinterpret_cast<const bf16x8*>(A3 + p * G3_PIECE + (wm * 64 + mi * 32 + li) * G3_LDB + 32 * sl + 16 * lh);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          b[ni][p] = *reinterpret_cast<const bf16x8*>(B3 + p * G3_PIECE + (wn * 64 + ni * 32 + li) * G3_LDB + 32 * sl + 16 * lh);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][2], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][2], b[ni][0], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][1], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][1], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][0], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][0], acc[mi][ni], 0, 0, 0);
        }
    }
  };

  // bias of the current tile's columns (fetched at the tile switch, consumed in the epilogue)
  float bv[2];
  auto fetch_bias = [&](int nn0) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int gc = nn0 + wn * 64 + ni * 32 + li;
      bv[ni] = (g.bias != nullptr && gc < g.N) ? g.bias[gc] : 0.f;
    }
  };
  fetch_bias(n0);

  int s1 = 0, k1 = 0;  // (segment, k) of the K-tile held in ra / rb
  enter_tile(grow, cls, n0);
  load_ab(0);
  while (true) {
    __syncthreads();  // previous multiply finished reading LDS
    store_tile();
    __syncthreads();
    // advance to the next K-tile; past the last segment it is the next tile's first K-tile
    k1 += BK;
    bool last = false;
    if (k1 >= segK) {
      ++s1;
      k1 = 0;
      last = s1 >= g.nseg;
      if (!last) enter_segment(s1, grow);
    }
    if (!last) {
      load_ab(k1);
    } else if (valid_n) {
      enter_tile(grow_n, cls_n, n0_n);
      load_ab(0);
    }
    compute();
    if (!last) continue;

    // ---- tile finished: epilogue, then switch to the next tile's context
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      int rows[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rows[r] = rid[wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int gc = n0 + wn * 64 + ni * 32 + li;
        epilogue_tile<EPI>(acc[mi][ni], rows, gc, g.N, bv[ni], g.relu, g.mask, g.ldmask, g.C, g.ldc);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
      }
    }
    if (!valid_n) break;
    __syncthreads();  // every wave has read rid[] of the finished tile
    v_cur += G;
#pragma unroll
    for (int q = 0; q < 4; ++q) grow[q] = grow_n[q];
    cls = cls_n;
    n0 = n0_n;
    if (tid < BM) rid[tid] = ridt_n;  // ordered before its first use by the two barriers of the next stage
    fetch_bias(n0);
    s1 = 0;
    k1 = 0;
    valid_n = row_tile_of(v_cur + G) < nrt;
    if (valid_n) {
      fetch_rows(v_cur + G, ti2_p0, ti2_pr, ti2_cls, grow_n, ridt_n, cls_n);
      n0_n = n0_of(v_cur + G);
    }
    if (grouped) fetch_ti(v_cur + 2 * G, ti2_p0, ti2_pr, ti2_cls);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Software-pipelined form of k_gemm3 (same contract, same arithmetic, same tile walk, two workgroups per CU).
//   * B never touches LDS: k_split_weights writes the images FRAGMENT-major, so a lane fetches its 16-byte MFMA operand
//     straight from L2 (a wave reads 1 KB contiguous per fragment); wave w multiplies all 128 rows by the 32 columns
//     32 w .., so the four waves read disjoint fragments (24 KB per K-tile and workgroup) and a slab's registers are
//     refilled with the next K-tile's fragments right after its MFMAs are issued;
//   * A: global loads three K-tiles ahead (two register stages), split + ds_write ONE K-tile ahead into the other half
//     of a two-stage LDS ring, inside the MFMA block of the current K-tile; ONE barrier per K-tile;
//   * every load of the loop is unconditional (clamped addresses, validity applied at LDS-store time): a load inside a
//     conditional block makes the compiler's s_waitcnt insertion fall back to vmcnt(0), i.e. exposes the memory latency;
//   * the second workgroup of the CU covers what one workgroup cannot hide: the LDS round trip behind the barrier,
//     prologue, epilogue and tile switch.
// Needs >= 4 K-tiles per output tile (the load cursor is at most one tile ahead of the multiply); used from 8.
// Bit-identical to k_gemm3 (same MFMAs, same operands, same order per accumulator: tests/test_gemm_split_gpu.py).
// Measured (tools/gemm3_diag.py, same box, us per launch, k_gemm3 -> k_gemm3p): post-layer 0 (K = 640) at 81 920 rows
// 104 -> 97, at 131 072 rows 167 -> 151, at 655 360 rows 744 -> 643 (0.40 of the bf16 peak); K = 384 product 71.5 -> 62.7;
// a lone workgroup's K-tile 1.36 us.  In-kernel wall-clock stamps (65 536 rows, one tile per workgroup, K = 640): prologue
// 3.4-5.1 us, K loop 2.0 us per K-tile in the CU's first workgroup and 2.4 us in its second (= 1.1-1.2 us per K-tile and CU
// against the 0.87 us MFMA floor), epilogue (64 four-byte stores per lane) 3.0-5.0 us, workgroup lifetime 48-57 us inside a
// 65 us kernel (dispatch of 2048 waves and the final write-back are the rest); on top of that cfg-2's 640 tiles meet 512
// workgroups; whole step (tools/ab_bench.py base / nopipe): cfg-2 7.75 vs 7.82 ms, cfg-4's batch 27.2 vs 27.5.
// ---------------------------------------------------------------------------------------------------------------
#define G3P_STAGES 2
#define G3P_MIN_KTILES 8  // shorter K: the two-barrier kernel is as fast (measured at 4 K-tiles) and has the shorter prologue; at 8
                         // K-tiles (cfg-3's 256-wide GINE layers, 327 680 rows) the pipelined kernel is worth 0.34 of the 19.4 ms step
#define G3P_LDS(MI) (G3P_STAGES * 3 * 32 * (MI) * G3_LDB + 2 * 32 * (MI) * 4)

// MI = 32-row blocks per tile (4: 128-row tiles; 3: 96-row tiles, taken when 128-row tiles leave the last round of the
// persistent workgroups mostly idle: cfg-2's 81 920 rows are 640 tiles for 512 workgroup slots).  Same arithmetic per row.
// Measured (round 3, same-box A/B of the cfg-2 step, 6.97-7.05 ms): 96-row tiles -0.01...-0.02 ms only -- the CU, not the
// workgroup slot, is the unit to balance (854 x 0.75 tiles are the same 2.5 tile-equivalents per CU as 640, and a workgroup
// alone on its CU runs nearly twice as fast); 160-row tiles (MI = 5: exactly two per CU, a fifth less B traffic per row, but
// 248 VGPRs and 76 KB of LDS) were +0.06 ms and are not instantiated.
template <int EPI, int MI>
__global__ void __launch_bounds__(256, 2) k_gemm3p(gemm_args g) {
  constexpr int TM = 32 * MI;             // rows per tile
  constexpr int PIECE = TM * G3_LDB;      // one bf16 image of a TM x 32 K-tile
  constexpr int OP = 3 * PIECE;           // the three images
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_p[];
  unsigned char* const Abuf = lds_p;                                  // [2][OP]
  int* const rid = reinterpret_cast<int*>(lds_p + G3P_STAGES * OP);   // [2][TM]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  // wave w multiplies ALL 128 rows by the 32 columns w*32..: the four waves read disjoint B fragments (24 KB per
  // K-tile and workgroup from L2; a 2 x 2 wave grid reads every fragment twice, and at ~70 GB/s per CU from L2 plus
  // ~30 GB/s per CU for A that, not the matrix pipe, bounds the K-tile)
  const int li = lane & 31, lh = lane >> 5;
  const int lrow = tid >> 3, lk4 = (tid & 7) * 4;
  const int NT = g.Npad / BN, KT = g.Kpad / BK;

  const int ny = g.ny, G = gridDim.x;
  const bool grouped = g.tile_info != nullptr;
  const int nrt = grouped ? g.ntiles[0] : (int)((g.M + TM - 1) / TM);
  auto row_tile_of = [&](int v) { return (v / (8 * ny)) * 8 + ((v % (8 * ny)) & 7); };
  auto n0_of = [&](int v) { return ((v % (8 * ny)) >> 3) * BN; };

  int v_l = blockIdx.x;  // tile of the A load cursor (the multiply is at most one tile behind it)
  if (row_tile_of(v_l) >= nrt) return;

  auto fetch_ti = [&](int v, int& p0, int& pr, int& c) {
    const int rt = row_tile_of(v);
    const int rtc = rt < nrt ? rt : 0;
    p0 = g.tile_info[3 * rtc];
    pr = g.tile_info[3 * rtc + 1];
    c = g.tile_info[3 * rtc + 2];
  };
  // row ids of a tile: RAW loaded values; the r < rows select is applied where the ids are consumed, a tile later (a
  // select right here would wait for these loads, and with them for every K-tile load in flight, at each tile switch)
  auto fetch_rows = [&](int v, int p0, int pr, int c, int (&gr)[MI], int& rt_id, int& cl) {
    const int t = tid < TM ? tid : 0;
    if (grouped) {
#pragma unroll
      for (int q = 0; q < MI; ++q) {
        const int r = lrow + 32 * q;
        gr[q] = g.row_index[p0 + (r < pr ? r : 0)];
      }
      rt_id = g.row_index[p0 + (t < pr ? t : 0)];
      cl = c;
    } else {
      const int64_t m0 = (int64_t)row_tile_of(v) * TM;
#pragma unroll
      for (int q = 0; q < MI; ++q) gr[q] = (int)(m0 + lrow + 32 * q);
      rt_id = (int)(m0 + t);
      cl = 0;
    }
  };
  auto rows_in_tile = [&](int v, int pr) {
    return grouped ? pr : (int)min((int64_t)TM, g.M - (int64_t)row_tile_of(v) * TM);
  };

  // A cursor's tile (cgrow), the tile after it (grow_n ..: row ids fetched one tile ahead, its tile_info two ahead), and
  // what the multiply / the B cursor take over when they reach the tile the A cursor has entered (ridt_p, n0_p, cls_p)
  int cgrow[MI], grow_n[MI] = {}, ridt_n = 0, pr_n = 0, cls_n = 0, n0_n = 0, ti2_p0 = 0, ti2_pr = 0, ti2_cls = 0;
  int n0_c, ridt_p = -1, n0_p = 0, cls_p = 0;
  bool valid_n, pend = false, cursor_valid = true;
  int tpar = 0, cls0 = 0;
  {
    int p0 = 0, pr = 0, c = 0, rt_id, cl;
    if (grouped) fetch_ti(v_l, p0, pr, c);
    fetch_rows(v_l, p0, pr, c, cgrow, rt_id, cl);
    {
      const int nr = rows_in_tile(v_l, pr);
#pragma unroll
      for (int q = 0; q < MI; ++q) cgrow[q] = (lrow + 32 * q < nr) ? cgrow[q] : -1;
      if (tid < TM) rid[tid] = tid < nr ? rt_id : -1;
    }
    n0_c = n0_of(v_l);
    valid_n = row_tile_of(v_l + G) < nrt;
    if (valid_n) {
      if (grouped) fetch_ti(v_l + G, p0, pr, c);
      fetch_rows(v_l + G, p0, pr, c, grow_n, ridt_n, cls_n);
      pr_n = rows_in_tile(v_l + G, pr);
      n0_n = n0_of(v_l + G);
    }
    if (grouped) fetch_ti(v_l + 2 * G, ti2_p0, ti2_pr, ti2_cls);
    cls0 = cl;
  }

  f32x16 acc[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  f32x4 ra[2][MI];          // A K-tiles in flight (fp32), two register stages
  int kvalid[2] = {0, 0};
  bf16x8 bfr[2][3];        // B fragments [slab][piece]: a slab is refilled (next K-tile) right after its MFMAs

  // ---- A cursor
  const float* rowp[MI];
  int rowv = 0, segK = 0, s1 = 0, k1 = 0;
  auto enter_segment = [&](int s_i) {
    const seg_dev& s = g.seg[s_i];
    rowv = 0;
#pragma unroll
    for (int q = 0; q < MI; ++q) {
      rowp[q] = s.a + (int64_t)(cgrow[q] >= 0 ? cgrow[q] : 0) * s.lda;
      rowv |= cgrow[q] >= 0 ? (1 << q) : 0;
    }
    segK = s.k;
  };
  auto load_a = [&](auto PC) {  // K-tile at the cursor -> register stage P (always issued)
    constexpr int P = decltype(PC)::value;
    const int k = k1 + lk4;
    const bool k_ok = k < segK;
    const int kc = k_ok ? k : 0;
#pragma unroll
    for (int q = 0; q < MI; ++q) ra[P][q] = *reinterpret_cast<const f32x4*>(rowp[q] + kc);
    kvalid[P] = (k_ok && cursor_valid) ? rowv : 0;
  };
  auto cursor_step = [&]() {
    if (!cursor_valid) return;
    k1 += BK;
    if (k1 < segK) return;
    ++s1;
    k1 = 0;
    if (s1 < g.nseg) {
      enter_segment(s1);
      return;
    }
    s1 = 0;
    if (valid_n) {
#pragma unroll
      for (int q = 0; q < MI; ++q) cgrow[q] = (lrow + 32 * q < pr_n) ? grow_n[q] : -1;
      enter_segment(0);
      ridt_p = tid < pr_n ? ridt_n : -1;
      n0_p = n0_n;
      cls_p = cls_n;
      pend = true;
      v_l += G;
      valid_n = row_tile_of(v_l + G) < nrt;
      if (valid_n) {
        fetch_rows(v_l + G, ti2_p0, ti2_pr, ti2_cls, grow_n, ridt_n, cls_n);
        pr_n = rows_in_tile(v_l + G, ti2_pr);
        n0_n = n0_of(v_l + G);
      }
      if (grouped) fetch_ti(v_l + 2 * G, ti2_p0, ti2_pr, ti2_cls);
    } else {
      cursor_valid = false;  // the loads go on over the last tile's first K-tile and are never multiplied
      enter_segment(0);
    }
  };

  // ---- B cursor: this lane's 16-byte slot of the fragment-major block [slab][piece][ni 0..3][lane][8] of a K-tile
  const __bf16* bptr;
  int kb = 0;
  bool b_valid = true;
  auto b_tile_base = [&](int cl, int nn0) {
    return g.bsplit + (((int64_t)cl * NT + nn0 / BN) * KT * 3) * (BN * BK) + (wave * 64 + lane) * 8;
  };
  auto load_b = [&](int sl) {
#pragma unroll
    for (int p = 0; p < 3; ++p) bfr[sl][p] = *reinterpret_cast<const bf16x8*>(bptr + ((sl * 3 + p) * 4 * 64) * 8);
  };
  auto b_step = [&]() {
    if (!b_valid) return;
    bptr += 3 * (BN * BK);
    if (++kb < KT) return;
    kb = 0;
    if (pend) {
      bptr = b_tile_base(cls_p, n0_p);
    } else {
      b_valid = false;
      bptr = b_tile_base(0, 0);
    }
  };

  typedef __attribute__((ext_vector_type(2))) float f32x2;
  auto store_a = [&](auto PC, unsigned char* A3) {  // register stage P -> LDS stage at A3
    constexpr int P = decltype(PC)::value;
#pragma unroll
    for (int h = 0; h < (MI + 1) / 2; ++h) {
      float xa[8];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          xa[4 * q + j] = (2 * h + q < MI && ((kvalid[P] >> (2 * h + q)) & 1)) ? ra[P][2 * h + q < MI ? 2 * h + q : 0][j] : 0.f;
      bf16x8 pc[3];
      split3(xa, pc[0], pc[1], pc[2]);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(&pc[p]);
        unsigned char* qa = A3 + p * PIECE + (lrow + 64 * h) * G3_LDB + lk4 * 2;
        *reinterpret_cast<f32x2*>(qa) = f32x2{w.x, w.y};
        if (2 * h + 1 < MI) *reinterpret_cast<f32x2*>(qa + 32 * G3_LDB) = f32x2{w.z, w.w};
      }
    }
  };
  const int afrag = li * G3_LDB + 16 * lh;  // this lane's fragment offset inside an image (mi = 0, slab 0)
  auto read_a1 = [&](const unsigned char* A3, int sl, int mi, bf16x8 (&a)[3]) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
      a[p] = *reinterpret_cast<const bf16x8*>(A3 + p * PIECE + afrag + mi * 32 * G3_LDB + 32 * sl);
  };
  auto read_a = [&](const unsigned char* A3, int sl, bf16x8 (&a)[MI][3]) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) read_a1(A3, sl, mi, a[mi]);
  };
  auto mfma_group = [&](f32x16& c, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
  };

  float bv;
  auto fetch_bias = [&](int nn0) {
    const int gc = nn0 + wave * 32 + li;
    bv = (g.bias != nullptr && gc < g.N) ? g.bias[gc] : 0.f;
  };
  fetch_bias(n0_c);

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---- prologue: A K-tile 0 in LDS stage 0; K-tiles 1, 2 in flight (register stages 1, 0); B K-tile 0 in flight
  enter_segment(0);
  bptr = b_tile_base(cls0, n0_c);
  load_a(I0{});
  cursor_step();
  load_a(I1{});
  cursor_step();
  load_b(0);
  load_b(1);
  b_step();
  store_a(I0{}, Abuf);
  load_a(I0{});
  cursor_step();
  __syncthreads();

  int kc = 0;
  bool done = false;
  auto iteration = [&](auto SC, auto PC) {  // S = j & 1: LDS stage of K-tile j; P = 1 - S: register stage of K-tile j + 1
    const unsigned char* const Ac = Abuf + decltype(SC)::value * OP;
    unsigned char* const As = Abuf + decltype(PC)::value * OP;
    // multiply K-tile j || split K-tile j + 1 into the other LDS stage || refill each slab's B fragments (K-tile j + 1)
    bf16x8 af[MI][3];
    read_a(Ac, 0, af);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      mfma_group(acc[mi], af[mi], bfr[0]);
      read_a1(Ac, 1, mi, af[mi]);
    }
    load_b(0);
    store_a(PC, As);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) mfma_group(acc[mi], af[mi], bfr[1]);
    load_b(1);
    b_step();
    load_a(PC);  // A K-tile j + 3 (register stage P was stored just above)
    if (++kc == KT) {  // tile finished
      const int* rt = rid + tpar * TM;
      const int gc = n0_c + wave * 32 + li;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        int rows[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rows[r] = rt[mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
        epilogue_tile<EPI>(acc[mi], rows, gc, g.N, bv, g.relu, g.mask, g.ldmask, g.C, g.ldc);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
      }
      if (!pend) {
        done = true;
        return;
      }
      // take over the tile the A cursor entered three K-tiles ago (other rid stage: its last readers were the epilogue
      // of the tile before this one)
      tpar ^= 1;
      if (tid < TM) rid[tpar * TM + tid] = ridt_p;
      n0_c = n0_p;
      fetch_bias(n0_c);
      pend = false;
      kc = 0;
    }
    cursor_step();
    __syncthreads();
  };
  while (true) {
    iteration(I0{}, I1{});
    if (done) break;
    iteration(I1{}, I0{});
    if (done) break;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Activation-stationary split product for SHORT K, WIDE N (one segment, 96 < K <= 128, N a multiple of 128 and >= 256:
// dA = g . W_eff(d)^T of post-layer 0, K = 128 -> N = 512, per degree class).  k_gemm3 gives every (row tile, column
// tile) pair its own workgroup, so the same 128 x 128 activation tile is loaded and split N / 128 times; with 4 K-tiles
// per tile the split costs as much as the MFMAs.  Here a workgroup owns a 64-row tile: loads + splits it ONCE into LDS
// (three bf16 images, 64 rows x 272 B, 51 KB: three workgroups per CU), then walks the column tiles; wave w multiplies
// all 64 rows by the columns 32 w .. of the column tile, its B fragments coming straight from the fragment-major weight
// image in L2 (k_split_weights, frag = 1) through a two-K-tile register ring that is refilled right after each slab's
// MFMAs and runs on across column-tile boundaries (every load unconditional: past the end the last K-tile is re-read).
// No predicates anywhere in the column loop, so every s_waitcnt of the ring is exact (vmcnt counts loads AND stores in
// order; behind a data-dependent number of stores hipcc waits for all of them):
//   * a partial tile repeats its last valid row (same inputs -> same outputs -> the same value stored to the same
//     address more than once);
//   * N % 128 == 0 and all element offsets below 2^30 are launch conditions (else k_gemm3 runs), so a store is a
//     uniform base + a 32-bit lane offset read from LDS (row * ldc, computed once per tile).
// Same MFMAs, same operands, same order per accumulator as k_gemm3: bit-identical results.
// A grouped call's 128-row class tiles (gnx_class_tiles) are taken as two 64-row halves.
// ---------------------------------------------------------------------------------------------------------------
#define AS_BM 64
#define AS_LDA 272                  // bytes per image row: 128 bf16 + 16 (row stride = 17 x 16 B: conflict-free ds_read_b128)
#define AS_PIECE (AS_BM * AS_LDA)
#define AS_LDS (3 * AS_PIECE + 2 * AS_BM * 4)

#ifdef AS_STAMP  // diagnostic build only (tools/ubench/gemm_as3_stamp.hip): per-workgroup phase cycle sums
__device__ unsigned long long* as_stamp_buf = nullptr;
#define AS_AT(i)                                     \
  do {                                               \
    __builtin_amdgcn_sched_barrier(0);               \
    const unsigned long long tn_ = clock64();        \
    tacc[i] += tn_ - tprev;                          \
    tprev = tn_;                                     \
    __builtin_amdgcn_sched_barrier(0);               \
  } while (0)
#else
#define AS_AT(i)
#endif

template <int EPI>
__global__ void __launch_bounds__(256, 3) k_gemm_as3(gemm_args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_as[];
#ifdef AS_STAMP
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long tstart = clock64();
  unsigned long long tprev = tstart;
#endif
  unsigned char* const A3 = lds_as;
  unsigned* const roff = reinterpret_cast<unsigned*>(lds_as + 3 * AS_PIECE);              // row * ldc
  unsigned* const moff = reinterpret_cast<unsigned*>(lds_as + 3 * AS_PIECE + AS_BM * 4);  // row * ldmask

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int NT = g.Npad / BN;
  const bool grouped = g.tile_info != nullptr;

  int p0, pr, cls = 0;
  if (grouped) {
    const int t = blockIdx.x >> 1, half = blockIdx.x & 1;
    if (t >= g.ntiles[0]) return;
    p0 = g.tile_info[3 * t] + AS_BM * half;
    pr = g.tile_info[3 * t + 1] - AS_BM * half;
    cls = g.tile_info[3 * t + 2];
    if (pr <= 0) return;
    pr = pr < AS_BM ? pr : AS_BM;
  } else {
    const int64_t m0 = (int64_t)blockIdx.x * AS_BM;
    p0 = (int)m0;
    pr = (int)min((int64_t)AS_BM, g.M - m0);
  }

  // ---- A tile: 32 lanes cover one 512-byte row, 8 rows per pass, 8 passes; B ring: K-tile 0 of column tile 0
  const seg_dev& s = g.seg[0];
  const int lr = tid >> 5, k4 = (tid & 31) * 4;
  const bool k_ok = k4 < s.k;
  int grow[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int r = lr + 8 * q;
    const int idx = p0 + (r < pr ? r : pr - 1);
    grow[q] = grouped ? g.row_index[idx] : idx;
  }
  int my_row = 0;
  if (tid < AS_BM) {
    const int idx = p0 + (tid < pr ? tid : pr - 1);
    my_row = grouped ? g.row_index[idx] : idx;
  }
  f32x4 ra[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) ra[q] = *reinterpret_cast<const f32x4*>(s.a + (int64_t)grow[q] * s.lda + (k_ok ? k4 : 0));

  const int total_kt = NT * 4;
  const __bf16* const bbase = g.bsplit + ((int64_t)cls * NT * 4 * 3) * (BN * BK) + (wave * 64 + lane) * 8;
  bf16x8 bq[2][2][3];  // [K-tile parity][slab][piece]
  auto load_b = [&](int slot, int sl, int j) {
    const __bf16* p = bbase + (int64_t)(j < total_kt ? j : total_kt - 1) * (3 * BN * BK);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) bq[slot][sl][pc] = *reinterpret_cast<const bf16x8*>(p + ((sl * 3 + pc) * 4 * 64) * 8);
  };
  load_b(0, 0, 0);
  load_b(0, 1, 0);
  AS_AT(0);  // tile info, row ids, issue of the A loads

  typedef __attribute__((ext_vector_type(2))) float f32x2;
  if (tid < AS_BM) {
    roff[tid] = (unsigned)my_row * (unsigned)g.ldc;
    moff[tid] = (unsigned)my_row * (unsigned)g.ldmask;
  }
#pragma unroll
  for (int q = 0; q < 8; q += 2) {
    float xa[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xa[j] = k_ok ? ra[q][j] : 0.f;
      xa[4 + j] = k_ok ? ra[q + 1][j] : 0.f;
    }
    bf16x8 pc[3];
    split3(xa, pc[0], pc[1], pc[2]);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(&pc[p]);
      unsigned char* qa = A3 + p * AS_PIECE + (lr + 8 * q) * AS_LDA + k4 * 2;
      *reinterpret_cast<f32x2*>(qa) = f32x2{w.x, w.y};
      *reinterpret_cast<f32x2*>(qa + 8 * AS_LDA) = f32x2{w.z, w.w};
    }
  }
  load_b(1, 0, 1);  // (after the split: held across it, the second K-tile's 24 registers spill)
  load_b(1, 1, 1);
  AS_AT(1);  // wait for A, split, LDS stores
  __syncthreads();
  AS_AT(2);

  // A fragments come from LDS for every slab (one slab ahead of its MFMAs).  The offset is made opaque per column tile:
  // the tile is loop-invariant, and hoisting its 48 fragments (192 VGPRs) out of the column loop spills.  (The OFFSET,
  // not the pointer: an opaque pointer loses its LDS address space and turns the reads into flat loads.)
  int afrag = li * AS_LDA + 16 * lh;
  bf16x8 af[2][2][3];  // [slab parity][mi][piece]
  auto read_a = [&](int slot, int off, int ks) {  // ks = 16-deep slab 0..7 of the K = 128 tile
    const unsigned char* base = A3 + off;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        af[slot][mi][p] = *reinterpret_cast<const bf16x8*>(base + p * AS_PIECE + mi * 32 * AS_LDA + ks * 32);
  };
  read_a(0, afrag, 0);

  // Epilogue geometry.  The MFMA result has lane = column, 4 consecutive ROWS per register group; a 4 x 4 transpose
  // inside every quad of lanes (two DPP exchange stages) turns that into lane = (row 8g + 4 lh + j, columns 4c .. 4c+3)
  // with j = li & 3, c = li >> 2: one 16-byte store per group, 8 lanes per 128-byte line, 8 full lines per wave
  // instruction (the 4-byte stores of the untransposed layout reached 2.2 TB/s: the epilogue took as long as the MFMAs).
  const int qj = li & 3;
  const unsigned lc = (unsigned)(wave * 32 + (li >> 2) * 4);
  unsigned ro[2][4], mo[2][4];  // row * ldc (row * ldmask) of this lane's 8 stores per column tile
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      ro[mi][gq] = roff[mi * 32 + 8 * gq + 4 * lh + qj] + lc;
      mo[mi][gq] = moff[mi * 32 + 8 * gq + 4 * lh + qj] + lc;
    }
  // bias of the NEXT column tile is fetched a tile ahead (unconditional, from a zero vector when there is none)
  const float* const bias_p = g.bias != nullptr ? g.bias : c_zero4;
  const int bias_on = g.bias != nullptr ? 1 : 0;
  auto bias_at = [&](int nt_) { return *reinterpret_cast<const f32x4*>(bias_p + (bias_on ? nt_ * BN + (int)lc : 0)); };
  f32x4 bv = bias_at(0);
  const float floor_v = g.relu ? 0.f : -__builtin_inff();
  asm volatile("" ::"v"(bv));  // waited for HERE: pending at the loop entry it would cost a vmcnt(0) in every iteration
  // exchange within lane pairs (xor 1) / across pairs (xor 2) of a quad: x and y swap their off-diagonal elements
  auto quad_swap = [&](float& x, float& y, auto CTRL, bool hi) {
    const float send = hi ? x : y;
    const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), decltype(CTRL)::value, 0xf, 0xf, true));
    x = hi ? got : x;
    y = hi ? y : got;
  };
  using XOR1 = std::integral_constant<int, 0xB1>;  // quad_perm [1,0,3,2]
  using XOR2 = std::integral_constant<int, 0x4E>;  // quad_perm [2,3,0,1]

  for (int nt = 0; nt < NT; ++nt) {
    asm volatile("" : "+v"(afrag));
    f32x16 acc[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
    const f32x4 bv_next = bias_at(nt + 1 < NT ? nt + 1 : nt);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int kt = ks >> 1, sl = ks & 1;
      read_a((ks + 1) & 1, afrag, (ks + 1) & 7);
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8(&b)[3] = bq[kt & 1][sl];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const bf16x8(&a)[3] = af[ks & 1][mi];
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[mi], 0, 0, 0);
      }
      load_b(kt & 1, sl, nt * 4 + kt + 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" ::"v"(bv_next));  // touched here, where the number of younger loads is static
    AS_AT(3);  // fragment reads + MFMAs of a column tile
    float* const cbase = g.C + nt * BN;
    const float* const mbase = (EPI == EPI_MASK ? g.mask : g.C) + nt * BN;
    // every load of the accumulate / mask operand precedes every store: a repeated row (partial tile) must read the
    // ORIGINAL C in all its copies (same wave, same address, program order), or it would be accumulated more than once
    f32x4 extra[2][4];
    if constexpr (EPI != EPI_PLAIN) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          extra[mi][gq] = *reinterpret_cast<const f32x4*>(mbase + (EPI == EPI_MASK ? mo[mi][gq] : ro[mi][gq]));
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        float c0 = acc[mi][4 * gq], c1 = acc[mi][4 * gq + 1], c2 = acc[mi][4 * gq + 2], c3 = acc[mi][4 * gq + 3];
        quad_swap(c0, c1, XOR1{}, (qj & 1) != 0);
        quad_swap(c2, c3, XOR1{}, (qj & 1) != 0);
        quad_swap(c0, c2, XOR2{}, (qj & 2) != 0);
        quad_swap(c1, c3, XOR2{}, (qj & 2) != 0);
        f32x4 v = {c0 + bv.x, c1 + bv.y, c2 + bv.z, c3 + bv.w};
        if constexpr (EPI == EPI_ACCUM) v += extra[mi][gq];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], floor_v);
        if constexpr (EPI == EPI_MASK)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (extra[mi][gq][e] > 0.f) ? v[e] : 0.f;
        *reinterpret_cast<f32x4*>(cbase + ro[mi][gq]) = v;
      }
    bv = bv_next;
    AS_AT(4);  // epilogue
  }
#ifdef AS_STAMP
  if (tid == 0 && as_stamp_buf != nullptr) {
    for (int i = 0; i < 5; ++i) as_stamp_buf[(size_t)blockIdx.x * 8 + i] = tacc[i];
    as_stamp_buf[(size_t)blockIdx.x * 8 + 5] = tstart;
    as_stamp_buf[(size_t)blockIdx.x * 8 + 6] = clock64();
  }
#endif
}

template <int EPI>
static hipError_t as3_launch_one(gnx_handle* h, const gemm_args& g, unsigned grid) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_as3<EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)AS_LDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((k_gemm_as3<EPI>), dim3(grid), dim3(256), AS_LDS, h->stream, g);
  return hipSuccess;
}

template <bool BT, int EPI, bool FAST, int NW>
static hipError_t ws3_launch_fast(gnx_handle* h, const ws_args& g, int grid) {
  // LDS: two stages of a TM-row tile, and at least the 64 KB the NN weight staging image needs
  constexpr size_t lds = (NW == 8) ? (size_t)(2 * W3_BUF) : (size_t)(64 * 1024);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_ws3<BT, EPI, FAST, NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(NW == 8 ? 160 * 1024 : 64 * 1024));
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((k_gemm_ws3<BT, EPI, FAST, NW>), dim3(grid), dim3(64 * NW), lds, h->stream, g);
  return hipGetLastError();
}

template <bool BT, int EPI>
static hipError_t ws3_launch_one(gnx_handle* h, const ws_args& g, int grid) {
  // the predicate-free kernel: all 128 columns, 16-byte stores (and mask loads), no read-modify-write of C
  const bool fast = h->opt[GNX_OPT_GEMM_WS_FAST] != 0 && EPI != EPI_ACCUM && g.N == 128 && aligned16(g.C) && g.ldc % 4 == 0 &&
                    (g.mask == nullptr || (aligned16(g.mask) && g.ldmask % 4 == 0));
  if (EPI != EPI_ACCUM && fast) {
    if (h->opt[GNX_OPT_GEMM_WS_FAST] == 2) {  // two 4-wave workgroups per CU on 32-row tiles
      ws_args g2 = g;
      g2.ntiles = (int)gnx_cdiv(g.M, (int64_t)32);
      const int cus = h->num_cus > 0 ? h->num_cus : 256;
      const int grid2 = g2.ntiles < 2 * cus ? g2.ntiles : 2 * cus;
      return ws3_launch_fast<BT, EPI, EPI != EPI_ACCUM, 4>(h, g2, grid2);
    }
    return ws3_launch_fast<BT, EPI, EPI != EPI_ACCUM, 8>(h, g, grid);
  }
  return ws3_launch_fast<BT, EPI, false, 8>(h, g, grid);
}

static int32_t gemm_ws_launch(gnx_handle* h, const gnx_gemm_seg& s, int64_t M, int32_t N, const float* bias,
                              const float* mask, int64_t ldmask, float* C, int64_t ldc, int32_t flags) {
  ws_args g;
  g.A = s.a;
  g.lda = s.lda;
  g.B = s.b;
  g.ldb = s.ldb;
  g.M = M;
  g.N = N;
  g.K = s.k;
  g.bias = bias;
  g.mask = mask;
  g.ldmask = ldmask;
  g.C = C;
  g.ldc = ldc;
  g.relu = (flags & GNX_GEMM_RELU) ? 1 : 0;
  g.ntiles = (int)gnx_cdiv(M, WS_BM);
  const bool bt = (flags & GNX_GEMM_B_TRANS) != 0;
  const int epi = mask ? EPI_MASK : ((flags & GNX_GEMM_ACCUMULATE) ? EPI_ACCUM : EPI_PLAIN);
  const size_t lds_bytes = sizeof(float) * (128 * WS_LD + 2 * WS_BM * WS_LD);
  const int cus = h->num_cus > 0 ? h->num_cus : 256;
  const int grid = g.ntiles < cus ? g.ntiles : cus;
  const bool split = h->opt[GNX_OPT_GEMM_SPLIT] != 0;  // 0 = exact-fp32 MFMA kernel (A/B switch)
  const double fl = 2.0 * (double)M * N * s.k;
  gnx_prof_scope prof(h, GNX_K_GEMM_WS, 4.0 * M * (s.k + N) + ((mask || (flags & GNX_GEMM_ACCUMULATE)) ? 4.0 * M * N : 0.0) +
                                            4.0 * N * s.k, fl, split ? 6.0 * fl : 0.0);
  hipError_t e;
  if (split) {
    if (bt)
      e = epi == EPI_MASK    ? ws3_launch_one<true, EPI_MASK>(h, g, grid)
          : epi == EPI_ACCUM ? ws3_launch_one<true, EPI_ACCUM>(h, g, grid)
                             : ws3_launch_one<true, EPI_PLAIN>(h, g, grid);
    else
      e = epi == EPI_MASK    ? ws3_launch_one<false, EPI_MASK>(h, g, grid)
          : epi == EPI_ACCUM ? ws3_launch_one<false, EPI_ACCUM>(h, g, grid)
                             : ws3_launch_one<false, EPI_PLAIN>(h, g, grid);
  } else if (bt)
    e = epi == EPI_MASK    ? ws_launch_one<true, EPI_MASK>(h, g, grid, lds_bytes)
        : epi == EPI_ACCUM ? ws_launch_one<true, EPI_ACCUM>(h, g, grid, lds_bytes)
                           : ws_launch_one<true, EPI_PLAIN>(h, g, grid, lds_bytes);
  else
    e = epi == EPI_MASK    ? ws_launch_one<false, EPI_MASK>(h, g, grid, lds_bytes)
        : epi == EPI_ACCUM ? ws_launch_one<false, EPI_ACCUM>(h, g, grid, lds_bytes)
                           : ws_launch_one<false, EPI_PLAIN>(h, g, grid, lds_bytes);
  if (e != hipSuccess) {
    gnx_set_error("gnx_gemm (weights-stationary): %s", hipGetErrorString(e));
    return GNX_E_HIP;
  }
  return GNX_OK;
}

// eligibility of the weights-stationary path (everything else goes to the tiled kernel)
static bool gemm_ws_eligible(const gnx_handle* h, int32_t nseg, const gnx_gemm_seg* segs, int64_t M, int32_t N,
                             const float* mask, int32_t flags) {
  if (nseg != 1 || M < 8192) return false;
  const gnx_gemm_seg& s = segs[0];
  if (s.rowscale != nullptr || s.k > 128 || s.k < 32 || N > 128 || N < 32) return false;
  if ((s.k % 4) != 0 || (N % 4) != 0) return false;
  if (!aligned16(s.a) || (s.lda % 4) != 0 || !aligned16(s.b) || (s.ldb % 4) != 0) return false;
  if (mask != nullptr && (flags & GNX_GEMM_ACCUMULATE)) return false;
  return h->opt[GNX_OPT_GEMM_WS] != 0;
}


// ---------------------------------------------------------------------------------------------------------------
// Small-M product (M <= 256: the 60-row bond-table chain  BondEmb -> edge_encoder -> pre-layer-0 slice and its
// gradients).  The tiled kernel spends ~22 us of pure load/barrier latency on such a problem in ONE workgroup; here a
// 16 x 16 output patch per 256-thread workgroup stages its A rows and B rows/columns in LDS once and every thread does
// one K-long dot product: ~M/16 * N/16 workgroups in parallel, a few microseconds.  One segment, no row scale / mask.
// ---------------------------------------------------------------------------------------------------------------
#define SM_T 16
#define SM_KC 128  // K chunk staged per pass

template <bool B_TRANS>
__global__ void __launch_bounds__(256) k_gemm_small(const float* __restrict__ A, int64_t lda,
                                                     const float* __restrict__ B, int64_t ldb, int M, int N, int K,
                                                     const float* __restrict__ bias, float* __restrict__ C, int64_t ldc,
                                                     int relu, int accumulate) {
  __shared__ float As[SM_T][SM_KC + 1];
  __shared__ float Bs[SM_T][SM_KC + 1];  // Bs[n][k]
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.x * SM_T, n0 = blockIdx.y * SM_T;
  float acc = 0.f;
  for (int k0 = 0; k0 < K; k0 += SM_KC) {
    const int kc = (K - k0) < SM_KC ? (K - k0) : SM_KC;
    // stage A[m0..+16][k0..+kc] : consecutive threads along k
    for (int i = threadIdx.x; i < SM_T * SM_KC; i += 256) {
      const int r = i / SM_KC, k = i % SM_KC;
      As[r][k] = (m0 + r < M && k < kc) ? A[(int64_t)(m0 + r) * lda + k0 + k] : 0.f;
    }
    if (B_TRANS) {  // B[n][k]
      for (int i = threadIdx.x; i < SM_T * SM_KC; i += 256) {
        const int r = i / SM_KC, k = i % SM_KC;
        Bs[r][k] = (n0 + r < N && k < kc) ? B[(int64_t)(n0 + r) * ldb + k0 + k] : 0.f;
      }
    } else {  // B[k][n] : consecutive threads along n
      for (int i = threadIdx.x; i < SM_T * SM_KC; i += 256) {
        const int k = i / SM_T, r = i % SM_T;
        Bs[r][k] = (n0 + r < N && k < kc) ? B[(int64_t)(k0 + k) * ldb + n0 + r] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < SM_KC; ++k) acc = fmaf(As[ty][k], Bs[tx][k], acc);  // same k-ordered fmaf chain as the MFMA
    __syncthreads();
  }
  const int gm = m0 + ty, gn = n0 + tx;
  if (gm < M && gn < N) {
    float v = acc + (bias != nullptr ? bias[gn] : 0.f);
    float* cp = C + (int64_t)gm * ldc + gn;
    if (accumulate) v += *cp;
    *cp = relu ? fmaxf(v, 0.f) : v;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Mid-size product (M < 4096 rows and few 128 x 128 tiles: a 32-graph batch's 640 atoms / 1 280 bonds): the whole gnx_gemm
// contract (segments, row scale, degree-class grouping, every epilogue) on 16 x 16 output patches, one output per thread,
// the k-ordered fp32 fmaf chain of k_gemm_small.  The tiled kernel gives such a problem 5-10 workgroups that walk K
// serially, two barriers per 32 k: 40 us per launch, 78 launches = 3.3 of the 4.4 ms of a cfg-1 (B = 32) step; here
// M / 16 x N / 16 workgroups stage 128 k per barrier pair with 16-byte loads and read LDS 16 bytes at a time.
// ---------------------------------------------------------------------------------------------------------------
#define MID_T 16
#define MID_KC 128
#define MID_LD (MID_KC + 4)

template <bool B_TRANS, int EPI>
__global__ void __launch_bounds__(256) k_gemm_mid(gemm_args g) {
  __shared__ __attribute__((aligned(16))) float As[MID_T][MID_LD];
  __shared__ __attribute__((aligned(16))) float Bs[MID_T][MID_LD];  // Bs[n][k]
  __shared__ int rid[MID_T];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int n0 = blockIdx.y * MID_T;
  int cls = 0;
  if (g.tile_info != nullptr) {  // a 128-row class tile = eight 16-row patches
    const int t = blockIdx.x >> 3, sub = blockIdx.x & 7;
    if (t >= g.ntiles[0]) return;
    const int p0 = g.tile_info[3 * t] + MID_T * sub, pr = g.tile_info[3 * t + 1] - MID_T * sub;
    cls = g.tile_info[3 * t + 2];
    if (pr <= 0) return;
    if (tid < MID_T) rid[tid] = tid < pr ? g.row_index[p0 + tid] : -1;
  } else {
    const int64_t m0 = (int64_t)blockIdx.x * MID_T;
    if (tid < MID_T) rid[tid] = (m0 + tid < g.M) ? (int)(m0 + tid) : -1;
  }
  __syncthreads();

  // chunk = 128 k of one segment; the NEXT chunk's granules are fetched into registers while this one is multiplied
  f32x4 ra[2], rb[2];
  auto fetch = [&](int si, int k0) {
    const seg_dev& s = g.seg[si];
    const float* __restrict__ sb = s.b + (int64_t)cls * s.cls_stride;
    const bool va = s.vec_a && (s.k % 4 == 0), vb = s.vec_b && (B_TRANS ? (s.k % 4 == 0) : (g.N % 4 == 0));
    const int kc = (s.k - k0) < MID_KC ? (s.k - k0) : MID_KC;
    // A[rows][k0 .. k0+kc): 512 four-float granules, two per thread, consecutive threads along k
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gi = tid + 256 * j, r = gi >> 5, k4 = (gi & 31) * 4;
      const int row = rid[r];
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row >= 0 && k4 < kc) {
        const float* ap = s.a + (int64_t)row * s.lda + k0 + k4;
        if (va) {
          v = *reinterpret_cast<const f32x4*>(ap);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (k4 + e < kc) ? ap[e] : 0.f;
        }
        if (s.rs != nullptr) {
          const float rs = s.rs[row];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rs * v[e];
        }
      }
      ra[j] = v;
    }
    if (B_TRANS) {  // B[n][k]: the same shape as A
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int gi = tid + 256 * j, r = gi >> 5, k4 = (gi & 31) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n0 + r < g.N && k4 < kc) {
          const float* bp = sb + (int64_t)(n0 + r) * s.ldb + k0 + k4;
          if (vb) {
            v = *reinterpret_cast<const f32x4*>(bp);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (k4 + e < kc) ? bp[e] : 0.f;
          }
        }
        rb[j] = v;
      }
    } else {  // B[k][n]: four granules per k row, transposed into Bs[n][k] at store time
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int gi = tid + 256 * j, k = gi >> 2, c4 = (gi & 3) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < kc && n0 + c4 < g.N) {
          const float* bp = sb + (int64_t)(k0 + k) * s.ldb + n0 + c4;
          if (vb) {
            v = *reinterpret_cast<const f32x4*>(bp);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (n0 + c4 + e < g.N) ? bp[e] : 0.f;
          }
        }
        rb[j] = v;
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gi = tid + 256 * j;
      *reinterpret_cast<f32x4*>(&As[gi >> 5][(gi & 31) * 4]) = ra[j];
      if (B_TRANS) {
        *reinterpret_cast<f32x4*>(&Bs[gi >> 5][(gi & 31) * 4]) = rb[j];
      } else {
        const int k = gi >> 2, c4 = (gi & 3) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) Bs[c4 + e][k] = rb[j][e];
      }
    }
  };

  float acc = 0.f;
  int si = 0, k0 = 0;
  fetch(0, 0);
  while (true) {
    stage();
    __syncthreads();
    const int kc = (g.seg[si].k - k0) < MID_KC ? (g.seg[si].k - k0) : MID_KC;
    int sn = si, kn = k0 + MID_KC;
    if (kn >= g.seg[si].k) {
      ++sn;
      kn = 0;
    }
    const bool more = sn < g.nseg;
    if (more) fetch(sn, kn);
    const int kend = (kc + 3) & ~3;
    for (int k = 0; k < kend; k += 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&As[ty][k]);
      const f32x4 b = *reinterpret_cast<const f32x4*>(&Bs[tx][k]);
      acc = fmaf(a.x, b.x, acc);
      acc = fmaf(a.y, b.y, acc);
      acc = fmaf(a.z, b.z, acc);
      acc = fmaf(a.w, b.w, acc);
    }
    if (!more) break;
    __syncthreads();
    si = sn;
    k0 = kn;
  }
  const int gm = rid[ty], gn = n0 + tx;
  if (gm >= 0 && gn < g.N) {
    float v = acc + (g.bias != nullptr ? g.bias[gn] : 0.f);
    float* cp = g.C + (int64_t)gm * g.ldc + gn;
    if constexpr (EPI == EPI_ACCUM) v += *cp;
    v = g.relu ? fmaxf(v, 0.f) : v;
    if constexpr (EPI == EPI_MASK) v = (g.mask[(int64_t)gm * g.ldmask + gn] > 0.f) ? v : 0.f;
    *cp = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Batched small products: up to GNX_SMALL_BATCH independent problems of k_gemm_small's kind in ONE launch (the 60-row
// bond-table chain, the merged lin o last-post weights and their gradients for ALL layers of a model; VERDICT r2 #4:
// ~360 launches per step, 54 of them 16 x 16-patch products of ~8 us each).  blockIdx.y = problem, blockIdx.x = output
// patch.  Either operand may be read transposed (the weight gradients of these tiny layers are A^T B products), the
// result may be accumulated plainly (problems of one launch must then write disjoint outputs) or with fp32 atomics.
// Same k-ordered fmaf chain and K chunking as k_gemm_small: bit-identical results for the untransposed forms.
// ---------------------------------------------------------------------------------------------------------------
struct small_batch_args {
  gnx_small_prob p[GNX_SMALL_BATCH];
  int n;
};

__global__ void __launch_bounds__(256) k_gemm_small_batched(small_batch_args b) {
  __shared__ float As[SM_T][SM_KC + 1];
  __shared__ float Bs[SM_T][SM_KC + 1];
  const gnx_small_prob& q = b.p[blockIdx.y];
  const int M = q.M, N = q.N, K = q.K;
  const int nt = (N + SM_T - 1) / SM_T, mt = (M + SM_T - 1) / SM_T;
  if ((int)blockIdx.x >= nt * mt) return;  // (uniform per workgroup: before any barrier)
  const int m0 = (blockIdx.x / nt) * SM_T, n0 = (blockIdx.x % nt) * SM_T;
  const bool at = (q.flags & GNX_SB_A_TRANS) != 0, bt = (q.flags & GNX_SB_B_TRANS) != 0;
  const float* __restrict__ A = q.A;
  const float* __restrict__ B = q.B;
  const int64_t lda = q.lda, ldb = q.ldb;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  float acc = 0.f;
  for (int k0 = 0; k0 < K; k0 += SM_KC) {
    const int kc = (K - k0) < SM_KC ? (K - k0) : SM_KC;
    if (!at) {  // A[m][k]: consecutive threads along k
      for (int i = threadIdx.x; i < SM_T * SM_KC; i += 256) {
        const int r = i / SM_KC, k = i % SM_KC;
        As[r][k] = (m0 + r < M && k < kc) ? A[(int64_t)(m0 + r) * lda + k0 + k] : 0.f;
      }
    } else {  // A[k][m]: consecutive threads along m
      for (int i = threadIdx.x; i < SM_T * SM_KC; i += 256) {
        const int k = i / SM_T, r = i % SM_T;
        As[r][k] = (m0 + r < M && k < kc) ? A[(int64_t)(k0 + k) * lda + m0 + r] : 0.f;
      }
    }
    if (bt) {  // B[n][k]
      for (int i = threadIdx.x; i < SM_T * SM_KC; i += 256) {
        const int r = i / SM_KC, k = i % SM_KC;
        Bs[r][k] = (n0 + r < N && k < kc) ? B[(int64_t)(n0 + r) * ldb + k0 + k] : 0.f;
      }
    } else {  // B[k][n]
      for (int i = threadIdx.x; i < SM_T * SM_KC; i += 256) {
        const int k = i / SM_T, r = i % SM_T;
        Bs[r][k] = (n0 + r < N && k < kc) ? B[(int64_t)(k0 + k) * ldb + n0 + r] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < SM_KC; ++k) acc = fmaf(As[ty][k], Bs[tx][k], acc);
    __syncthreads();
  }
  const int gm = m0 + ty, gn = n0 + tx;
  if (gm < M && gn < N) {
    float v = acc + (q.bias != nullptr ? q.bias[gn] : 0.f);
    float* cp = q.C + (int64_t)gm * q.ldc + gn;
    if (q.flags & GNX_SB_ATOMIC) {
      atomicAdd(cp, v);
    } else {
      if (q.flags & GNX_SB_ACCUMULATE) v += *cp;
      *cp = (q.flags & GNX_SB_RELU) ? fmaxf(v, 0.f) : v;
    }
  }
}

extern "C" int32_t gnx_gemm_small_batched(gnx_handle* h, int32_t nprob, const gnx_small_prob* probs) {
  GNX_CHECK_ARG(h && nprob >= 0 && (probs || nprob == 0), "gnx_gemm_small_batched: bad argument");
  for (int32_t i0 = 0; i0 < nprob; i0 += GNX_SMALL_BATCH) {
    small_batch_args b;
    b.n = nprob - i0 < GNX_SMALL_BATCH ? nprob - i0 : GNX_SMALL_BATCH;
    int maxt = 0;
    double bytes = 0.0, flops = 0.0;
    for (int i = 0; i < b.n; ++i) {
      const gnx_small_prob& q = probs[i0 + i];
      GNX_CHECK_ARG(q.A && q.B && q.C && q.M > 0 && q.N > 0 && q.K > 0, "gnx_gemm_small_batched: problem %d is empty or NULL",
                    i0 + i);
      GNX_CHECK_ARG(!((q.flags & GNX_SB_ATOMIC) && (q.flags & GNX_SB_RELU)), "gnx_gemm_small_batched: atomic + relu");
      b.p[i] = q;
      const int t = (int)(gnx_cdiv(q.M, SM_T) * gnx_cdiv(q.N, SM_T));
      maxt = t > maxt ? t : maxt;
      bytes += 4.0 * ((double)q.M * (q.K + q.N) + (double)q.N * q.K);
      flops += 2.0 * q.M * q.N * q.K;
    }
    for (int i = b.n; i < GNX_SMALL_BATCH; ++i) b.p[i] = b.p[0];
    gnx_prof_scope prof(h, GNX_K_GEMM_SMALL, bytes, flops, 0.0);
    hipLaunchKernelGGL(k_gemm_small_batched, dim3((unsigned)maxt, (unsigned)b.n), dim3(256), 0, h->stream, b);
    GNX_LAUNCH_CHECK();
  }
  return GNX_OK;
}

// Which products take the split-operand tiled kernel (k_split_weights -> k_gemm3) and how many bytes of split weight
// images they need.  Shared by gnx_gemm_workspace_bytes and the launch path so both always agree.
static size_t gemm_split_bytes(const gnx_handle* h, int32_t nseg, const gnx_gemm_seg* segs, const int64_t* cls_strides,
                               int32_t num_classes, int64_t M, int32_t N, const float* mask, int32_t flags,
                               bool grouped) {
  if (h == nullptr || segs == nullptr || nseg < 1 || nseg > MAX_SEGS || M < 4096 || N <= 0) return 0;
  if (h->opt[GNX_OPT_GEMM_SPLIT] == 0 || h->opt[GNX_OPT_GEMM_VEC] == 0) return 0;
  if (!grouped && gemm_ws_eligible(h, nseg, segs, M, N, mask, flags)) return 0;  // weights live in registers there
  const bool bt = (flags & GNX_GEMM_B_TRANS) != 0;
  int64_t kpad = 0;
  int ksteps = 0;
  for (int q = 0; q < nseg; ++q) {
    const gnx_gemm_seg& in = segs[q];
    const int64_t cs = cls_strides ? cls_strides[q] : 0;
    if (in.rowscale != nullptr || in.k <= 0) return 0;
    if (!aligned16(in.a) || (in.lda % 4) != 0 || !aligned16(in.b) || (in.ldb % 4) != 0 || (cs % 4) != 0) return 0;
    if ((in.k % 4) != 0 || (!bt && (N % 4) != 0)) return 0;
    ksteps += (int)gnx_cdiv((int64_t)in.k, BK);
    kpad += gnx_cdiv((int64_t)in.k, BK) * BK;
  }
  if (ksteps < 2) return 0;
  const int64_t npad = gnx_cdiv((int64_t)N, BN) * BN;
  const int64_t D = num_classes > 0 ? num_classes : 1;
  return (size_t)(D * 3 * npad * kpad) * sizeof(__bf16);
}

// Row-tile height of the pipelined tiled product for M rows x N columns: 96 where that shortens the longest workgroup's
// walk (rounds x tile height; persistent workgroups, two per CU), else 128.  GNX_OPT_GEMM_TILE_ROWS forces 96 / 128.
static int gemm_pipe_tile_rows(const gnx_handle* h, int64_t M, int32_t N) {
  const int forced = h->opt[GNX_OPT_GEMM_TILE_ROWS];
  if (forced == 96 || forced == 128) return forced;
  const int64_t ny = gnx_cdiv((int64_t)N, BN);
  const int64_t slots = 2 * (int64_t)(h->num_cus > 0 ? h->num_cus : 256);
  int best = 128;
  int64_t best_cost = gnx_cdiv(gnx_cdiv(M, (int64_t)128) * ny, slots) * 128;
  for (int rows : {96}) {  // strictly shorter longest walk only (ties keep 128, the most measured shape)
    const int64_t c = gnx_cdiv(gnx_cdiv(M, (int64_t)rows) * ny, slots) * rows;
    if (c < best_cost) {
      best_cost = c;
      best = rows;
    }
  }
  return best;
}

extern "C" int32_t gnx_gemm_tile_rows(gnx_handle* h, int64_t M, int32_t N) {
  if (h == nullptr || M <= 0 || N <= 0) return 128;
  return gemm_pipe_tile_rows(h, M, N);
}

static int32_t gemm_launch(gnx_handle* h, int32_t nseg, const gnx_gemm_seg* segs, const int64_t* cls_strides,
                           int32_t num_classes, int64_t M, int32_t N, const float* bias, const float* mask, int64_t ldmask, float* C,
                           int64_t ldc, int32_t flags, const int32_t* row_index, const int32_t* tile_info,
                           const int32_t* ntiles, int64_t max_tiles, void* ws, size_t ws_bytes, int32_t tile_rows = 0) {
  GNX_CHECK_ARG(h && segs && C, "gnx_gemm: NULL argument");
  GNX_CHECK_ARG(tile_rows == 0 || tile_rows == 96 || tile_rows == 128, "gnx_gemm: tile_rows=%d not in {0, 96, 128}", tile_rows);
  GNX_CHECK_ARG(nseg >= 1 && nseg <= MAX_SEGS, "gnx_gemm: nseg=%d not in [1,%d]", nseg, MAX_SEGS);
  GNX_CHECK_ARG(M >= 0 && N > 0 && ldc >= N, "gnx_gemm: bad shape M=%lld N=%d ldc=%lld", (long long)M, N, (long long)ldc);
  GNX_CHECK_ARG(!((flags & GNX_GEMM_RELU) && (flags & GNX_GEMM_ACCUMULATE)), "gnx_gemm: relu+accumulate rejected");
  GNX_CHECK_ARG(!(mask && (flags & GNX_GEMM_ACCUMULATE)), "gnx_gemm: mask+accumulate rejected");
  GNX_CHECK_ARG(mask == nullptr || ldmask >= N, "gnx_gemm: ldmask < N");
  if (M == 0) return GNX_OK;
  const bool bt = (flags & GNX_GEMM_B_TRANS) != 0;
  for (int s = 0; s < nseg; ++s) {
    const gnx_gemm_seg& in = segs[s];
    GNX_CHECK_ARG(in.a && in.b && in.k > 0, "gnx_gemm: segment %d: NULL operand or k<=0", s);
    GNX_CHECK_ARG(in.lda >= in.k, "gnx_gemm: segment %d: lda < k", s);
    GNX_CHECK_ARG(bt ? in.ldb >= in.k : in.ldb >= N, "gnx_gemm: segment %d: ldb too small", s);
  }
  const bool split_only = (flags & GNX_GEMM_SPLIT_ONLY) != 0;  // only the weight images (if this call uses any) are written
  if (tile_info == nullptr && gemm_ws_eligible(h, nseg, segs, M, N, mask, flags))
    return split_only ? GNX_OK : gemm_ws_launch(h, segs[0], M, N, bias, mask, ldmask, C, ldc, flags);
  if (split_only && M < 4096) return GNX_OK;  // (no split path below 4096 rows)
  if (tile_info == nullptr && nseg == 1 && M <= 256 && mask == nullptr && segs[0].rowscale == nullptr &&
      h->opt[GNX_OPT_GEMM_MID] == 0) {  // (k_gemm_mid below computes the same k-ordered chain with 16-byte staging loads and
                                        // the next chunk prefetched: 3-4 us instead of 7-10 us for a 20-row product)
    const gnx_gemm_seg& s0 = segs[0];
    dim3 grid((unsigned)gnx_cdiv(M, SM_T), (unsigned)gnx_cdiv(N, SM_T));
    gnx_prof_scope prof(h, GNX_K_GEMM_SMALL, 4.0 * (M * (double)(s0.k + N) + (double)N * s0.k), 2.0 * M * N * s0.k, 0.0);
    if (bt)
      hipLaunchKernelGGL(k_gemm_small<true>, grid, dim3(256), 0, h->stream, s0.a, s0.lda, s0.b, s0.ldb, (int)M, (int)N,
                         (int)s0.k, bias, C, ldc, (flags & GNX_GEMM_RELU) ? 1 : 0, (flags & GNX_GEMM_ACCUMULATE) ? 1 : 0);
    else
      hipLaunchKernelGGL(k_gemm_small<false>, grid, dim3(256), 0, h->stream, s0.a, s0.lda, s0.b, s0.ldb, (int)M, (int)N,
                         (int)s0.k, bias, C, ldc, (flags & GNX_GEMM_RELU) ? 1 : 0, (flags & GNX_GEMM_ACCUMULATE) ? 1 : 0);
    GNX_LAUNCH_CHECK();
    return GNX_OK;
  }

  gemm_args g;
  for (int s = 0; s < MAX_SEGS; ++s) g.seg[s] = seg_dev{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0};
  for (int s = 0; s < nseg; ++s) {
    const gnx_gemm_seg& in = segs[s];
    seg_dev d;
    d.a = in.a;
    d.rs = in.rowscale;
    d.b = in.b;
    d.lda = in.lda;
    d.ldb = in.ldb;
    d.cls_stride = cls_strides ? cls_strides[s] : 0;
    d.k = in.k;
    d.vec_a = aligned16(in.a) && (in.lda % 4 == 0);
    d.vec_b = aligned16(in.b) && (in.ldb % 4 == 0) && (d.cls_stride % 4 == 0);
    g.seg[s] = d;
  }
  g.nseg = nseg;
  g.M = M;
  g.N = N;
  g.bias = bias;
  g.mask = mask;
  g.ldmask = ldmask;
  g.C = C;
  g.ldc = ldc;
  g.relu = (flags & GNX_GEMM_RELU) ? 1 : 0;
  g.row_index = row_index;
  g.tile_info = tile_info;
  g.ntiles = ntiles;
  const int epi = mask ? EPI_MASK : ((flags & GNX_GEMM_ACCUMULATE) ? EPI_ACCUM : EPI_PLAIN);
  dim3 grid((unsigned)(tile_info ? max_tiles : gnx_cdiv(M, BM)), (unsigned)gnx_cdiv(N, BN));
  g.ny = (int)grid.y;
  if (h->opt[GNX_OPT_GEMM_MID] != 0 && M < 4096 && gnx_cdiv(M, BM) * (int64_t)grid.y <= 48) {  // (a grouped call's
    // max_tiles is an upper bound with one partial tile per class: the row count decides)
    // few 128 x 128 tiles: 16 x 16 patches on many workgroups instead (see k_gemm_mid)
    double ktot = 0.0;
    for (int q = 0; q < nseg; ++q) ktot += g.seg[q].k;
    gnx_prof_scope prof(h, GNX_K_GEMM_SMALL, 4.0 * M * (ktot + N) + 4.0 * N * ktot, 2.0 * (double)M * N * ktot, 0.0);
    const dim3 gm((unsigned)(tile_info ? max_tiles * 8 : gnx_cdiv(M, MID_T)), (unsigned)gnx_cdiv(N, MID_T));
#define GNX_LAUNCH_MID(BT)                                                                              \
  do {                                                                                                  \
    if (epi == EPI_MASK)                                                                                \
      hipLaunchKernelGGL((k_gemm_mid<BT, EPI_MASK>), gm, dim3(256), 0, h->stream, g);                   \
    else if (epi == EPI_ACCUM)                                                                          \
      hipLaunchKernelGGL((k_gemm_mid<BT, EPI_ACCUM>), gm, dim3(256), 0, h->stream, g);                  \
    else                                                                                                \
      hipLaunchKernelGGL((k_gemm_mid<BT, EPI_PLAIN>), gm, dim3(256), 0, h->stream, g);                  \
  } while (0)
    if (bt)
      GNX_LAUNCH_MID(true);
    else
      GNX_LAUNCH_MID(false);
#undef GNX_LAUNCH_MID
    GNX_LAUNCH_CHECK();
    return GNX_OK;
  }
  // k_gemm3 / k_gemm3p: persistent workgroups, two per CU, count a multiple of 8 * ny (the XCD-aware tile walk needs it)
  auto persistent_grid = [&](unsigned row_tiles) {
    unsigned g3 = (unsigned)(gnx_cdiv((int64_t)row_tiles, 8) * 8 * grid.y);
    const unsigned unit = 8u * grid.y;
    unsigned slots = 2u * (unsigned)(h->num_cus > 0 ? h->num_cus : 256);
    slots = slots / unit * unit;
    if (slots < unit) slots = unit;
    return g3 > slots ? slots : g3;
  };
  dim3 grid3(persistent_grid(grid.x));
  bool pipe = h->opt[GNX_OPT_GEMM_PIPE] != 0;  // decided below: needs >= G3P_MIN_KTILES K-tiles per output tile
  bool as3 = false;                            // decided below: one segment, 4 K-tiles, >= 2 column tiles
  bool mi3 = false;                            // pipelined kernel with 96-row tiles
  bool vec = h->opt[GNX_OPT_GEMM_VEC] != 0;
  for (int s = 0; s < nseg; ++s)
    vec = vec && g.seg[s].vec_a && g.seg[s].vec_b && (g.seg[s].k % 4 == 0) && (bt || (N % 4 == 0));
  // Split-operand path (M >= 4096, >= 2 K-tiles, no row scale): needs the caller's workspace for the split weight
  // images (gnx_gemm_workspace_bytes).  Row-scaled segments (the 4-segment post-layer-0 of hub-heavy batches) and
  // small, launch-bound problems stay on the fp32-MFMA kernel, and so does a call without a workspace.
  const size_t need = gemm_split_bytes(h, nseg, segs, cls_strides, num_classes, M, N, mask, flags, tile_info != nullptr);
  const bool split = need > 0 && ws != nullptr;
  if (split && ws_bytes < need) {
    gnx_set_error("gnx_gemm: workspace of %zu bytes, the split weight images need %zu", ws_bytes, need);
    return GNX_E_WORKSPACE;
  }
  GNX_CHECK_ARG(!split || aligned16(ws), "gnx_gemm: workspace must be 16-byte aligned");
  double ktot = 0.0;
  for (int q = 0; q < nseg; ++q) ktot += g.seg[q].k;
  const double fl = 2.0 * (double)M * N * ktot;
  gnx_prof_scope prof(h, GNX_K_GEMM_TILED, 4.0 * M * (ktot + N) + ((mask || (flags & GNX_GEMM_ACCUMULATE)) ? 4.0 * M * N : 0.0) +
                                               4.0 * N * ktot * (num_classes > 0 ? num_classes : 1), fl, split ? 6.0 * fl : 0.0);
  if (split) {
    split_args sa;
    for (int q = 0; q < MAX_SEGS; ++q) sa.seg[q] = g.seg[q];
    int kp = 0;
    for (int q = 0; q < nseg; ++q) {
      sa.koff[q] = kp;
      kp += (int)gnx_cdiv((int64_t)g.seg[q].k, BK) * BK;
    }
    for (int q = nseg; q <= MAX_SEGS; ++q) sa.koff[q] = kp;
    sa.nseg = nseg;
    sa.N = N;
    sa.Npad = (int)gnx_cdiv((int64_t)N, BN) * BN;
    sa.Kpad = kp;
    sa.D = num_classes > 0 ? num_classes : 1;
    pipe = pipe && kp / BK >= G3P_MIN_KTILES;
    if (pipe) {  // tile height of the pipelined kernel: the caller's (grouped: what its tile table was built with) or ours
      const int rows = tile_info ? (tile_rows ? tile_rows : BM) : (tile_rows ? tile_rows : gemm_pipe_tile_rows(h, M, N));
      mi3 = rows == 96;
      if (!tile_info) grid3 = dim3(persistent_grid((unsigned)gnx_cdiv(M, (int64_t)rows)));
    }
    as3 = h->opt[GNX_OPT_GEMM_AS] != 0 && nseg == 1 && kp == 4 * BK && N >= 2 * BN && N % BN == 0 &&
          (double)M * (double)(ldc > ldmask ? ldc : ldmask) + (double)N < 1073741824.0 && aligned16(C) && ldc % 4 == 0 &&
          (mask == nullptr || (aligned16(mask) && ldmask % 4 == 0)) && (bias == nullptr || aligned16(bias));
    sa.frag = (pipe || as3) ? 1 : 0;
    sa.out = reinterpret_cast<__bf16*>(ws);
    const int64_t items = (int64_t)sa.D * sa.Npad * (sa.Kpad / 8);
    if (flags & GNX_GEMM_PRESPLIT) {
      // the caller ran this very split before (GNX_GEMM_SPLIT_ONLY, same arguments)
    } else if (bt)
      hipLaunchKernelGGL(k_split_weights<true>, dim3((unsigned)gnx_cdiv(items, 256)), dim3(256), 0, h->stream, sa);
    else
      hipLaunchKernelGGL(k_split_weights<false>, dim3((unsigned)gnx_cdiv(items, 256)), dim3(256), 0, h->stream, sa);
    if (split_only) {
      GNX_LAUNCH_CHECK();
      return GNX_OK;
    }
    g.bsplit = sa.out;
    g.Npad = sa.Npad;
    g.Kpad = sa.Kpad;
    for (int q = 0; q < MAX_SEGS; ++q) g.koff[q] = sa.koff[q];
    g.steps = sa.Kpad / BK;
  }
  if (split_only) return GNX_OK;  // (this call takes a kernel that needs no images)
#define GNX_LAUNCH_GEMM(BT, EPI)                                                          \
  do {                                                                                    \
    if (split) {                                                                          \
      if (as3) {                                                                          \
        const unsigned ga = tile_info ? (unsigned)(2 * max_tiles) : (unsigned)gnx_cdiv(M, AS_BM); \
        GNX_HIP(as3_launch_one<EPI>(h, g, ga));                                           \
      } else if (pipe && mi3)                                                             \
        hipLaunchKernelGGL((k_gemm3p<EPI, 3>), grid3, dim3(256), G3P_LDS(3), h->stream, g); \
      else if (pipe)                                                                      \
        hipLaunchKernelGGL((k_gemm3p<EPI, 4>), grid3, dim3(256), G3P_LDS(4), h->stream, g); \
      else                                                                                \
        hipLaunchKernelGGL((k_gemm3<EPI>), grid3, dim3(256), 0, h->stream, g);            \
    } else if (vec)                                                                       \
      hipLaunchKernelGGL((k_gemm<BT, EPI, true>), grid, dim3(256), 0, h->stream, g);      \
    else                                                                                  \
      hipLaunchKernelGGL((k_gemm<BT, EPI, false>), grid, dim3(256), 0, h->stream, g);     \
  } while (0)
  if (bt) {
    if (epi == EPI_MASK)
      GNX_LAUNCH_GEMM(true, EPI_MASK);
    else if (epi == EPI_ACCUM)
      GNX_LAUNCH_GEMM(true, EPI_ACCUM);
    else
      GNX_LAUNCH_GEMM(true, EPI_PLAIN);
  } else {
    if (epi == EPI_MASK)
      GNX_LAUNCH_GEMM(false, EPI_MASK);
    else if (epi == EPI_ACCUM)
      GNX_LAUNCH_GEMM(false, EPI_ACCUM);
    else
      GNX_LAUNCH_GEMM(false, EPI_PLAIN);
  }
#undef GNX_LAUNCH_GEMM
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

extern "C" int32_t gnx_gemm(gnx_handle* h, int32_t nseg, const gnx_gemm_seg* segs, int64_t M, int32_t N,
                            const float* bias, const float* mask, int64_t ldmask, float* C, int64_t ldc, int32_t flags,
                            void* ws, size_t ws_bytes) {
  return gemm_launch(h, nseg, segs, nullptr, 1, M, N, bias, mask, ldmask, C, ldc, flags, nullptr, nullptr, nullptr, 0,
                     ws, ws_bytes);
}

extern "C" size_t gnx_gemm_workspace_bytes(gnx_handle* h, int32_t nseg, const gnx_gemm_seg* segs,
                                           const int64_t* cls_strides, int32_t num_classes, int64_t M, int32_t N,
                                           const float* mask, int32_t flags, int32_t grouped) {
  return gemm_split_bytes(h, nseg, segs, cls_strides, num_classes, M, N, mask, flags, grouped != 0);
}

extern "C" int32_t gnx_gemm_grouped(gnx_handle* h, int32_t nseg, const gnx_gemm_seg* segs, const int64_t* cls_strides,
                                    int32_t num_classes, int64_t M, int32_t N, const float* bias, const float* mask,
                                    int64_t ldmask, float* C, int64_t ldc, int32_t flags, const int32_t* row_index,
                                    const int32_t* tile_info, const int32_t* ntiles, int64_t max_tiles, void* ws,
                                    size_t ws_bytes) {
  GNX_CHECK_ARG(cls_strides && row_index && tile_info && ntiles && max_tiles > 0, "gnx_gemm_grouped: NULL argument");
  GNX_CHECK_ARG(num_classes >= 1 && num_classes <= 4096, "gnx_gemm_grouped: num_classes=%d not in [1,4096]", num_classes);
  return gemm_launch(h, nseg, segs, cls_strides, num_classes, M, N, bias, mask, ldmask, C, ldc, flags, row_index,
                     tile_info, ntiles, max_tiles, ws, ws_bytes);
}

extern "C" int32_t gnx_gemm_grouped_rows(gnx_handle* h, int32_t nseg, const gnx_gemm_seg* segs, const int64_t* cls_strides,
                                         int32_t num_classes, int64_t M, int32_t N, const float* bias, const float* mask,
                                         int64_t ldmask, float* C, int64_t ldc, int32_t flags, const int32_t* row_index,
                                         const int32_t* tile_info, const int32_t* ntiles, int64_t max_tiles, void* ws,
                                         size_t ws_bytes, int32_t tile_rows) {
  GNX_CHECK_ARG(cls_strides && row_index && tile_info && ntiles && max_tiles > 0, "gnx_gemm_grouped_rows: NULL argument");
  GNX_CHECK_ARG(num_classes >= 1 && num_classes <= 4096, "gnx_gemm_grouped_rows: num_classes=%d not in [1,4096]", num_classes);
  GNX_CHECK_ARG(tile_rows == 96 || tile_rows == 128, "gnx_gemm_grouped_rows: tile_rows=%d not in {96, 128}", tile_rows);
  return gemm_launch(h, nseg, segs, cls_strides, num_classes, M, N, bias, mask, ldmask, C, ldc, flags, row_index,
                     tile_info, ntiles, max_tiles, ws, ws_bytes, tile_rows);
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient  dW[n,k] += sum_m dC[m,n] * rs[m]*A[m,k]   (both operands k-row images: the contraction index m is the
// row of both).  grid = (M chunks, n tiles, k tiles); fp32 atomics into dW; the k-tile-0 workgroups also reduce dbias.
// ---------------------------------------------------------------------------------------------------------------
struct wgrad_args {
  const float* X;  // dC [M, N]
  int64_t ldx;
  const float* Y;  // A  [M, K]
  int64_t ldy;
  const float* rs;
  int64_t M;
  int N, K;
  float* dW;
  int64_t lddw;
  float* dbias;
  int64_t rows_per_block;
  int vec_x, vec_y;
  // grouped mode: blockIdx.x = chunk; rows row_index[chunk_info[3b] .. +chunk_info[3b+1]) add into dW + class * stride
  const int* row_index;
  const int* chunk_info;
  const int* nchunks;
  int64_t dw_cls_stride;
  const float* zero;  // 16 readable zero bytes (k_gemm_wgrad3p: target of masked-out loads)
};

template <bool VEC, bool GROUPED>
__device__ __forceinline__ void wgrad_body(const wgrad_args& g, const int bx, const int by, const int bz, float* Xs,
                                           float* Ys) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = by * BN;  // dW row tile (output features)
  const int c0 = bz * BN;  // dW col tile (input features)
  if (n0 >= g.N || c0 >= g.K) return;
  int64_t r_begin = (int64_t)bx * g.rows_per_block;
  int64_t r_end = r_begin + g.rows_per_block;
  if (r_end > g.M) r_end = g.M;
  float* dW = g.dW;
  if constexpr (GROUPED) {
    if (bx >= g.nchunks[0]) return;
    r_begin = g.chunk_info[3 * bx];
    r_end = r_begin + g.chunk_info[3 * bx + 1];
    dW += (int64_t)g.chunk_info[3 * bx + 2] * g.dw_cls_stride;
  }
  if (r_begin >= r_end) return;  // (workgroup-uniform) nothing to contribute: skip the zero-valued atomic flush

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum = 0.f;  // threads 0..127: column n0+tid of dC

  const int br = tid >> 5;
  const int bc = (tid & 31) * 4;
  f32x4 rx[4], ry[4];
  int okmask = 0;
  float rsv[4] = {1.f, 1.f, 1.f, 1.f};

  auto load_tile = [&](int64_t r0) {
    if constexpr (VEC) {
      // raw loads; select / row scale deferred to the LDS store (see k_gemm)
      const int xn = n0 + bc, yk = c0 + bc;
      const bool x_ok = xn < g.N, y_ok = yk < g.K;
      const int xc = x_ok ? xn : 0, yc = y_ok ? yk : 0;
      int64_t rowi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t pos = r0 + br + 8 * i;
        const int64_t pc = pos < r_end ? pos : r_begin;  // clamped (r_begin < r_end here)
        rowi[i] = GROUPED ? (int64_t)g.row_index[pc] : pc;
      }
      okmask = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool rv = r0 + br + 8 * i < r_end;
        rx[i] = *reinterpret_cast<const f32x4*>(g.X + rowi[i] * g.ldx + xc);
        ry[i] = *reinterpret_cast<const f32x4*>(g.Y + rowi[i] * g.ldy + yc);
        rsv[i] = *(g.rs != nullptr ? g.rs + rowi[i] : &c_one);
        okmask |= (rv && x_ok) ? (1 << i) : 0;
        okmask |= (rv && y_ok) ? (16 << i) : 0;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int64_t gm = r0 + br + 8 * i;
      bool rv = gm < r_end;
      if (GROUPED && rv) gm = g.row_index[gm];
      int nv = rv ? g.N - (n0 + bc) : 0;
      rx[i] = ld4(g.X + gm * g.ldx + n0 + bc, g.vec_x, nv);
      int kv = rv ? g.K - (c0 + bc) : 0;
      f32x4 v = ld4(g.Y + gm * g.ldy + c0 + bc, g.vec_y, kv);
      if (g.rs != nullptr && rv) {
        float sc = g.rs[gm];
        v.x *= sc;
        v.y *= sc;
        v.z *= sc;
        v.w *= sc;
      }
      ry[i] = v;
    }
  };

  int64_t r0 = r_begin;
  if (r0 < r_end) load_tile(r0);
  while (r0 < r_end) {
    __syncthreads();
    if constexpr (VEC) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = ry[i];
        v.x *= rsv[i];
        v.y *= rsv[i];
        v.z *= rsv[i];
        v.w *= rsv[i];
        rx[i] = ((okmask >> i) & 1) ? rx[i] : z;
        ry[i] = ((okmask >> (4 + i)) & 1) ? v : z;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int kr = br + 8 * i;
      *reinterpret_cast<f32x4*>(&Xs[kr * LDN + bc]) = rx[i];
      *reinterpret_cast<f32x4*>(&Ys[kr * LDN + bc]) = ry[i];
    }
    __syncthreads();
    r0 += BK;
    if (r0 < r_end) load_tile(r0);

    if (g.dbias != nullptr && bz == 0 && tid < BN) {
#pragma unroll 8
      for (int r = 0; r < BK; ++r) bsum += Xs[r * LDN + tid];
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 a[2], b[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const float* p = &Xs[(kk * 8 + 4 * lh) * LDN + wm * 64 + mi * 32 + li];
        a[mi].x = p[0];
        a[mi].y = p[LDN];
        a[mi].z = p[2 * LDN];
        a[mi].w = p[3 * LDN];
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const float* p = &Ys[(kk * 8 + 4 * lh) * LDN + wn * 64 + ni * 32 + li];
        b[ni].x = p[0];
        b[ni].y = p[LDN];
        b[ni].z = p[2 * LDN];
        b[ni].w = p[3 * LDN];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][t], b[ni][t], acc[mi][ni], 0, 0, 0);
    }
  }

#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      int gc = c0 + wn * 64 + ni * 32 + li;
      if (gc >= g.K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int gr = n0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (gr >= g.N) continue;
        atomicAdd(dW + (int64_t)gr * g.lddw + gc, acc[mi][ni][r]);
      }
    }
  if (g.dbias != nullptr && bz == 0 && tid < BN && n0 + tid < g.N) atomicAdd(g.dbias + n0 + tid, bsum);
}

template <bool VEC, bool GROUPED>
__global__ void __launch_bounds__(256, 2) k_gemm_wgrad(wgrad_args g) {
  __shared__ __attribute__((aligned(16))) float Xs[BK * LDN];
  __shared__ __attribute__((aligned(16))) float Ys[BK * LDN];
  wgrad_body<VEC, GROUPED>(g, blockIdx.x, blockIdx.y, blockIdx.z, Xs, Ys);
}

// ---------------------------------------------------------------------------------------------------------------
// Split-operand weight gradient (same contract as wgrad_body<true, GROUPED> without a row scale; see the split-operand
// notes at k_gemm_ws3).  dW[n,k] = sum_m dC[m,n] A[m,k] contracts over the ROW index of both operands, so both bf16
// images must be column-major for the matrix core (a lane needs 8 consecutive m of one column).  The transpose is done
// by the loader: thread = (column tid & 127, 16 rows), 16 four-byte loads per operand and 32-row step (each wave
// instruction reads 256 contiguous bytes of one row); the 16 values of a column are split and stored with two
// ds_write_b128 per image into [3][128 columns][32 m (+8 pad)] -- the same images and the same multiply loop as
// k_gemm3.  The fp32-MFMA version above runs at ~70 TF and was the largest group of kernels of the step.
// ---------------------------------------------------------------------------------------------------------------
template <bool GROUPED>
__device__ __forceinline__ void wgrad3_body(const wgrad_args& g, const int bx, const int by, const int bz,
                                            unsigned char* A3, unsigned char* B3, unsigned* offs) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = by * BN;  // dW row tile (output features)
  const int c0 = bz * BN;  // dW col tile (input features)
  if (n0 >= g.N || c0 >= g.K) return;
  int64_t r_begin = (int64_t)bx * g.rows_per_block;
  int64_t r_end = r_begin + g.rows_per_block;
  if (r_end > g.M) r_end = g.M;
  float* dW = g.dW;
  if constexpr (GROUPED) {
    if (bx >= g.nchunks[0]) return;
    r_begin = g.chunk_info[3 * bx];
    r_end = r_begin + g.chunk_info[3 * bx + 1];
    dW += (int64_t)g.chunk_info[3 * bx + 2] * g.dw_cls_stride;
  }
  if (r_begin >= r_end) return;  // (workgroup-uniform) nothing to contribute: skip the zero-valued atomic flush

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum = 0.f;  // column n0 + col of dC, this thread's 16-row half

  const int col = tid & 127, rh = (tid >> 7) * 16;
  const bool x_ok = n0 + col < g.N, y_ok = c0 + col < g.K;
  const float* xp = g.X + (x_ok ? n0 + col : 0);
  const float* yp = g.Y + (y_ok ? c0 + col : 0);
  float gx[16], gy[16];
  int nvalid = 0;  // valid rows among this thread's 16 of the tile in flight

  // GROUPED: rows are gathered through row_index.  The element offsets row * ld of a step's 32 rows are computed once
  // per workgroup (threads 0..31, two steps ahead, into a double-buffered LDS array offs[parity][X|Y][32]) instead of
  // 32 index loads + 32 64-bit multiplies per thread and step; the host guarantees M * ld < 2^32.
  auto stage_offsets = [&](int64_t r0, int par) {
    if constexpr (GROUPED) {
      if (tid < 32) {
        const int64_t pos = r0 + tid < r_end ? r0 + tid : r_begin;
        const unsigned row = (unsigned)g.row_index[pos];
        offs[par * 64 + tid] = row * (unsigned)g.ldx;
        offs[par * 64 + 32 + tid] = row * (unsigned)g.ldy;
      }
    }
  };
  auto load_tile = [&](int64_t r0, int par) {
    const int64_t first = r0 + rh;
    const int64_t left = r_end - first;
    nvalid = left >= 16 ? 16 : (left > 0 ? (int)left : 0);
    if constexpr (GROUPED) {
      const unsigned* ox = offs + par * 64 + rh;
      const unsigned* oy = ox + 32;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        gx[j] = xp[ox[j]];
        gy[j] = yp[oy[j]];
      }
    } else if (r0 + BK <= r_end) {  // full step (workgroup-uniform): one 64-bit product per operand, uniform strides
      const float* xb = xp + first * g.ldx;
      const float* yb = yp + first * g.ldy;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        gx[j] = xb[j * g.ldx];
        gy[j] = yb[j * g.ldy];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int64_t row = first + j < r_end ? first + j : r_begin;  // clamped (r_begin < r_end here)
        gx[j] = xp[row * g.ldx];
        gy[j] = yp[row * g.ldy];
      }
    }
  };

  auto store_tile = [&]() {
#pragma unroll
    for (int hgrp = 0; hgrp < 2; ++hgrp) {
      float xa[8], ya[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = 8 * hgrp + j < nvalid;
        xa[j] = (ok && x_ok) ? gx[8 * hgrp + j] : 0.f;
        ya[j] = (ok && y_ok) ? gy[8 * hgrp + j] : 0.f;
        bsum += xa[j];
      }
      bf16x8 p1, p2, p3;
      split3(xa, p1, p2, p3);
      unsigned char* q = A3 + col * G3_LDB + (rh + 8 * hgrp) * 2;
      *reinterpret_cast<bf16x8*>(q) = p1;
      *reinterpret_cast<bf16x8*>(q + G3_PIECE) = p2;
      *reinterpret_cast<bf16x8*>(q + 2 * G3_PIECE) = p3;
      split3(ya, p1, p2, p3);
      q = B3 + col * G3_LDB + (rh + 8 * hgrp) * 2;
      *reinterpret_cast<bf16x8*>(q) = p1;
      *reinterpret_cast<bf16x8*>(q + G3_PIECE) = p2;
      *reinterpret_cast<bf16x8*>(q + 2 * G3_PIECE) = p3;
    }
  };

  int64_t r0 = r_begin;
  int par = 0;
  stage_offsets(r0, 0);
  stage_offsets(r0 + BK, 1);
  if constexpr (GROUPED) __syncthreads();
  load_tile(r0, 0);
  while (r0 < r_end) {
    __syncthreads();  // previous multiply finished reading LDS (and the offsets staged during it are visible)
    store_tile();
    __syncthreads();
    r0 += BK;
    par ^= 1;
    if (r0 < r_end) load_tile(r0, par);
    // offsets of the step after that one go into the other buffer (last read one step ago, two barriers back)
    if (r0 + BK < r_end) stage_offsets(r0 + BK, par ^ 1);
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          a[mi][p] = *reinterpret_cast<const bf16x8*>(A3 + p * G3_PIECE + (wm * 64 + mi * 32 + li) * G3_LDB + 32 * sl + 16 * lh);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          b[ni][p] = *reinterpret_cast<const bf16x8*>(B3 + p * G3_PIECE + (wn * 64 + ni * 32 + li) * G3_LDB + 32 * sl + 16 * lh);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][2], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][2], b[ni][0], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][1], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][1], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][0], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][0], acc[mi][ni], 0, 0, 0);
        }
    }
  }

#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      int gc = c0 + wn * 64 + ni * 32 + li;
      if (gc >= g.K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int gr = n0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (gr >= g.N) continue;
        atomicAdd(dW + (int64_t)gr * g.lddw + gc, acc[mi][ni][r]);
      }
    }
  if (g.dbias != nullptr && bz == 0 && x_ok) atomicAdd(g.dbias + n0 + col, bsum);
}

template <bool GROUPED>
__global__ void __launch_bounds__(256, 2) k_gemm_wgrad3(wgrad_args g) {
  __shared__ __attribute__((aligned(16))) unsigned char A3[G3_OP];
  __shared__ __attribute__((aligned(16))) unsigned char B3[G3_OP];
  __shared__ __attribute__((aligned(16))) unsigned offs[128];
  wgrad3_body<GROUPED>(g, blockIdx.x, blockIdx.y, blockIdx.z, A3, B3, offs);
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-specialised split-operand weight gradient (same contract and arithmetic as wgrad3_body for 16-byte aligned
// operands with N, K multiples of 4), ONE 512-thread workgroup per CU -- the weight-gradient launches are sized to one
// workgroup per CU anyway, and in the two-barrier body above such a lone workgroup runs its phases back to back: 32
// four-byte loads per thread, wait, split, LDS stores, barrier, fragment reads, 48 MFMAs, barrier = 2.5-2.8 us per
// 32-row step against 0.65-0.87 us of MFMA time (SQ_VALU_MFMA_BUSY_CYCLES: 0.20-0.26 of the matrix pipe; ablations:
// split + LDS stores alone 1.0 us, fragment reads + MFMAs alone 1.27 us, and the compiler keeps the two apart even
// inside one basic block).  Here the phases belong to different waves of the same SIMD, which the hardware overlaps:
//   * waves 0..3 multiply: fragment reads + 48 MFMAs per step out of the LDS stage of step j (2 x 2 tiles of 64 x 64);
//   * waves 4..7 stage: threads 256..383 dC, 384..511 A -- eight 16-byte loads each (8 rows x 4 columns) issued five
//     steps ahead into four register stages (128 KB in flight per CU), always unconditional (invalid rows / columns read a zero line instead of
//     being masked), split column by column and written as 16-byte LDS words into the [column][32 m] bf16 images of
//     step j + 1 (two-stage ring, 120 KB); they also accumulate the bias gradient;
//   * ONE barrier per step joins the two groups.
// Rows gathered by degree class stay on wgrad3_body: the same structure with LDS-staged row offsets was built, correct, and
// SLOWER there (145 vs 123 us per launch: the per-class chunks are 1024 rows = 32 steps, too short for its prologue).
// Measured (tools/wgrad_ab.py, a layer's eight 81 920 x 128 x 128 problems in one launch = 671 MB): 235 -> 205 us (3.4 -> 3.9
// TB/s); K = 512: 111 -> 94 us.  Ablations of this kernel: the loads alone 124 us (the HBM floor), + the multiply waves
// 132 us, + the staging waves' split and stores instead 138 us, both 195-205 us: the two groups slow each other down.
// tools/ubench/wave_specialised_overlap.hip isolates that: per step, multiply waves alone 0.91 us, staging waves alone
// 0.60 us (split) / 0.84 us (+ the twelve ds_write_b128), both 1.04 us without and 1.41-1.46 us with the LDS stores --
// the 16-byte LDS stores, not the split, are what the MFMA waves feel; 1.45 us is this structure's floor, the kernel
// runs at 2.2 us (loads, masks, address arithmetic, bias sums on top).
// ---------------------------------------------------------------------------------------------------------------
#define WG3P_LDS (4 * G3_OP)
#define WG3P_NST 4  // register stages of the staging waves (must be 4: the prologue and the unrolled loop assume it)

__device__ __forceinline__ void wgrad3p_body(const wgrad_args& g, const int bx, const int by, const int bz,
                                             unsigned char* lds) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int n0 = by * BN;  // dW row tile (output features)
  const int c0 = bz * BN;  // dW col tile (input features)
  if (n0 >= g.N || c0 >= g.K) return;
  int64_t r_begin = (int64_t)bx * g.rows_per_block;
  int64_t r_end = r_begin + g.rows_per_block;
  if (r_end > g.M) r_end = g.M;
  float* const dW = g.dW;
  if (r_begin >= r_end) return;
  const int64_t nsteps = (r_end - r_begin + BK - 1) / BK;

  if (wave < 4) {
    // ================================================================ multiply waves
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    __syncthreads();  // the staging waves' prologue
    for (int64_t j = 0; j < nsteps; ++j) {
      const unsigned char* const A3 = lds + (j & 1) * 2 * G3_OP;
      const unsigned char* const B3 = A3 + G3_OP;
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int p = 0; p < 3; ++p)
            a[mi][p] = *reinterpret_cast<const bf16x8*>(A3 + p * G3_PIECE + (wm * 64 + mi * 32 + li) * G3_LDB + 32 * sl + 16 * lh);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int p = 0; p < 3; ++p)
            b[ni][p] = *reinterpret_cast<const bf16x8*>(B3 + p * G3_PIECE + (wn * 64 + ni * 32 + li) * G3_LDB + 32 * sl + 16 * lh);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][2], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][2], b[ni][0], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][1], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][1], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][0], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][0], acc[mi][ni], 0, 0, 0);
          }
      }
      __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        int gc = c0 + wn * 64 + ni * 32 + li;
        if (gc >= g.K) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int gr = n0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (gr >= g.N) continue;
          atomicAdd(dW + (int64_t)gr * g.lddw + gc, acc[mi][ni][r]);
        }
      }
    return;
  }

  // ================================================================== staging waves
  const int lt = tid - 256;
  const int role = lt >> 7;  // 0: dC (dW rows n0..), 1: A (dW columns c0..)
  const int c4 = (lt & 31) * 4, rg = (lt >> 5) & 3;
  const bool col_ok = role ? (c0 + c4 < g.K) : (n0 + c4 < g.N);  // N, K multiples of 4: the quad is valid as a whole
  const float* const base = (role ? g.Y + (col_ok ? c0 + c4 : 0) : g.X + (col_ok ? n0 + c4 : 0));
  const float* const zero = g.zero;  // 16 zero bytes: what an invalid row or column quad reads
  const int64_t ld = role ? g.ldy : g.ldx;
  const int img = role * G3_OP;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};  // role 0: column sums of dC (bias gradient)

  f32x4 v[WG3P_NST][8];  // register stages: the loads of WG3P_NST steps in flight (32 KB per stage and workgroup)

  auto load_step = [&](auto PC, int64_t r0) {
    constexpr int P = decltype(PC)::value;
    const int64_t left = r_end - (r0 + rg * 8);
    const int nval = (left >= 8 && col_ok) ? 8 : ((left > 0 && col_ok) ? (int)left : 0);
    const float* p = base + (r0 + rg * 8) * ld;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[P][j] = *reinterpret_cast<const f32x4*>(j < nval ? p + j * ld : zero);
  };
  auto store_step = [&](auto PC, unsigned char* stage) {
    constexpr int P = decltype(PC)::value;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float xa[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) xa[j] = v[P][j][q];
      if (role == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[q] += xa[j];
      }
      bf16x8 p1, p2, p3;
      split3(xa, p1, p2, p3);
      unsigned char* dst = stage + img + (c4 + q) * G3_LDB + rg * 16;
      *reinterpret_cast<bf16x8*>(dst) = p1;
      *reinterpret_cast<bf16x8*>(dst + G3_PIECE) = p2;
      *reinterpret_cast<bf16x8*>(dst + 2 * G3_PIECE) = p3;
    }
  };
  auto row_of = [&](int64_t step) { return r_begin + step * BK; };

  // ---- prologue: steps 0 .. 3 loaded, step 0 split into LDS stage 0, step 4 loaded into its register stage
  load_step(std::integral_constant<int, 0>{}, row_of(0));
  load_step(std::integral_constant<int, 1>{}, row_of(1));
  load_step(std::integral_constant<int, 2>{}, row_of(2));
  load_step(std::integral_constant<int, 3>{}, row_of(3));
  store_step(std::integral_constant<int, 0>{}, lds);
  load_step(std::integral_constant<int, 0>{}, row_of(4));
  __syncthreads();

  // iteration j (K = j % NST): split register stage (K + 1) % NST = step j + 1 into LDS stage (j + 1) & 1, then refill
  // that register stage with step j + 1 + NST
  auto iteration = [&](auto KC, int64_t j) {
    constexpr int K = decltype(KC)::value;
    constexpr int P = (K + 1) % WG3P_NST;
    store_step(std::integral_constant<int, P>{}, lds + ((K + 1) & 1) * 2 * G3_OP);
    load_step(std::integral_constant<int, P>{}, row_of(j + 1 + WG3P_NST));
    __syncthreads();
  };
  for (int64_t j = 0; j < nsteps; j += 4) {
    iteration(std::integral_constant<int, 0>{}, j);
    if (j + 1 >= nsteps) break;
    iteration(std::integral_constant<int, 1>{}, j + 1);
    if (j + 2 >= nsteps) break;
    iteration(std::integral_constant<int, 2>{}, j + 2);
    if (j + 3 >= nsteps) break;
    iteration(std::integral_constant<int, 3>{}, j + 3);
  }
  if (g.dbias != nullptr && bz == 0 && role == 0 && col_ok) {
#pragma unroll
    for (int q = 0; q < 4; ++q) atomicAdd(g.dbias + n0 + c4 + q, bsum[q]);
  }
}

__global__ void __launch_bounds__(512, 1) k_gemm_wgrad3p(wgrad_args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_w[];
  wgrad3p_body(g, blockIdx.x, blockIdx.y, blockIdx.z, lds_w);
}

// Several independent weight gradients in ONE launch (a layer's same-shaped dW = g^T a products): blockIdx.x =
// problem * chunks + chunk.  With P problems sharing the grid every workgroup owns a P x longer row range, so the
// per-problem atomic flush shrinks P x at equal parallelism (a lone 128x128 dW over 82k rows flushes 33 MB of fp32
// atomics for 17 us of MFMA work).
#define WGRAD_MAX_BATCH 8
struct wgrad_batch_args {
  wgrad_args p[WGRAD_MAX_BATCH];
  int nprob;
  // 1-D work list: problem i owns workgroups [wg_off[i], wg_off[i+1]) = its row chunks x its 128x128 output tiles.
  // Chunk counts are per problem (proportional to its share of the work): one uniform count starved the small
  // problems of a batch that also held a 512x512 one (cfg-5: 32 workgroups busy on 256 CUs, 26 TF).
  int wg_off[WGRAD_MAX_BATCH + 1];
  int tiles_k[WGRAD_MAX_BATCH];  // column tiles of dW
  int tiles[WGRAD_MAX_BATCH];    // row tiles x column tiles of dW
};

// (problem, chunk, dW row tile, dW column tile) of a workgroup of the 1-D batched grid
__device__ __forceinline__ void wgrad_batch_locate(const wgrad_batch_args& b, int w, int& prob, int& chunk, int& by,
                                                   int& bz) {
  prob = 0;
#pragma unroll
  for (int i = 1; i < WGRAD_MAX_BATCH; ++i) prob += (i < b.nprob && w >= b.wg_off[i]) ? 1 : 0;
  int off = b.wg_off[0], tiles = b.tiles[0], tk = b.tiles_k[0];
#pragma unroll
  for (int i = 1; i < WGRAD_MAX_BATCH; ++i)
    if (i == prob) {
      off = b.wg_off[i];
      tiles = b.tiles[i];
      tk = b.tiles_k[i];
    }
  const int local = w - off;
  chunk = local / tiles;
  const int t = local - chunk * tiles;
  by = t / tk;
  bz = t - by * tk;
}

__global__ void __launch_bounds__(256, 2) k_gemm_wgrad_batched(wgrad_batch_args b) {
  __shared__ __attribute__((aligned(16))) float Xs[BK * LDN];
  __shared__ __attribute__((aligned(16))) float Ys[BK * LDN];
  int prob, chunk, by, bz;
  wgrad_batch_locate(b, blockIdx.x, prob, chunk, by, bz);
  // copy the selected descriptor (wave-uniform index) so the body sees scalars
  wgrad_args g = b.p[0];
#pragma unroll
  for (int i = 1; i < WGRAD_MAX_BATCH; ++i)
    if (i == prob) g = b.p[i];
  wgrad_body<true, false>(g, chunk, by, bz, Xs, Ys);
}

__global__ void __launch_bounds__(256, 2) k_gemm_wgrad3_batched(wgrad_batch_args b) {
  __shared__ __attribute__((aligned(16))) unsigned char A3[G3_OP];
  __shared__ __attribute__((aligned(16))) unsigned char B3[G3_OP];
  int prob, chunk, by, bz;
  wgrad_batch_locate(b, blockIdx.x, prob, chunk, by, bz);
  wgrad_args g = b.p[0];
#pragma unroll
  for (int i = 1; i < WGRAD_MAX_BATCH; ++i)
    if (i == prob) g = b.p[i];
  wgrad3_body<false>(g, chunk, by, bz, A3, B3, nullptr);
}

__global__ void __launch_bounds__(512, 1) k_gemm_wgrad3p_batched(wgrad_batch_args b) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_wb[];
  int prob, chunk, by, bz;
  wgrad_batch_locate(b, blockIdx.x, prob, chunk, by, bz);
  wgrad_args g = b.p[0];
#pragma unroll
  for (int i = 1; i < WGRAD_MAX_BATCH; ++i)
    if (i == prob) g = b.p[i];
  wgrad3p_body(g, chunk, by, bz, lds_wb);
}

template <typename KERNEL>
static hipError_t wgrad3p_attr(KERNEL kern) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG3P_LDS);
}

// the split-operand weight-gradient kernels take over for large row counts (GNX_GEMM_SPLIT=0: fp32 MFMA everywhere)
static bool wgrad_split_enabled(const gnx_handle* h, int64_t M, bool any_rowscale) {
  if (any_rowscale || M < 4096) return false;
  return h->opt[GNX_OPT_GEMM_SPLIT] != 0;
}

static int32_t wgrad_launch(gnx_handle* h, const float* dC, int64_t lddc, const float* A, int64_t lda,
                            const float* rowscale, int64_t M, int32_t N, int32_t K, float* dW, int64_t lddw,
                            float* dbias, const int32_t* row_index, const int32_t* chunk_info, const int32_t* nchunks,
                            int64_t max_chunks, int64_t dw_cls_stride) {
  GNX_CHECK_ARG(h && dC && A && dW, "gnx_gemm_wgrad: NULL argument");
  GNX_CHECK_ARG(M >= 0 && N > 0 && K > 0 && lddc >= N && lda >= K && lddw >= K, "gnx_gemm_wgrad: bad shape");
  if (M == 0) return GNX_OK;
  wgrad_args g;
  g.X = dC;
  g.ldx = lddc;
  g.Y = A;
  g.ldy = lda;
  g.rs = rowscale;
  g.M = M;
  g.N = N;
  g.K = K;
  g.dW = dW;
  g.lddw = lddw;
  g.dbias = dbias;
  g.vec_x = aligned16(dC) && (lddc % 4 == 0);
  g.vec_y = aligned16(A) && (lda % 4 == 0);
  int64_t tiles = gnx_cdiv(N, BN) * gnx_cdiv(K, BN);
  // aim for ~512 workgroups; at least 128 rows each (4 K-steps) so the atomic flush stays amortised
  // default: one workgroup per CU.  Measured on cfg-2 (tools/ab_bench.py, same box): 512 / 1024 workgroups cost 0.4 ms
  // per step more -- every workgroup flushes its 128 x 128 partial sum with fp32 atomics, and a weight-gradient launch
  // that fills every CU slot starves the input-gradient chain it overlaps with on the main stream.
  const int64_t target_wgs = h->opt[GNX_OPT_WGRAD_WGS] > 0 ? h->opt[GNX_OPT_WGRAD_WGS] : (h->num_cus > 0 ? h->num_cus : 256);
  int64_t chunks = gnx_cdiv(target_wgs, tiles);
  int64_t rows = gnx_cdiv(gnx_cdiv(M, chunks), BK) * BK;
  if (rows < 128) rows = 128;
  g.rows_per_block = rows;
  g.row_index = row_index;
  g.chunk_info = chunk_info;
  g.nchunks = nchunks;
  g.dw_cls_stride = dw_cls_stride;
  g.zero = h->d_zero;
  dim3 grid((unsigned)(chunk_info ? max_chunks : gnx_cdiv(M, rows)), (unsigned)gnx_cdiv(N, BN), (unsigned)gnx_cdiv(K, BN));
  bool vec = g.vec_x && g.vec_y && (N % 4 == 0) && (K % 4 == 0);
  if (h->opt[GNX_OPT_WGRAD_VEC] == 0) vec = false;
  const bool offs32 = (uint64_t)M * (uint64_t)lddc < (1ull << 32) && (uint64_t)M * (uint64_t)lda < (1ull << 32);
  const bool wsplit = wgrad_split_enabled(h, M, rowscale != nullptr) && (!chunk_info || offs32);
  const double wfl = 2.0 * (double)M * N * K;
  gnx_prof_scope prof(h, GNX_K_GEMM_WGRAD, 4.0 * M * ((double)N + K) + 4.0 * N * K, wfl, wsplit ? 6.0 * wfl : 0.0);
  // (short row ranges -- the readout's 4096-row problems -- keep the two-barrier kernel: 14.7 vs 24 us)
  if (wsplit && vec && !chunk_info && rows >= 512 && h->opt[GNX_OPT_WGRAD_PIPE] != 0) {
    static bool attr_set = false;
    if (!attr_set) {
      GNX_HIP(wgrad3p_attr(&k_gemm_wgrad3p));
      attr_set = true;
    }
    hipLaunchKernelGGL(k_gemm_wgrad3p, grid, dim3(512), WG3P_LDS, h->stream, g);
  } else if (wsplit) {
    if (chunk_info)
      hipLaunchKernelGGL((k_gemm_wgrad3<true>), grid, dim3(256), 0, h->stream, g);
    else
      hipLaunchKernelGGL((k_gemm_wgrad3<false>), grid, dim3(256), 0, h->stream, g);
  } else if (chunk_info) {
    if (vec)
      hipLaunchKernelGGL((k_gemm_wgrad<true, true>), grid, dim3(256), 0, h->stream, g);
    else
      hipLaunchKernelGGL((k_gemm_wgrad<false, true>), grid, dim3(256), 0, h->stream, g);
  } else {
    if (vec)
      hipLaunchKernelGGL((k_gemm_wgrad<true, false>), grid, dim3(256), 0, h->stream, g);
    else
      hipLaunchKernelGGL((k_gemm_wgrad<false, false>), grid, dim3(256), 0, h->stream, g);
  }
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

extern "C" int32_t gnx_gemm_wgrad(gnx_handle* h, const float* dC, int64_t lddc, const float* A, int64_t lda,
                                  const float* rowscale, int64_t M, int32_t N, int32_t K, float* dW, int64_t lddw,
                                  float* dbias) {
  return wgrad_launch(h, dC, lddc, A, lda, rowscale, M, N, K, dW, lddw, dbias, nullptr, nullptr, nullptr, 0, 0);
}

extern "C" int32_t gnx_gemm_wgrad_grouped(gnx_handle* h, const float* dC, int64_t lddc, const float* A, int64_t lda,
                                          int64_t M, int32_t N, int32_t K, float* dW_cls, int64_t lddw,
                                          int64_t dw_cls_stride, const int32_t* row_index, const int32_t* chunk_info,
                                          const int32_t* nchunks, int64_t max_chunks) {
  GNX_CHECK_ARG(row_index && chunk_info && nchunks && max_chunks > 0, "gnx_gemm_wgrad_grouped: NULL argument");
  return wgrad_launch(h, dC, lddc, A, lda, nullptr, M, N, K, dW_cls, lddw, nullptr, row_index, chunk_info, nchunks,
                      max_chunks, dw_cls_stride);
}

// ---------------------------------------------------------------------------------------------------------------
// PNA post-layer 0 effective weights per in-degree class d (amp/att depend on d only):
//   Weff[d][o][j] = W[o][F + j] + amp(d) W[o][5F + j] + att(d) W[o][9F + j],   o < F, j < 4F
// and the matching weight gradient:  dW[:, F:5F] += sum_d dWeff[d], [:, 5F:9F] += sum_d amp(d) dWeff[d], ...
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_pna_weff(const float* __restrict__ W, int64_t ldw, int F, int D, float avg_log,
                           float* __restrict__ Weff) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t per = (int64_t)F * 4 * F;
  if (i >= per * D) return;
  int d = (int)(i / per);
  int o = (int)((i % per) / (4 * F)), j = (int)(i % (4 * F));
  float dd = (float)d;
  float amp = logf(dd + 1.0f) / avg_log;
  float att = avg_log / logf(fmaxf(dd, 1.0f) + 1.0f);
  const float* w = W + (int64_t)o * ldw;
  Weff[i] = w[F + j] + amp * w[5 * F + j] + att * w[9 * F + j];
}

__global__ void k_pna_weff_bwd(const float* __restrict__ dWeff, int F, int D, float avg_log, float* __restrict__ dW,
                               int64_t lddw) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t per = (int64_t)F * 4 * F;
  if (i >= per) return;
  int o = (int)(i / (4 * F)), j = (int)(i % (4 * F));
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int d = 0; d < D; ++d) {
    float dd = (float)d;
    float amp = logf(dd + 1.0f) / avg_log;
    float att = avg_log / logf(fmaxf(dd, 1.0f) + 1.0f);
    float v = dWeff[(int64_t)d * per + i];
    s1 += v;
    s2 += amp * v;
    s3 += att * v;
  }
  float* w = dW + (int64_t)o * lddw;
  w[F + j] += s1;
  w[5 * F + j] += s2;
  w[9 * F + j] += s3;
}

// the same for up to GNX_SMALL_BATCH (layer, tower) pairs in one launch (blockIdx.y = pair)
struct weff_batch_args {
  const float* W[GNX_SMALL_BATCH];
  float* out[GNX_SMALL_BATCH];
  float avg_log[GNX_SMALL_BATCH];
  int64_t ldw;
  int F, D;
};

__global__ void k_pna_weff_batched(weff_batch_args a) {
  const int b = blockIdx.y;
  const int F = a.F;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t per = (int64_t)F * 4 * F;
  if (i >= per * a.D) return;
  int d = (int)(i / per);
  int o = (int)((i % per) / (4 * F)), j = (int)(i % (4 * F));
  float dd = (float)d;
  float amp = logf(dd + 1.0f) / a.avg_log[b];
  float att = a.avg_log[b] / logf(fmaxf(dd, 1.0f) + 1.0f);
  const float* w = a.W[b] + (int64_t)o * a.ldw;
  a.out[b][i] = w[F + j] + amp * w[5 * F + j] + att * w[9 * F + j];
}

struct weff_bwd_batch_args {
  const float* dWeff[GNX_SMALL_BATCH];
  float* dW[GNX_SMALL_BATCH];
  float avg_log[GNX_SMALL_BATCH];
  int64_t lddw;
  int F, D;
};

__global__ void k_pna_weff_bwd_batched(weff_bwd_batch_args a) {
  const int b = blockIdx.y;
  const int F = a.F;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t per = (int64_t)F * 4 * F;
  if (i >= per) return;
  int o = (int)(i / (4 * F)), j = (int)(i % (4 * F));
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const float avg_log = a.avg_log[b];
  const float* dWeff = a.dWeff[b];
  for (int d = 0; d < a.D; ++d) {  // same order and expressions as k_pna_weff_bwd
    float dd = (float)d;
    float amp = logf(dd + 1.0f) / avg_log;
    float att = avg_log / logf(fmaxf(dd, 1.0f) + 1.0f);
    float v = dWeff[(int64_t)d * per + i];
    s1 += v;
    s2 += amp * v;
    s3 += att * v;
  }
  float* w = a.dW[b] + (int64_t)o * a.lddw;
  w[F + j] += s1;
  w[5 * F + j] += s2;
  w[9 * F + j] += s3;
}

extern "C" int32_t gnx_pna_weff_bwd_batched(gnx_handle* h, int32_t n, const float* const* dWeff, int32_t F, int32_t D,
                                            const float* avg_deg_log, float* const* dW, int64_t lddw) {
  GNX_CHECK_ARG(h && dWeff && dW && avg_deg_log && n >= 0 && F > 0 && D > 0 && lddw >= 13 * F,
                "gnx_pna_weff_bwd_batched: bad argument");
  const int64_t cnt = (int64_t)F * 4 * F;
  for (int32_t i0 = 0; i0 < n; i0 += GNX_SMALL_BATCH) {
    weff_bwd_batch_args a;
    const int m = n - i0 < GNX_SMALL_BATCH ? n - i0 : GNX_SMALL_BATCH;
    for (int i = 0; i < GNX_SMALL_BATCH; ++i) {
      const int k = i < m ? i0 + i : i0;
      a.dWeff[i] = dWeff[k];
      a.dW[i] = dW[k];
      a.avg_log[i] = avg_deg_log[k];
      GNX_CHECK_ARG(a.dWeff[i] && a.dW[i], "gnx_pna_weff_bwd_batched: NULL pointer at %d", k);
    }
    a.lddw = lddw;
    a.F = F;
    a.D = D;
    hipLaunchKernelGGL(k_pna_weff_bwd_batched, dim3((unsigned)gnx_cdiv(cnt, 256), (unsigned)m), dim3(256), 0, h->stream, a);
    GNX_LAUNCH_CHECK();
  }
  return GNX_OK;
}

// W[i] / out[i]: HOST arrays of n device pointers (post_nns[t][0].weight [F,13F] -> Weff [D,F,4F]); avg_log: HOST [n]
extern "C" int32_t gnx_pna_weff_batched(gnx_handle* h, int32_t n, const float* const* W, int64_t ldw, int32_t F, int32_t D,
                                        const float* avg_deg_log, float* const* Weff) {
  GNX_CHECK_ARG(h && W && Weff && avg_deg_log && n >= 0 && F > 0 && D > 0 && ldw >= 13 * F, "gnx_pna_weff_batched: bad argument");
  const int64_t cnt = (int64_t)F * 4 * F * D;
  for (int32_t i0 = 0; i0 < n; i0 += GNX_SMALL_BATCH) {
    weff_batch_args a;
    const int m = n - i0 < GNX_SMALL_BATCH ? n - i0 : GNX_SMALL_BATCH;
    for (int i = 0; i < GNX_SMALL_BATCH; ++i) {
      const int k = i < m ? i0 + i : i0;
      a.W[i] = W[k];
      a.out[i] = Weff[k];
      a.avg_log[i] = avg_deg_log[k];
      GNX_CHECK_ARG(a.W[i] && a.out[i], "gnx_pna_weff_batched: NULL pointer at %d", k);
    }
    a.ldw = ldw;
    a.F = F;
    a.D = D;
    hipLaunchKernelGGL(k_pna_weff_batched, dim3((unsigned)gnx_cdiv(cnt, 256), (unsigned)m), dim3(256), 0, h->stream, a);
    GNX_LAUNCH_CHECK();
  }
  return GNX_OK;
}

extern "C" int32_t gnx_pna_weff(gnx_handle* h, const float* W, int64_t ldw, int32_t F, int32_t D, float avg_deg_log,
                                float* Weff) {
  GNX_CHECK_ARG(h && W && Weff && F > 0 && D > 0 && ldw >= 13 * F, "gnx_pna_weff: bad argument");
  int64_t n = (int64_t)F * 4 * F * D;
  hipLaunchKernelGGL(k_pna_weff, dim3((unsigned)gnx_cdiv(n, 256)), dim3(256), 0, h->stream, W, ldw, (int)F, (int)D,
                     avg_deg_log, Weff);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

extern "C" int32_t gnx_pna_weff_bwd(gnx_handle* h, const float* dWeff, int32_t F, int32_t D, float avg_deg_log,
                                    float* dW, int64_t lddw) {
  GNX_CHECK_ARG(h && dWeff && dW && F > 0 && D > 0 && lddw >= 13 * F, "gnx_pna_weff_bwd: bad argument");
  int64_t n = (int64_t)F * 4 * F;
  hipLaunchKernelGGL(k_pna_weff_bwd, dim3((unsigned)gnx_cdiv(n, 256)), dim3(256), 0, h->stream, dWeff, (int)F, (int)D,
                     avg_deg_log, dW, lddw);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

extern "C" int32_t gnx_gemm_wgrad_batched(gnx_handle* h, int32_t nprob, const gnx_wgrad_prob* probs) {
  GNX_CHECK_ARG(h && probs && nprob >= 1 && nprob <= WGRAD_MAX_BATCH, "gnx_gemm_wgrad_batched: nprob must be in [1,%d]",
                WGRAD_MAX_BATCH);
  wgrad_batch_args b;
  int64_t maxM = 0;
  int maxN = 0, maxK = 0;
  for (int i = 0; i < nprob; ++i) {
    const gnx_wgrad_prob& q = probs[i];
    GNX_CHECK_ARG(q.dC && q.A && q.dW && q.M >= 0 && q.N > 0 && q.K > 0 && q.lddc >= q.N && q.lda >= q.K && q.lddw >= q.K,
                  "gnx_gemm_wgrad_batched: problem %d: bad argument", i);
    const bool vec = aligned16(q.dC) && (q.lddc % 4 == 0) && aligned16(q.A) && (q.lda % 4 == 0) && (q.N % 4 == 0) &&
                     (q.K % 4 == 0);
    GNX_CHECK_ARG(vec, "gnx_gemm_wgrad_batched: problem %d is not 16-byte aligned / multiple-of-4 shaped", i);
    wgrad_args g;
    g.X = q.dC;
    g.ldx = q.lddc;
    g.Y = q.A;
    g.ldy = q.lda;
    g.rs = q.rowscale;
    g.M = q.M;
    g.N = q.N;
    g.K = q.K;
    g.dW = q.dW;
    g.lddw = q.lddw;
    g.dbias = q.dbias;
    g.vec_x = g.vec_y = 1;
    g.row_index = nullptr;
    g.chunk_info = nullptr;
    g.nchunks = nullptr;
    g.dw_cls_stride = 0;
    g.rows_per_block = 0;
    g.zero = h->d_zero;
    b.p[i] = g;
    if (q.M > maxM) maxM = q.M;
    if (q.N > maxN) maxN = q.N;
    if (q.K > maxK) maxK = q.K;
  }
  for (int i = nprob; i < WGRAD_MAX_BATCH; ++i) b.p[i] = b.p[0];
  if (maxM == 0) return GNX_OK;
  // ~1024 workgroups in total, shared out by work (rows x output tiles); every chunk is a multiple of 32 rows
  double total_cost = 0.0;
  for (int i = 0; i < nprob; ++i)
    total_cost += (double)b.p[i].M * (double)(gnx_cdiv(b.p[i].N, BN) * gnx_cdiv(b.p[i].K, BN));
  int off = 0;
  for (int i = 0; i < WGRAD_MAX_BATCH; ++i) {
    b.wg_off[i] = off;
    b.tiles[i] = 1;
    b.tiles_k[i] = 1;
    if (i >= nprob) continue;
    const int tn = (int)gnx_cdiv(b.p[i].N, BN), tk = (int)gnx_cdiv(b.p[i].K, BN);
    const int64_t M = b.p[i].M > 0 ? b.p[i].M : 1;
    // workgroups of the launch: at most one per CU, and at least ~5120 rows of a 128 x 128 output tile each (every
    // workgroup ends with a 64 KB fp32 atomic flush and, on its CU, displaces the main stream's workgroups: at cfg-2's
    // 650 k row-tiles per layer 160 workgroups beat 256 by 1.7 % of the step in round 2; with round 3's shorter main
    // stream 128 beat 160 / 96 / 192 / 256: 6.765 vs 6.836 / 6.932 / 6.880 / 6.880 ms; at cfg-3/4/5's sizes 256 are best)
    const double cus_d = (double)(h->num_cus > 0 ? h->num_cus : 256);
    double auto_budget = total_cost / 5120.0;
    auto_budget = auto_budget < cus_d / 4 ? cus_d / 4 : (auto_budget > cus_d ? cus_d : auto_budget);
    const double budget = h->opt[GNX_OPT_WGRAD_WGS] > 0 ? (double)h->opt[GNX_OPT_WGRAD_WGS] : auto_budget;
    int64_t chunks = (int64_t)(budget * ((double)M * tn * tk / total_cost) / (tn * tk) + 0.5);
    const int64_t max_chunks = gnx_cdiv(M, 128);
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    const int64_t rows = gnx_cdiv(gnx_cdiv(M, chunks), BK) * BK;
    chunks = gnx_cdiv(M, rows);  // no empty chunks
    b.p[i].rows_per_block = rows;
    b.tiles[i] = tn * tk;
    b.tiles_k[i] = tk;
    off += (int)chunks * tn * tk;
  }
  b.wg_off[WGRAD_MAX_BATCH] = off;
  for (int i = nprob; i < WGRAD_MAX_BATCH; ++i) b.wg_off[i] = off;
  b.nprob = nprob;
  dim3 grid((unsigned)off);
  bool any_rs = false;
  for (int i = 0; i < nprob; ++i) any_rs = any_rs || probs[i].rowscale != nullptr;
  double wby = 0.0, wfl = 0.0;
  for (int i = 0; i < nprob; ++i) {
    wby += 4.0 * probs[i].M * ((double)probs[i].N + probs[i].K) + 4.0 * probs[i].N * probs[i].K;
    wfl += 2.0 * (double)probs[i].M * probs[i].N * probs[i].K;
  }
  gnx_prof_scope prof(h, GNX_K_GEMM_WGRAD_BATCHED, wby, wfl, wgrad_split_enabled(h, maxM, any_rs) ? 6.0 * wfl : 0.0);
  if (wgrad_split_enabled(h, maxM, any_rs) && h->opt[GNX_OPT_WGRAD_PIPE] != 0) {
    static bool attr_set = false;
    if (!attr_set) {
      GNX_HIP(wgrad3p_attr(&k_gemm_wgrad3p_batched));
      attr_set = true;
    }
    hipLaunchKernelGGL(k_gemm_wgrad3p_batched, grid, dim3(512), WG3P_LDS, h->stream, b);
  } else if (wgrad_split_enabled(h, maxM, any_rs))
    hipLaunchKernelGGL(k_gemm_wgrad3_batched, grid, dim3(256), 0, h->stream, b);
  else
    hipLaunchKernelGGL(k_gemm_wgrad_batched, grid, dim3(256), 0, h->stream, b);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Embedding backward on the matrix cores: dTable[r, :] += sum_n onehot[n, r] * dOut[n, :], where onehot[n, r] counts the
// features k with offs[k] + idx[n,k] == r ([3P] embedding_dense_backward summed over the K tables of an ogb encoder).
// It is the weight-gradient contraction over rows with a left operand GENERATED from the integer features (exact: the
// products are by 0/1), so the 10^8 LDS atomic lane-ops of the table-scatter kernel (~2 clk each) become ~5 GFLOP of MFMA.
// grid = (row chunks, R tiles of 128, H tiles of 128).
// ---------------------------------------------------------------------------------------------------------------
#define EMB_MAX_K 12
struct embed_bwd_args {
  const int64_t* idx;  // [N, K]
  int offs[EMB_MAX_K + 1];
  int K;
  const float* Y;  // dOut [N, H]
  int64_t ldy;
  int64_t N;
  int R, H;
  float* dT;  // [R, H]
  int64_t rows_per_block;
};

__global__ void __launch_bounds__(256, 2) k_embed_bwd_mfma(embed_bwd_args g) {
  __shared__ __attribute__((aligned(16))) float Xs[BK * LDN];
  __shared__ __attribute__((aligned(16))) float Ys[BK * LDN];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * BN;  // table-row tile
  const int c0 = blockIdx.z * BN;  // channel tile
  const int64_t r_begin = (int64_t)blockIdx.x * g.rows_per_block;
  int64_t r_end = r_begin + g.rows_per_block;
  if (r_end > g.N) r_end = g.N;
  if (r_begin >= r_end) return;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int br = tid >> 5;
  const int bc = (tid & 31) * 4;
  const int yk = c0 + bc;
  const bool y_ok = yk < g.H;
  const int yc = y_ok ? yk : 0;
  f32x4 ry[4];
  int fr[4][EMB_MAX_K];  // table rows hit by the 4 batch rows this thread stages (raw loads, used at LDS-store time)
  int okmask = 0;

  auto load_tile = [&](int64_t r0) {
    okmask = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t pos = r0 + br + 8 * i;
      const bool rv = pos < r_end;
      const int64_t row = rv ? pos : r_begin;
      ry[i] = *reinterpret_cast<const f32x4*>(g.Y + row * g.ldy + yc);
#pragma unroll
      for (int k = 0; k < EMB_MAX_K; ++k) fr[i][k] = (int)g.idx[row * g.K + (k < g.K ? k : 0)];  // unconditional
      okmask |= rv ? (1 << i) : 0;
    }
  };

  int64_t r0 = r_begin;
  load_tile(r0);
  while (r0 < r_end) {
    __syncthreads();
    {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool rv = (okmask >> i) & 1;
        f32x4 x = z;
#pragma unroll
        for (int k = 0; k < EMB_MAX_K; ++k) {
          const int span = g.offs[k + 1 <= g.K ? k + 1 : g.K] - g.offs[k < g.K ? k : g.K];
          const int f = fr[i][k];
          const int r = g.offs[k < g.K ? k : 0] + f - (n0 + bc);  // position relative to this thread's 4 table rows
          const bool hit = rv && k < g.K && f >= 0 && f < span;
          x.x += (hit && r == 0) ? 1.f : 0.f;
          x.y += (hit && r == 1) ? 1.f : 0.f;
          x.z += (hit && r == 2) ? 1.f : 0.f;
          x.w += (hit && r == 3) ? 1.f : 0.f;
        }
        const int kr = br + 8 * i;
        *reinterpret_cast<f32x4*>(&Xs[kr * LDN + bc]) = x;
        *reinterpret_cast<f32x4*>(&Ys[kr * LDN + bc]) = (rv && y_ok) ? ry[i] : z;
      }
    }
    __syncthreads();
    r0 += BK;
    if (r0 < r_end) load_tile(r0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 a[2], b[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const float* p = &Xs[(kk * 8 + 4 * lh) * LDN + wm * 64 + mi * 32 + li];
        a[mi].x = p[0];
        a[mi].y = p[LDN];
        a[mi].z = p[2 * LDN];
        a[mi].w = p[3 * LDN];
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const float* p = &Ys[(kk * 8 + 4 * lh) * LDN + wn * 64 + ni * 32 + li];
        b[ni].x = p[0];
        b[ni].y = p[LDN];
        b[ni].z = p[2 * LDN];
        b[ni].w = p[3 * LDN];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][t], b[ni][t], acc[mi][ni], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int gc = c0 + wn * 64 + ni * 32 + li;
      if (gc >= g.H) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gr = n0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (gr >= g.R) continue;
        const float v = acc[mi][ni][r];
        if (v != 0.f) atomicAdd(g.dT + (int64_t)gr * g.H + gc, v);
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same gradient on the bf16 matrix pipe (R <= 192 table rows, e.g. the 174 rows of the nine atom-feature tables):
// the one-hot operand is EXACT in bf16 and is not computed but SCATTERED -- its [table row][batch row] LDS image stays
// zero except for the K ones per batch row, which the thread that set them clears again after the multiply --, the
// gradient goes through the three-piece split like every other product (three bf16 MFMAs per slab instead of eight
// fp32 ones at a sixteenth of the rate: the fp32 kernel above spends 34 of its 107 us in the matrix pipe and most of the
// rest building the one-hot operand with K x 4 compares per element).  A workgroup owns ALL table rows (six 32-row
// blocks) x 128 channels (wave w: channels 32 w ..) of a row chunk; 32 batch rows per step; fp32 atomics at the end.
// ---------------------------------------------------------------------------------------------------------------
#define EB_R 192
#define EB_LDX 80   // bytes per image row: 32 batch rows of bf16 + 16 (odd number of 16-byte slots)
#define EB_X_BYTES (EB_R * EB_LDX)
#define EB_Y_BYTES (128 * EB_LDX)
__global__ void __launch_bounds__(256, 2) k_embed_bwd_bf16(embed_bwd_args g) {
  __shared__ __attribute__((aligned(16))) unsigned char Xs[EB_X_BYTES];       // one-hot  [table row][32 batch rows]
  __shared__ __attribute__((aligned(16))) unsigned char Ys[3 * EB_Y_BYTES];   // gradient [piece][channel][32 batch rows]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int c0 = blockIdx.y * 128;
  const int64_t r_begin = (int64_t)blockIdx.x * g.rows_per_block;
  int64_t r_end = r_begin + g.rows_per_block;
  if (r_end > g.N) r_end = g.N;
  if (r_begin >= r_end) return;

  for (int i = tid; i < EB_X_BYTES / 16; i += 256) reinterpret_cast<f32x4*>(Xs)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();  // (the first step's ones are set by other threads than the ones that zeroed their words)

  f32x16 acc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // gradient loader: thread = (channel, 16 batch rows): 4-byte loads, 256 contiguous bytes per wave instruction
  const int yc = tid & 127, yh = tid >> 7;
  const bool y_ok = c0 + yc < g.H;
  // one-hot setter: work item = (batch row, feature): K <= 12 -> at most 384 items, two per thread at most
  const int nitems = 32 * g.K;
  int xpos[2] = {-1, -1};

  for (int64_t r0 = r_begin; r0 < r_end; r0 += 32) {
    float yv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int64_t row = r0 + 16 * yh + j;
      yv[j] = (y_ok && row < r_end) ? g.Y[row * g.ldy + c0 + yc] : 0.f;
    }
    int fi[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int it = tid + 256 * q;
      const int m = it / g.K, k = it - m * g.K;
      const int64_t row = r0 + m;
      fi[q] = -1;
      if (it < nitems && row < r_end) {
        const int f = (int)g.idx[row * g.K + k];
        const int span = g.offs[k + 1] - g.offs[k];
        if (f >= 0 && f < span) fi[q] = (g.offs[k] + f) * EB_LDX + m * 2;  // byte offset of X[table row][m]
      }
    }
    {
      bf16x8 pc[2][3];
      float x8[8];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x8[j] = yv[8 * hf + j];
        split3(x8, pc[hf][0], pc[hf][1], pc[hf][2]);
      }
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        unsigned char* q = Ys + p * EB_Y_BYTES + yc * EB_LDX + 32 * yh;
        *reinterpret_cast<bf16x8*>(q) = pc[0][p];
        *reinterpret_cast<bf16x8*>(q + 16) = pc[1][p];
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      xpos[q] = fi[q];
      if (fi[q] >= 0) *reinterpret_cast<unsigned short*>(Xs + fi[q]) = 0x3F80;  // bf16 1.0
    }
    __syncthreads();
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      bf16x8 b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
        b[p] = *reinterpret_cast<const bf16x8*>(Ys + p * EB_Y_BYTES + (wave * 32 + li) * EB_LDX + 32 * sl + 16 * lh);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(Xs + (i * 32 + li) * EB_LDX + 32 * sl + 16 * lh);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[2], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[1], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[0], acc[i], 0, 0, 0);
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (xpos[q] >= 0) *reinterpret_cast<unsigned short*>(Xs + xpos[q]) = 0;  // (same thread that set it)
  }
  const int gc = c0 + wave * 32 + li;
  if (gc < g.H) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gr = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float v = acc[i][r];
        if (gr < g.R && v != 0.f) atomicAdd(g.dT + (int64_t)gr * g.H + gc, v);
      }
  }
}

// returns GNX_OK after launching, or 1 if the shape is not eligible (caller falls back to the LDS-atomic kernel)
int32_t gnx_embed_bwd_mfma(gnx_handle* h, const int64_t* idx, int64_t N, int K, const int32_t* offsets, int R,
                           const float* dout, int H, float* dtable) {
  if (K > EMB_MAX_K || (H % 4) != 0 || !aligned16(dout) || N < 256) return 1;  // (below 4096 rows the LDS-atomic kernel
  // is contention-bound: 76 us for a 32-graph batch's 640 atoms, whose feature values repeat in every row)
  if (h->opt[GNX_OPT_EMBED_BWD_MFMA] == 0) return 1;
  embed_bwd_args g;
  g.idx = idx;
  for (int k = 0; k <= EMB_MAX_K; ++k) g.offs[k] = offsets[k <= K ? k : K];
  g.K = K;
  g.Y = dout;
  g.ldy = H;
  g.N = N;
  g.R = R;
  g.H = H;
  g.dT = dtable;
  if (h->opt[GNX_OPT_EMBED_BWD_MFMA] == 1 && R <= EB_R) {  // (2 = the fp32-MFMA kernel below)
    // ~256 workgroups; at least 128 rows each; small batches: ONE chunk per channel tile (one adder per element: deterministic)
    int64_t rows = gnx_cdiv(gnx_cdiv(N, (int64_t)(N < 4096 ? 1 : 256)), 32) * 32;
    if (rows < 128) rows = 128;
    g.rows_per_block = rows;
    dim3 grid((unsigned)gnx_cdiv(N, rows), (unsigned)gnx_cdiv(H, 128));
    hipLaunchKernelGGL(k_embed_bwd_bf16, grid, dim3(256), 0, h->stream, g);
    GNX_LAUNCH_CHECK();
    return GNX_OK;
  }
  const int64_t tiles = gnx_cdiv(R, BN) * gnx_cdiv(H, BN);
  int64_t chunks = gnx_cdiv(512, tiles);
  int64_t rows = gnx_cdiv(gnx_cdiv(N, chunks), BK) * BK;
  if (rows < 128) rows = 128;
  if (N < 4096) rows = gnx_cdiv(N, (int64_t)BK) * BK;  // small batches: ONE row chunk per output tile, i.e. one adder per
                                                       // table element -> a deterministic sum (and 20 K-steps at most)
  g.rows_per_block = rows;
  dim3 grid((unsigned)gnx_cdiv(N, rows), (unsigned)gnx_cdiv(R, BN), (unsigned)gnx_cdiv(H, BN));
  hipLaunchKernelGGL(k_embed_bwd_mfma, grid, dim3(256), 0, h->stream, g);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

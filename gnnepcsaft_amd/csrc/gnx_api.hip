// Handle lifecycle, error text, profiling hook, tiny elementwise helpers.
#include "gnx_common.hpp"

#include <cstdlib>
#include <cstring>

static thread_local char g_err[512] = "";

void gnx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* gnx_last_error(void) { return g_err; }
extern "C" int32_t gnx_abi_version(void) { return GNX_ABI_VERSION; }

extern "C" int32_t gnx_create(gnx_handle** out, int32_t device) {
  GNX_CHECK_ARG(out != nullptr, "gnx_create: out is NULL");
  int count = 0;
  GNX_HIP(hipGetDeviceCount(&count));
  GNX_CHECK_ARG(device >= 0 && device < count, "gnx_create: device %d not in [0,%d)", device, count);
  GNX_HIP(hipSetDevice(device));
  gnx_handle* h = new gnx_handle();
  h->device = device;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->num_cus = prop.multiProcessorCount;
  }
  hipError_t e = hipMalloc(&h->d_flag, 256);
  if (e == hipSuccess) e = hipMalloc(&h->d_scratch, 4096);
  if (e == hipSuccess) e = hipMemset(h->d_flag, 0, 256);
  if (e == hipSuccess) e = hipMalloc(&h->d_zero, 256);
  if (e == hipSuccess) e = hipMemset(h->d_zero, 0, 256);
  if (e != hipSuccess) {
    gnx_set_error("gnx_create: %s", hipGetErrorString(e));
    delete h;
    return GNX_E_HIP;
  }
  // option defaults, overridable once by the environment (read here, never per launch) and later by gnx_set_option
  static const struct {
    int id;
    const char* env;
    int def;
  } opts[] = {{GNX_OPT_GEMM_SPLIT, "GNX_GEMM_SPLIT", 1},       {GNX_OPT_GEMM_WS, "GNX_GEMM_WS", 1},
              {GNX_OPT_GEMM_VEC, "GNX_GEMM_VEC", 1},           {GNX_OPT_WGRAD_VEC, "GNX_WGRAD_VEC", 1},
              {GNX_OPT_WGRAD_WGS, "GNX_WGRAD_WGS", 0},         {GNX_OPT_AGG_BWD_RECOMPUTE, "GNX_AGG_BWD_RECOMPUTE", 1},
              {GNX_OPT_EMBED_BWD_MFMA, "GNX_EMBED_BWD_MFMA", 1}, {GNX_OPT_STD_BWD_CENTERED, "GNX_STD_BWD_CENTERED", 1},
              {GNX_OPT_GEMM_PIPE, "GNX_GEMM_PIPE", 1},
              {GNX_OPT_WGRAD_PIPE, "GNX_WGRAD_PIPE", 1},       {GNX_OPT_EDGE_FUSED, "GNX_EDGE_FUSED", 1},
              {GNX_OPT_SIDE_CUS, "GNX_SIDE_CUS", 0},           {GNX_OPT_GEMM_AS, "GNX_GEMM_AS", 1},
              {GNX_OPT_GEMM_WS_FAST, "GNX_GEMM_WS_FAST", 2},   {GNX_OPT_GEMM_TILE_ROWS, "GNX_GEMM_TILE_ROWS", 0},
              {GNX_OPT_GEMM_MID, "GNX_GEMM_MID", 1},           {GNX_OPT_SPLIT_AHEAD, "GNX_SPLIT_AHEAD", 1}};
  for (const auto& o : opts) {
    const char* e = getenv(o.env);
    h->opt[o.id] = e ? atoi(e) : o.def;
  }
  *out = h;
  return GNX_OK;
}

extern "C" int32_t gnx_set_option(gnx_handle* h, int32_t opt, int32_t value) {
  GNX_CHECK_ARG(h != nullptr && opt >= 0 && opt < GNX_OPT_COUNT, "gnx_set_option: bad handle or option %d", opt);
  h->opt[opt] = value;
  return GNX_OK;
}

extern "C" int32_t gnx_get_option(gnx_handle* h, int32_t opt, int32_t* value) {
  GNX_CHECK_ARG(h != nullptr && value != nullptr && opt >= 0 && opt < GNX_OPT_COUNT, "gnx_get_option: bad argument");
  *value = h->opt[opt];
  return GNX_OK;
}

extern "C" int32_t gnx_destroy(gnx_handle* h) {
  if (!h) return GNX_OK;
  for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
  (void)hipFree(h->d_flag);
  (void)hipFree(h->d_scratch);
  (void)hipFree(h->d_zero);
  for (int i = 0; i < gnx_handle::kSideStreams; ++i) {
    if (h->side_fork[i]) (void)hipEventDestroy(h->side_fork[i]);
    if (h->side_done[i]) (void)hipEventDestroy(h->side_done[i]);
    if (h->side[i]) (void)hipStreamDestroy(h->side[i]);
  }
  delete h;
  return GNX_OK;
}

extern "C" int32_t gnx_set_stream(gnx_handle* h, void* hip_stream) {
  GNX_CHECK_ARG(h != nullptr, "gnx_set_stream: handle is NULL");
  GNX_CHECK_ARG(!h->on_side, "gnx_set_stream: between gnx_side_begin and gnx_side_end");
  h->stream = reinterpret_cast<hipStream_t>(hip_stream);
  return GNX_OK;
}

static int32_t side_ensure(gnx_handle* h, int which) {
  GNX_CHECK_ARG(h != nullptr, "gnx_side_*: handle is NULL");
  GNX_CHECK_ARG(which >= 0 && which < gnx_handle::kSideStreams, "gnx_side_*: stream %d not in [0,%d)", which,
                gnx_handle::kSideStreams);
  if (h->side[which] == nullptr) {
    GNX_HIP(hipSetDevice(h->device));
    // GNX_OPT_SIDE_CUS (A/B): side stream 0 (weight gradients) confined to that many CUs, spread evenly
    // over the chip, so that the memory-bound kernels of the main stream keep the rest to themselves
    const int want = which == 0 ? h->opt[GNX_OPT_SIDE_CUS] : 0;
    const int ncu = h->num_cus > 0 ? h->num_cus : 256;
    if (want > 0 && want < ncu) {
      std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
      for (int k = 0; k < want; ++k) {
        const int cu = (int)(((int64_t)k * ncu) / want);
        mask[cu >> 5] |= 1u << (cu & 31);
      }
      GNX_HIP(hipExtStreamCreateWithCUMask(&h->side[which], (uint32_t)mask.size(), mask.data()));
    } else {
      GNX_HIP(hipStreamCreateWithFlags(&h->side[which], hipStreamNonBlocking));
    }
    GNX_HIP(hipEventCreateWithFlags(&h->side_fork[which], hipEventDisableTiming));
    GNX_HIP(hipEventCreateWithFlags(&h->side_done[which], hipEventDisableTiming));
  }
  return GNX_OK;
}

extern "C" int32_t gnx_side_begin_n(gnx_handle* h, int32_t which) {
  const int32_t st = side_ensure(h, which);
  if (st != GNX_OK) return st;
  GNX_CHECK_ARG(!h->on_side, "gnx_side_begin: already on a side stream");
  GNX_HIP(hipEventRecord(h->side_fork[which], h->stream));
  GNX_HIP(hipStreamWaitEvent(h->side[which], h->side_fork[which], 0));
  h->main_saved = h->stream;
  h->stream = h->side[which];
  h->on_side = true;
  return GNX_OK;
}

extern "C" int32_t gnx_side_begin(gnx_handle* h) { return gnx_side_begin_n(h, 0); }

extern "C" int32_t gnx_side_stream_n(gnx_handle* h, int32_t which, void** hip_stream) {
  GNX_CHECK_ARG(hip_stream != nullptr, "gnx_side_stream: NULL argument");
  const int32_t st = side_ensure(h, which);
  if (st != GNX_OK) return st;
  *hip_stream = reinterpret_cast<void*>(h->side[which]);
  return GNX_OK;
}

extern "C" int32_t gnx_side_stream(gnx_handle* h, void** hip_stream) { return gnx_side_stream_n(h, 0, hip_stream); }

extern "C" int32_t gnx_side_end(gnx_handle* h) {
  GNX_CHECK_ARG(h != nullptr, "gnx_side_end: handle is NULL");
  if (h->on_side) {
    h->stream = h->main_saved;
    h->on_side = false;
  }
  return GNX_OK;
}

extern "C" int32_t gnx_side_join_n(gnx_handle* h, int32_t which) {
  GNX_CHECK_ARG(h != nullptr, "gnx_side_join: handle is NULL");
  GNX_CHECK_ARG(which >= 0 && which < gnx_handle::kSideStreams, "gnx_side_join: bad stream %d", which);
  GNX_CHECK_ARG(!h->on_side, "gnx_side_join: between gnx_side_begin and gnx_side_end");
  if (h->side[which] == nullptr) return GNX_OK;
  GNX_HIP(hipEventRecord(h->side_done[which], h->side[which]));
  GNX_HIP(hipStreamWaitEvent(h->stream, h->side_done[which], 0));
  return GNX_OK;
}

extern "C" int32_t gnx_side_join(gnx_handle* h) { return gnx_side_join_n(h, 0); }

int32_t gnx_read_flag(gnx_handle* h, int* value) {
  int v = 0;
  GNX_HIP(hipMemcpyAsync(&v, h->d_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  GNX_HIP(hipStreamSynchronize(h->stream));
  if (v != 0) GNX_HIP(hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream));
  *value = v;
  return GNX_OK;
}

extern "C" int32_t gnx_check_range(gnx_handle* h) {
  GNX_CHECK_ARG(h != nullptr, "gnx_check_range: handle is NULL");
  int v = 0;
  int32_t st = gnx_read_flag(h, &v);
  if (st != GNX_OK) return st;
  if (v != 0) {
    gnx_set_error("integer input out of range:%s%s%s%s%s%s", (v & 32) ? " an in-degree (or bond code) exceeds the caller's hint;" : "",
                  (v & 1) ? " edge_index holds a node id outside [0,N);" : "",
                  (v & 2) ? " edge/node feature outside its vocabulary;" : "",
                  (v & 4) ? " batch holds a graph id outside [0,B);" : "", (v & 8) ? " batch is not sorted;" : "",
                  (v & 16) ? " embedding index outside its table;" : "");
    return GNX_E_RANGE;
  }
  return GNX_OK;
}

extern "C" int32_t gnx_prof_begin(gnx_handle* h, uint32_t kernel_mask) {
  GNX_CHECK_ARG(h != nullptr, "gnx_prof_begin: handle is NULL");
  GNX_CHECK_ARG((kernel_mask >> GNX_K_COUNT) == 0, "gnx_prof_begin: bad kernel mask 0x%x", kernel_mask);
  h->prof_mask = kernel_mask & ~1u;
  h->ev_used = 0;
  return GNX_OK;
}

extern "C" int32_t gnx_prof_read(gnx_handle* h, int32_t kid, int64_t* launches, double* total_ms, double* alg_bytes,
                                 double* alg_flops, double* mfma_bf16_flops) {
  GNX_CHECK_ARG(h != nullptr && launches != nullptr && total_ms != nullptr, "gnx_prof_read: NULL argument");
  GNX_CHECK_ARG(kid > 0 && kid < GNX_K_COUNT, "gnx_prof_read: bad kernel id %d", kid);
  GNX_HIP(hipStreamSynchronize(h->stream));
  double tot = 0.0, w[3] = {0.0, 0.0, 0.0};
  int64_t n = 0;
  for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
    if (h->ev_kid[i / 2] != kid) continue;
    float ms = 0.f;
    GNX_HIP(hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
    tot += ms;
    for (int q = 0; q < 3; ++q) w[q] += h->ev_work[3 * (i / 2) + q];
    ++n;
  }
  *launches = n;
  *total_ms = tot;
  if (alg_bytes) *alg_bytes = w[0];
  if (alg_flops) *alg_flops = w[1];
  if (mfma_bf16_flops) *mfma_bf16_flops = w[2];
  return GNX_OK;
}

extern "C" int32_t gnx_prof_end(gnx_handle* h) {
  GNX_CHECK_ARG(h != nullptr, "gnx_prof_end: handle is NULL");
  h->prof_mask = 0;
  h->ev_used = 0;
  return GNX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
__global__ void k_fill(float* __restrict__ p, int64_t n, float v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

extern "C" int32_t gnx_fill(gnx_handle* h, float* p, int64_t n, float v) {
  GNX_CHECK_ARG(h && (p || n == 0) && n >= 0, "gnx_fill: bad argument");
  if (n == 0) return GNX_OK;
  int blocks = (int)(gnx_cdiv(n, 256) < 2048 ? gnx_cdiv(n, 256) : 2048);
  hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, h->stream, p, n, v);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

__global__ void k_scale(float* __restrict__ p, int64_t n, float v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] *= v;
}

extern "C" int32_t gnx_scale(gnx_handle* h, float* p, int64_t n, float v) {
  GNX_CHECK_ARG(h && (p || n == 0) && n >= 0, "gnx_scale: bad argument");
  if (n == 0) return GNX_OK;
  int blocks = (int)(gnx_cdiv(n, 256) < 2048 ? gnx_cdiv(n, 256) : 2048);
  hipLaunchKernelGGL(k_scale, dim3(blocks), dim3(256), 0, h->stream, p, n, v);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Dropout (reference: torch.nn.Dropout(p) in front of every conv, train/models.py:177,209; p = 0.25 in
// configs/pna_msigmae_7.py:40).  Counter-based Philox4x32-10: element i of a call is decided by the 128-bit counter
// (i / 4, offset) under the 64-bit key `seed` -- no state, so the backward pass RECOMPUTES the mask from
// (seed, offset) instead of storing it, and forward / backward agree by construction.  One thread = 4 consecutive
// elements = one Philox block.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0;
    c[1] = lo1;
    c[2] = n2;
    c[3] = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

__global__ void __launch_bounds__(256) k_dropout(const float* __restrict__ x, int64_t n, float p, float scale,
                                                 uint64_t seed, uint64_t offset, float* __restrict__ y) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // block of 4 elements
  const int64_t i0 = q * 4;
  if (i0 >= n) return;
  uint32_t c[4] = {(uint32_t)q, (uint32_t)((uint64_t)q >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  float v[4];
  if (i0 + 4 <= n && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + i0);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = ((float)(c[j] >> 8) * 5.9604644775390625e-8f >= p) ? v[j] * scale : 0.f;
    *reinterpret_cast<f32x4*>(y + i0) = f32x4{v[0], v[1], v[2], v[3]};
  } else {
    for (int j = 0; j < 4 && i0 + j < n; ++j)
      y[i0 + j] = ((float)(c[j] >> 8) * 5.9604644775390625e-8f >= p) ? x[i0 + j] * scale : 0.f;
  }
}

extern "C" int32_t gnx_dropout(gnx_handle* h, const float* x, int64_t n, float p, uint64_t seed, uint64_t offset,
                               float* y) {
  GNX_CHECK_ARG(h && ((x && y) || n == 0) && n >= 0, "gnx_dropout: bad argument");
  GNX_CHECK_ARG(p >= 0.f && p < 1.f, "gnx_dropout: p=%g not in [0,1)", (double)p);
  if (n == 0) return GNX_OK;
  hipLaunchKernelGGL(k_dropout, dim3((unsigned)gnx_cdiv(gnx_cdiv(n, 4), 256)), dim3(256), 0, h->stream, x, n, p,
                     1.0f / (1.0f - p), seed, offset, y);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

__global__ void k_axpy(float* __restrict__ y, const float* __restrict__ x, int64_t n, float a) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = fmaf(a, x[i], y[i]);
}

extern "C" int32_t gnx_axpy(gnx_handle* h, float* y, const float* x, int64_t n, float a) {
  GNX_CHECK_ARG(h && ((y && x) || n == 0) && n >= 0, "gnx_axpy: bad argument");
  if (n == 0) return GNX_OK;
  int blocks = (int)(gnx_cdiv(n, 256) < 2048 ? gnx_cdiv(n, 256) : 2048);
  hipLaunchKernelGGL(k_axpy, dim3(blocks), dim3(256), 0, h->stream, y, x, n, a);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

__global__ void k_clip_rows(const float* __restrict__ x, int64_t M, int P, const float* __restrict__ lo,
                            const float* __restrict__ hi, float* __restrict__ y) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * P) return;
  int c = (int)(i % P);
  const float v = x[i];
  y[i] = (v != v) ? v : fminf(fmaxf(v, lo[c]), hi[c]);  // Tensor.clip propagates NaN (fminf/fmaxf would not)
}

extern "C" int32_t gnx_clip_rows(gnx_handle* h, const float* x, int64_t M, int32_t P, const float* lo,
                                 const float* hi, float* y) {
  GNX_CHECK_ARG(h && x && lo && hi && y && M >= 0 && P > 0, "gnx_clip_rows: bad argument");
  if (M == 0) return GNX_OK;
  hipLaunchKernelGGL(k_clip_rows, dim3((unsigned)gnx_cdiv(M * P, 256)), dim3(256), 0, h->stream, x, M, (int)P, lo, hi,
                     y);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

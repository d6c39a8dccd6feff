// BatchNorm1d (+ReLU) over the rows of a [M, H] activation (PyG BatchNorm wraps torch.nn.BatchNorm1d; the readout MLP
// uses BatchNorm1d directly), and the APE-Huber loss.  HBM-bound column reductions + elementwise passes.
//
// Statistics: single pass of SHIFTED sums (pivot = row 0 of every column) so that var = E[(x-k)^2] - E[x-k]^2 does not
// cancel catastrophically; per-chunk partials are written to the workspace and folded by one thread per column in chunk
// order -> deterministic, no atomics.
#include "gnx_common.hpp"

#define BN_ROWS_PER_BLOCK 256

// workspace layout (floats): [2*H] alpha/beta' | [nchunk * 2 * H] partials
extern "C" size_t gnx_batchnorm_workspace_bytes(int64_t M, int32_t H) {
  if (M < 0) M = 0;
  size_t chunks = (size_t)gnx_cdiv(M > 0 ? M : 1, BN_ROWS_PER_BLOCK);
  return sizeof(float) * ((size_t)4 * H + chunks * 2 * (size_t)H) + 256;
}

// grid = (chunks, column blocks of 64); block = 64 columns x 4 row lanes
__global__ void __launch_bounds__(256) k_bn_partial(const float* __restrict__ x, int64_t M, int H,
                                                    float* __restrict__ part) {
  __shared__ float s1[4][64], s2[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + cx;
  int64_t r0 = (int64_t)blockIdx.x * BN_ROWS_PER_BLOCK;
  int64_t r1 = r0 + BN_ROWS_PER_BLOCK;
  if (r1 > M) r1 = M;
  float a1 = 0.f, a2 = 0.f;
  if (col < H) {
    const float k = x[col];
    for (int64_t r = r0 + ry; r < r1; r += 4) {
      float v = x[r * H + col] - k;
      a1 += v;
      a2 += v * v;
    }
  }
  s1[ry][cx] = a1;
  s2[ry][cx] = a2;
  __syncthreads();
  if (ry == 0 && col < H) {
    float t1 = (s1[0][cx] + s1[1][cx]) + (s1[2][cx] + s1[3][cx]);
    float t2 = (s2[0][cx] + s2[1][cx]) + (s2[2][cx] + s2[3][cx]);
    part[((int64_t)blockIdx.x * 2 + 0) * H + col] = t1;
    part[((int64_t)blockIdx.x * 2 + 1) * H + col] = t2;
  }
}


// float4 variants (H % 4 == 0): block = 32 column-quads x 8 row lanes; grid.y covers 128 columns per block.
__global__ void __launch_bounds__(256) k_bn_partial_v4(const float* __restrict__ x, int64_t M, int H,
                                                       float* __restrict__ part) {
  __shared__ f32x4 s1[8][32], s2[8][32];
  const int cq = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int col = blockIdx.y * 128 + cq * 4;
  int64_t r0 = (int64_t)blockIdx.x * BN_ROWS_PER_BLOCK;
  int64_t r1 = r0 + BN_ROWS_PER_BLOCK;
  if (r1 > M) r1 = M;
  f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
  if (col < H) {
    const f32x4 k = *reinterpret_cast<const f32x4*>(x + col);
    int64_t r = r0 + ry;
    for (; r + 24 < r1; r += 32) {  // four rows in flight (same order of additions as the one-row loop below)
      f32x4 v0 = *reinterpret_cast<const f32x4*>(x + r * H + col);
      f32x4 v1 = *reinterpret_cast<const f32x4*>(x + (r + 8) * H + col);
      f32x4 v2 = *reinterpret_cast<const f32x4*>(x + (r + 16) * H + col);
      f32x4 v3 = *reinterpret_cast<const f32x4*>(x + (r + 24) * H + col);
      v0 -= k;
      a1 += v0;
      a2 += v0 * v0;
      v1 -= k;
      a1 += v1;
      a2 += v1 * v1;
      v2 -= k;
      a1 += v2;
      a2 += v2 * v2;
      v3 -= k;
      a1 += v3;
      a2 += v3 * v3;
    }
    for (; r < r1; r += 8) {
      f32x4 v = *reinterpret_cast<const f32x4*>(x + r * H + col);
      v -= k;
      a1 += v;
      a2 += v * v;
    }
  }
  s1[ry][cq] = a1;
  s2[ry][cq] = a2;
  __syncthreads();
  if (ry == 0 && col < H) {
    f32x4 t1 = s1[0][cq], t2 = s2[0][cq];
#pragma unroll
    for (int l = 1; l < 8; ++l) {
      t1 += s1[l][cq];
      t2 += s2[l][cq];
    }
    *reinterpret_cast<f32x4*>(part + ((int64_t)blockIdx.x * 2 + 0) * H + col) = t1;
    *reinterpret_cast<f32x4*>(part + ((int64_t)blockIdx.x * 2 + 1) * H + col) = t2;
  }
}

__global__ void __launch_bounds__(256) k_bn_bwd_partial_v4(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ y, int64_t M, int H,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, int relu,
                                                           float* __restrict__ part) {
  __shared__ f32x4 s1[8][32], s2[8][32];
  const int cq = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int col = blockIdx.y * 128 + cq * 4;
  int64_t r0 = (int64_t)blockIdx.x * BN_ROWS_PER_BLOCK;
  int64_t r1 = r0 + BN_ROWS_PER_BLOCK;
  if (r1 > M) r1 = M;
  f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
  if (col < H) {
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + col);
    const f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + col);
    auto row = [&](int64_t r, f32x4& g, f32x4& xv, f32x4& yv) {
      g = *reinterpret_cast<const f32x4*>(dy + r * H + col);
      xv = *reinterpret_cast<const f32x4*>(x + r * H + col);
      yv = *reinterpret_cast<const f32x4*>((relu ? y : x) + r * H + col);  // (unconditional: keeps the loads countable)
    };
    auto fold = [&](f32x4 g, const f32x4& xv, const f32x4& yv) {
      if (relu) {
        g.x = yv.x > 0.f ? g.x : 0.f;
        g.y = yv.y > 0.f ? g.y : 0.f;
        g.z = yv.z > 0.f ? g.z : 0.f;
        g.w = yv.w > 0.f ? g.w : 0.f;
      }
      a1 += g;
      a2 += g * ((xv - mu) * rs);
    };
    int64_t r = r0 + ry;
    for (; r + 8 < r1; r += 16) {  // two rows (six loads) in flight; same order of additions as the one-row loop below
      f32x4 g0, x0, y0, g1, x1, y1;
      row(r, g0, x0, y0);
      row(r + 8, g1, x1, y1);
      fold(g0, x0, y0);
      fold(g1, x1, y1);
    }
    for (; r < r1; r += 8) {
      f32x4 g, xv, yv;
      row(r, g, xv, yv);
      fold(g, xv, yv);
    }
  }
  s1[ry][cq] = a1;
  s2[ry][cq] = a2;
  __syncthreads();
  if (ry == 0 && col < H) {
    f32x4 t1 = s1[0][cq], t2 = s2[0][cq];
#pragma unroll
    for (int l = 1; l < 8; ++l) {
      t1 += s1[l][cq];
      t2 += s2[l][cq];
    }
    *reinterpret_cast<f32x4*>(part + ((int64_t)blockIdx.x * 2 + 0) * H + col) = t1;
    *reinterpret_cast<f32x4*>(part + ((int64_t)blockIdx.x * 2 + 1) * H + col) = t2;
  }
}

template <int VEC>
__global__ void __launch_bounds__(256) k_bn_bwd_apply_v(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ y, int64_t total, int H,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ coef, int relu,
                                                        float* __restrict__ dx) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (; i < total; i += stride) {
    const int c = (int)(i % H);
    f32x4 g = *reinterpret_cast<const f32x4*>(dy + i);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + i);
    if (relu) {
      const f32x4 yv = *reinterpret_cast<const f32x4*>(y + i);
      g.x = yv.x > 0.f ? g.x : 0.f;
      g.y = yv.y > 0.f ? g.y : 0.f;
      g.z = yv.z > 0.f ? g.z : 0.f;
      g.w = yv.w > 0.f ? g.w : 0.f;
    }
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
    const f32x4 k0 = *reinterpret_cast<const f32x4*>(coef + c), k1 = *reinterpret_cast<const f32x4*>(coef + H + c),
                k2 = *reinterpret_cast<const f32x4*>(coef + 2 * H + c);
    const f32x4 xh = (xv - mu) * rs;
    *reinterpret_cast<f32x4*>(dx + i) = k0 * (g - k1 - xh * k2);
  }
}

// Fold the per-chunk partials of 16 columns with 16 lanes per column (lane l takes chunks l, l+16, ... in order, then
// the 16 lane sums are added in lane order): deterministic, ~chunks/16 dependent loads instead of `chunks`.
// blockDim = 256 = 16 chunk-lanes (y) x 16 columns (x); returns the totals to the y == 0 threads.
__device__ __forceinline__ void bn_fold(const float* __restrict__ part, int64_t chunks, int H, int c, float& t1,
                                        float& t2) {
  __shared__ float f1[16][17], f2[16][17];
  const int cx = threadIdx.x & 15, ly = threadIdx.x >> 4;
  float a1 = 0.f, a2 = 0.f;
  if (c < H) {
    // four chunks' loads in flight per step (same order of additions): the one-at-a-time loop was ~20 dependent L2 round
    // trips = 8-9 us for a kernel with no work, 16 times per step
    int64_t b = ly;
    for (; b + 48 < chunks; b += 64) {
      const float p0 = part[(b * 2 + 0) * H + c], q0 = part[(b * 2 + 1) * H + c];
      const float p1 = part[((b + 16) * 2 + 0) * H + c], q1 = part[((b + 16) * 2 + 1) * H + c];
      const float p2 = part[((b + 32) * 2 + 0) * H + c], q2 = part[((b + 32) * 2 + 1) * H + c];
      const float p3 = part[((b + 48) * 2 + 0) * H + c], q3 = part[((b + 48) * 2 + 1) * H + c];
      a1 += p0;
      a2 += q0;
      a1 += p1;
      a2 += q1;
      a1 += p2;
      a2 += q2;
      a1 += p3;
      a2 += q3;
    }
    for (; b < chunks; b += 16) {
      a1 += part[(b * 2 + 0) * H + c];
      a2 += part[(b * 2 + 1) * H + c];
    }
  }
  f1[ly][cx] = a1;
  f2[ly][cx] = a2;
  __syncthreads();
  t1 = 0.f;
  t2 = 0.f;
  if (ly == 0) {
#pragma unroll
    for (int l = 0; l < 16; ++l) {
      t1 += f1[l][cx];
      t2 += f2[l][cx];
    }
  }
}

// 16 lanes per column fold the chunks; lane 0 derives mean/rstd, updates running stats, emits alpha/beta'
__global__ void k_bn_finalize(const float* __restrict__ x, const float* __restrict__ part, int64_t chunks, int64_t M,
                              int H, const float* __restrict__ gamma, const float* __restrict__ beta,
                              float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
                              float eps, int training, float* __restrict__ save_mean, float* __restrict__ save_rstd,
                              float* __restrict__ ab, int64_t* __restrict__ num_batches_tracked) {
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  if (training && num_batches_tracked != nullptr && blockIdx.x == 0 && threadIdx.x == 0) num_batches_tracked[0] += 1;
  float t1 = 0.f, t2 = 0.f;
  if (training) bn_fold(part, chunks, H, c, t1, t2);
  if (c >= H || (threadIdx.x >> 4) != 0) return;
  float mean, var;
  if (training) {
    const float inv = 1.0f / (float)M;
    const float d = t1 * inv;
    mean = x[c] + d;
    var = fmaxf(t2 * inv - d * d, 0.f);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    if (running_var) {
      float unb = (M > 1) ? var * ((float)M / (float)(M - 1)) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
    }
  } else {
    mean = running_mean[c];
    var = running_var[c];
  }
  float rstd = 1.0f / sqrtf(var + eps);
  if (save_mean) save_mean[c] = mean;
  if (save_rstd) save_rstd[c] = rstd;
  float a = rstd * (gamma ? gamma[c] : 1.f);
  ab[c] = a;
  ab[H + c] = (beta ? beta[c] : 0.f) - mean * a;
}

template <int VEC>
__global__ void __launch_bounds__(256) k_bn_apply(const float* __restrict__ x, int64_t total, int H,
                                                  const float* __restrict__ ab, int relu, float* __restrict__ y) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  int64_t stride = (int64_t)gridDim.x * blockDim.x * VEC;
  for (; i < total; i += stride) {
    int c = (int)(i % H);
    if constexpr (VEC == 4) {
      f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
      f32x4 a = *reinterpret_cast<const f32x4*>(ab + c);
      f32x4 b = *reinterpret_cast<const f32x4*>(ab + H + c);
      f32x4 o = {v.x * a.x + b.x, v.y * a.y + b.y, v.z * a.z + b.z, v.w * a.w + b.w};
      if (relu) {
        o.x = fmaxf(o.x, 0.f);
        o.y = fmaxf(o.y, 0.f);
        o.z = fmaxf(o.z, 0.f);
        o.w = fmaxf(o.w, 0.f);
      }
      *reinterpret_cast<f32x4*>(y + i) = o;
    } else {
      float o = x[i] * ab[c] + ab[H + c];
      y[i] = relu ? fmaxf(o, 0.f) : o;
    }
  }
}

extern "C" int32_t gnx_batchnorm_fwd(gnx_handle* h, const float* x, int64_t M, int32_t H, const float* gamma,
                                     const float* beta, float* running_mean, float* running_var,
                                     int64_t* num_batches_tracked, float momentum, float eps, int32_t training,
                                     int32_t relu, float* y, float* save_mean, float* save_rstd, void* ws,
                                     size_t ws_bytes) {
  GNX_CHECK_ARG(h && H > 0 && M >= 0, "gnx_batchnorm_fwd: bad argument");
  if (M == 0) return GNX_OK;
  GNX_CHECK_ARG(x && y && ws, "gnx_batchnorm_fwd: NULL argument");
  GNX_CHECK_ARG(training || (running_mean && running_var), "gnx_batchnorm_fwd: eval mode needs running stats");
  if (ws_bytes < gnx_batchnorm_workspace_bytes(M, H)) {
    gnx_set_error("gnx_batchnorm_fwd: workspace %zu < %zu", ws_bytes, gnx_batchnorm_workspace_bytes(M, H));
    return GNX_E_WORKSPACE;
  }
  float* ab = reinterpret_cast<float*>(ws);
  float* part = ab + 4 * (size_t)H;
  int64_t chunks = gnx_cdiv(M, BN_ROWS_PER_BLOCK);
  gnx_prof_scope prof(h, GNX_K_BN_FWD, (training ? 12.0 : 8.0) * M * H);  // statistics pass + read x, write y
  if (training) {
    if (H % 4 == 0)
      hipLaunchKernelGGL(k_bn_partial_v4, dim3((unsigned)chunks, (unsigned)gnx_cdiv(H, 128)), dim3(256), 0, h->stream, x,
                         M, (int)H, part);
    else
      hipLaunchKernelGGL(k_bn_partial, dim3((unsigned)chunks, (unsigned)gnx_cdiv(H, 64)), dim3(256), 0, h->stream, x, M,
                         (int)H, part);
    GNX_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_bn_finalize, dim3((unsigned)gnx_cdiv(H, 16)), dim3(256), 0, h->stream, x, part, chunks, M, (int)H,
                     gamma, beta, running_mean, running_var, momentum, eps, (int)training, save_mean, save_rstd, ab,
                     num_batches_tracked);
  GNX_LAUNCH_CHECK();
  int64_t total = M * H;
  if (H % 4 == 0) {
    int64_t blocks = gnx_cdiv(total / 4, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_bn_apply<4>, dim3((unsigned)blocks), dim3(256), 0, h->stream, x, total, (int)H, ab, (int)relu, y);
  } else {
    int64_t blocks = gnx_cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_bn_apply<1>, dim3((unsigned)blocks), dim3(256), 0, h->stream, x, total, (int)H, ab, (int)relu, y);
  }
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// backward: g = dy * (y>0 if relu); dbeta = sum g; dgamma = sum g*xhat; dx = gamma*rstd*(g - dbeta/M - xhat*dgamma/M)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_bn_bwd_partial(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ y, int64_t M, int H,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        int relu, float* __restrict__ part) {
  __shared__ float s1[4][64], s2[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + cx;
  int64_t r0 = (int64_t)blockIdx.x * BN_ROWS_PER_BLOCK;
  int64_t r1 = r0 + BN_ROWS_PER_BLOCK;
  if (r1 > M) r1 = M;
  float a1 = 0.f, a2 = 0.f;
  if (col < H) {
    const float mu = mean[col], rs = rstd[col];
    for (int64_t r = r0 + ry; r < r1; r += 4) {
      float g = dy[r * H + col];
      if (relu && !(y[r * H + col] > 0.f)) g = 0.f;
      a1 += g;
      a2 += g * ((x[r * H + col] - mu) * rs);
    }
  }
  s1[ry][cx] = a1;
  s2[ry][cx] = a2;
  __syncthreads();
  if (ry == 0 && col < H) {
    part[((int64_t)blockIdx.x * 2 + 0) * H + col] = (s1[0][cx] + s1[1][cx]) + (s1[2][cx] + s1[3][cx]);
    part[((int64_t)blockIdx.x * 2 + 1) * H + col] = (s2[0][cx] + s2[1][cx]) + (s2[2][cx] + s2[3][cx]);
  }
}

// coef[c] = (gamma*rstd, dbeta/M, dgamma/M * rstd... ) stored as 4 rows of H: k0 = gamma*rstd, k1 = dbeta/M, k2 = dgamma/M
__global__ void k_bn_bwd_finalize(const float* __restrict__ part, int64_t chunks, int64_t M, int H,
                                  const float* __restrict__ gamma, const float* __restrict__ rstd,
                                  float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef) {
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  float t1 = 0.f, t2 = 0.f;
  bn_fold(part, chunks, H, c, t1, t2);
  if (c >= H || (threadIdx.x >> 4) != 0) return;
  if (dbeta) dbeta[c] += t1;
  if (dgamma) dgamma[c] += t2;
  const float inv = 1.0f / (float)M;
  coef[c] = (gamma ? gamma[c] : 1.f) * rstd[c];
  coef[H + c] = t1 * inv;
  coef[2 * H + c] = t2 * inv;
}

__global__ void __launch_bounds__(256) k_bn_bwd_apply(const float* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ y, int64_t total, int H,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ coef, int relu, float* __restrict__ dx) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    int c = (int)(i % H);
    float g = dy[i];
    if (relu && !(y[i] > 0.f)) g = 0.f;
    float xh = (x[i] - mean[c]) * rstd[c];
    dx[i] = coef[c] * (g - coef[H + c] - xh * coef[2 * H + c]);
  }
}

extern "C" int32_t gnx_batchnorm_bwd(gnx_handle* h, const float* dy, const float* x, const float* y, int64_t M,
                                     int32_t H, const float* gamma, const float* save_mean, const float* save_rstd,
                                     int32_t relu, float* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes) {
  GNX_CHECK_ARG(h && H > 0 && M >= 0, "gnx_batchnorm_bwd: bad argument");
  if (M == 0) return GNX_OK;
  GNX_CHECK_ARG(dy && x && save_mean && save_rstd && dx && ws && (!relu || y), "gnx_batchnorm_bwd: NULL argument");
  if (ws_bytes < gnx_batchnorm_workspace_bytes(M, H)) {
    gnx_set_error("gnx_batchnorm_bwd: workspace %zu < %zu", ws_bytes, gnx_batchnorm_workspace_bytes(M, H));
    return GNX_E_WORKSPACE;
  }
  float* coef = reinterpret_cast<float*>(ws);
  float* part = coef + 4 * (size_t)H;
  int64_t chunks = gnx_cdiv(M, BN_ROWS_PER_BLOCK);
  gnx_prof_scope prof(h, GNX_K_BN_BWD, 28.0 * M * H);  // two passes over dy, x, y + write dx
  if (H % 4 == 0)
    hipLaunchKernelGGL(k_bn_bwd_partial_v4, dim3((unsigned)chunks, (unsigned)gnx_cdiv(H, 128)), dim3(256), 0, h->stream,
                       dy, x, y, M, (int)H, save_mean, save_rstd, (int)relu, part);
  else
    hipLaunchKernelGGL(k_bn_bwd_partial, dim3((unsigned)chunks, (unsigned)gnx_cdiv(H, 64)), dim3(256), 0, h->stream, dy,
                       x, y, M, (int)H, save_mean, save_rstd, (int)relu, part);
  GNX_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((unsigned)gnx_cdiv(H, 16)), dim3(256), 0, h->stream, part, chunks, M, (int)H,
                     gamma, save_rstd, dgamma, dbeta, coef);
  GNX_LAUNCH_CHECK();
  int64_t total = M * H;
  if (H % 4 == 0) {
    int64_t blocks = gnx_cdiv(total / 4, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_bn_bwd_apply_v<4>, dim3((unsigned)blocks), dim3(256), 0, h->stream, dy, x, y, total, (int)H,
                       save_mean, save_rstd, coef, (int)relu, dx);
  } else {
    int64_t blocks = gnx_cdiv(total, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)blocks), dim3(256), 0, h->stream, dy, x, y, total, (int)H,
                       save_mean, save_rstd, coef, (int)relu, dx);
  }
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// APE-Huber loss + MAPE metric.  Stage 1: <=64 blocks write (huber, mape) partial sums into the handle scratch and
// dpred; stage 2: one wave folds them in block order.
// ---------------------------------------------------------------------------------------------------------------
#define LOSS_MAX_BLOCKS 64

__global__ void __launch_bounds__(256) k_huber_partial(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                       int64_t count, float delta, float* __restrict__ part,
                                                       float* __restrict__ dpred) {
  __shared__ float sh[256], sm[256];
  float ah = 0.f, am = 0.f;
  const float inv = 1.0f / (float)count;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) {
    float p = pred[i], t = tgt[i];
    float a = (p - t) / t;
    float aa = fabsf(a);
    ah += (aa <= delta) ? 0.5f * a * a : delta * (aa - 0.5f * delta);
    am += fabsf(p - t) / fmaxf(fabsf(t), 1.17e-06f);
    if (dpred) {
      float da = (aa <= delta) ? a : (a > 0.f ? delta : -delta);
      dpred[i] = da * inv / t;
    }
  }
  sh[threadIdx.x] = ah;
  sm[threadIdx.x] = am;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      sh[threadIdx.x] += sh[threadIdx.x + off];
      sm[threadIdx.x] += sm[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[blockIdx.x * 2] = sh[0];
    part[blockIdx.x * 2 + 1] = sm[0];
  }
}

__global__ void k_huber_final(const float* __restrict__ part, int blocks, int64_t count, float* __restrict__ out2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float h = 0.f, m = 0.f;
    for (int b = 0; b < blocks; ++b) {
      h += part[b * 2];
      m += part[b * 2 + 1];
    }
    out2[0] = h / (float)count;
    out2[1] = m / (float)count;
  }
}

extern "C" int32_t gnx_huber_ape(gnx_handle* h, const float* pred, const float* target, int64_t count, float delta,
                                 float* out2, float* dpred) {
  GNX_CHECK_ARG(h && pred && target && out2 && count > 0, "gnx_huber_ape: bad argument");
  int blocks = (int)(gnx_cdiv(count, 256) < LOSS_MAX_BLOCKS ? gnx_cdiv(count, 256) : LOSS_MAX_BLOCKS);
  hipLaunchKernelGGL(k_huber_partial, dim3(blocks), dim3(256), 0, h->stream, pred, target, count, delta, h->d_scratch,
                     dpred);
  GNX_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_huber_final, dim3(1), dim3(64), 0, h->stream, h->d_scratch, blocks, count, out2);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

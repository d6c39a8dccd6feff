// Optimizer step on device: AdamW(amsgrad=True, eps=1e-5, decoupled weight decay) and plain SGD as
// the reference configures them (/root/reference/gnnepcsaft/train/models.py:47-63), fused over ONE flat fp32 buffer
// (the same layout the gradient all-reduce uses): one HBM-bound elementwise pass instead of ~10 torch kernels per
// parameter tensor.  Arithmetic follows torch.optim.AdamW's single-tensor path step by step.
#include "gnx_common.hpp"

__global__ void __launch_bounds__(256) k_adamw_amsgrad(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v,
                                                       float* __restrict__ vmax, int64_t n, float lr, float beta1,
                                                       float beta2, float eps, float wd, float step_size,
                                                       float bc2_sqrt) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (; i < n; i += stride) {
    if (i + 3 < n) {
      f32x4 P = *reinterpret_cast<f32x4*>(p + i);
      const f32x4 G = *reinterpret_cast<const f32x4*>(g + i);
      f32x4 M = *reinterpret_cast<f32x4*>(m + i), V = *reinterpret_cast<f32x4*>(v + i),
            X = *reinterpret_cast<f32x4*>(vmax + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pp = P[k] * (1.0f - lr * wd);
        float mm = M[k] + (G[k] - M[k]) * (1.0f - beta1);
        float vv = V[k] * beta2 + (1.0f - beta2) * G[k] * G[k];
        float xx = fmaxf(X[k], vv);
        float denom = sqrtf(xx) / bc2_sqrt + eps;
        P[k] = pp - step_size * (mm / denom);
        M[k] = mm;
        V[k] = vv;
        X[k] = xx;
      }
      *reinterpret_cast<f32x4*>(p + i) = P;
      *reinterpret_cast<f32x4*>(m + i) = M;
      *reinterpret_cast<f32x4*>(v + i) = V;
      *reinterpret_cast<f32x4*>(vmax + i) = X;
    } else {
      for (int64_t j = i; j < n; ++j) {
        float pp = p[j] * (1.0f - lr * wd);
        float mm = m[j] + (g[j] - m[j]) * (1.0f - beta1);
        float vv = v[j] * beta2 + (1.0f - beta2) * g[j] * g[j];
        float xx = fmaxf(vmax[j], vv);
        float denom = sqrtf(xx) / bc2_sqrt + eps;
        p[j] = pp - step_size * (mm / denom);
        m[j] = mm;
        v[j] = vv;
        vmax[j] = xx;
      }
    }
  }
}

extern "C" int32_t gnx_adamw_amsgrad(gnx_handle* h, float* p, const float* g, float* m, float* v, float* vmax,
                                     int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                     int64_t step) {
  GNX_CHECK_ARG(h && n >= 0 && step >= 1, "gnx_adamw_amsgrad: bad argument");
  if (n == 0) return GNX_OK;
  GNX_CHECK_ARG(p && g && m && v && vmax, "gnx_adamw_amsgrad: NULL buffer");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  int64_t blocks = gnx_cdiv(gnx_cdiv(n, 4), 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_adamw_amsgrad, dim3((unsigned)blocks), dim3(256), 0, h->stream, p, g, m, v, vmax, n, lr, beta1,
                     beta2, eps, weight_decay, step_size, bc2_sqrt);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

__global__ void __launch_bounds__(256) k_sgd(float* __restrict__ p, const float* __restrict__ g, int64_t n, float lr) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = p[i] - lr * g[i];
}

extern "C" int32_t gnx_sgd(gnx_handle* h, float* p, const float* g, int64_t n, float lr) {
  GNX_CHECK_ARG(h && n >= 0 && (n == 0 || (p && g)), "gnx_sgd: bad argument");
  if (n == 0) return GNX_OK;
  int64_t blocks = gnx_cdiv(n, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_sgd, dim3((unsigned)blocks), dim3(256), 0, h->stream, p, g, n, lr);
  GNX_LAUNCH_CHECK();
  return GNX_OK;
}

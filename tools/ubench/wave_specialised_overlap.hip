// Micro-benchmark: do a MULTIPLY wave (48 bf16 MFMAs per step) and a STAGING wave (split of 32 floats into 3 x bf16 per
// step, optionally + 12 ds_write_b128) overlap when they are different waves of the same SIMD, joined by one barrier per
// step (the structure of k_gemm_wgrad3p)?  512 threads: waves 0..3 multiply, waves 4..7 stage.
// Build: hipcc --offload-arch=gfx950 -O3 -o wave_specialised_overlap wave_specialised_overlap.hip
// MI355X, round 2 (us per step, all CUs, random operands): barrier only 0.02; multiply waves only 0.91 (0.87 with fragment
// reads); staging waves only 0.60 (split) / 0.84 (+ LDS stores); both 1.04 / 1.41 / 1.46 (MFMA + split / + stores / + reads).
// With -fno-slp-vectorize (no packed-f32 VALU in the split): staging alone 0.60 / 1.02, both 1.04 / 1.24 / 1.30.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& p1, bf16x8& p2, bf16x8& p3) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h1 = (__bf16)x[j];
    const float r1 = x[j] - (float)h1;
    const __bf16 h2 = (__bf16)r1;
    const float r2 = r1 - (float)h2;
    p1[j] = h1; p2[j] = h2; p3[j] = (__bf16)r2;
  }
}

// MODE bit 0: multiply waves work, bit 1: staging waves split, bit 2: staging waves also store to LDS, bit 3: multiply waves
// also read their 24 fragments from LDS
template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[61440];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float sink = 0.f;
  if (wave < 4) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[6], b[6];
    unsigned hsh = 0x9E3779B9u * (tid + 1) + 0x85EBCA6Bu * (blockIdx.x + 1);
    for (int p = 0; p < 6; ++p) for (int j = 0; j < 8; ++j) {
      hsh = hsh * 1664525u + 1013904223u; a[p][j] = __builtin_bit_cast(__bf16, (unsigned short)(0x3F00u | ((hsh >> 9) & 0x80FFu)));
      hsh = hsh * 1664525u + 1013904223u; b[p][j] = __builtin_bit_cast(__bf16, (unsigned short)(0x3F00u | ((hsh >> 9) & 0x80FFu)));
    }
    for (int it = 0; it < iters; ++it) {
      if (MODE & 1) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          if (MODE & 8) {
#pragma unroll
            for (int p = 0; p < 6; ++p) {
              a[p] = *reinterpret_cast<const bf16x8*>(lds + (p % 3) * 10240 + ((wave >> 1) * 64 + (p / 3) * 32 + (lane & 31)) * 80 + 32 * sl + 16 * (lane >> 5));
              b[p] = *reinterpret_cast<const bf16x8*>(lds + 30720 + (p % 3) * 10240 + ((wave & 1) * 64 + (p / 3) * 32 + (lane & 31)) * 80 + 32 * sl + 16 * (lane >> 5));
            }
          }
#pragma unroll
          for (int i = 0; i < 24; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i % 6], b[(i / 6 + i) % 6], acc[i & 3], 0, 0, 0);
        }
      }
      __syncthreads();
    }
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) sink += acc[i][r];
  } else {
    float x[32];
    for (int j = 0; j < 32; ++j) x[j] = 1.25f * (j + 1) + tid;
    const int lt = tid - 256;
    for (int it = 0; it < iters; ++it) {
      if (MODE & 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float xx[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) xx[j] = x[8 * q + j];
          bf16x8 p1, p2, p3;
          split3(xx, p1, p2, p3);
          if (MODE & 4) {
            unsigned char* dst = lds + (lt >> 7) * 30720 + ((lt & 31) * 4 + q) * 80 + ((lt >> 5) & 3) * 16;
            *reinterpret_cast<bf16x8*>(dst) = p1;
            *reinterpret_cast<bf16x8*>(dst + 10240) = p2;
            *reinterpret_cast<bf16x8*>(dst + 20480) = p3;
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) x[8 * q + j] = x[8 * q + j] * 1.0009765625f + (float)p3[j];
        }
      }
      __syncthreads();
    }
    for (int j = 0; j < 32; ++j) sink += x[j];
  }
  out[blockIdx.x * 512 + tid] = sink;
}

template <int MODE>
static float run(int iters, float* d) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  const int it = 20000;
  printf("us per step, 256 CUs:  barrier only %.3f | multiply waves only %.3f (+ LDS fragment reads %.3f) | staging waves only: split %.3f, split + LDS stores %.3f\n",
         run<0>(it, d), run<1>(it, d), run<9>(it, d), run<2>(it, d), run<6>(it, d));
  printf("both groups: MFMA + split %.3f | MFMA + split + stores %.3f | MFMA + reads + split + stores %.3f\n", run<3>(it, d), run<7>(it, d), run<15>(it, d));
  return 0;
}

// Micro-benchmark: does the fp32 -> 3 x bf16 split (VALU) of the NEXT K-tile overlap with the bf16 MFMAs of the
// current one when both are issued by the SAME wave (one basic block), and what do two waves per SIMD buy?
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip ; run on the GPU box.
// MI355X, round 2 (long runs, one wave per SIMD): 48 MFMAs 0.65 us, + split of 32 floats 0.78 us (split alone 0.60 us);
// random operands on all 256 CUs: 0.84 -> 1.01 us; <= 128 CUs: 0.65 -> 0.72 us.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
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

// MODE 0: 48 MFMAs / iter.  1: split of 32 floats / iter (the per-thread share of a 128x32 A K-tile + nothing else).
// 2: both, source order MFMA then split (compiler schedules).  3: both, interleaved with sched_group_barrier.
template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters, float seed, int rnd) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a[3], b[3];
  for (int p = 0; p < 3; ++p) for (int j = 0; j < 8; ++j) { a[p][j] = (__bf16)(seed + p + j + threadIdx.x); b[p][j] = (__bf16)(seed * 0.5f + p - j); }
  if (rnd) {  // random significands and signs, magnitudes around 1 (what activations look like to the multipliers)
    unsigned hsh = 0x9E3779B9u * (threadIdx.x + 1) + 0x85EBCA6Bu * (blockIdx.x + 1);
    for (int p = 0; p < 3; ++p) for (int j = 0; j < 8; ++j) {
      hsh = hsh * 1664525u + 1013904223u; const unsigned short ua = 0x3F00u | ((hsh >> 9) & 0x80FFu);
      hsh = hsh * 1664525u + 1013904223u; const unsigned short ub = 0x3F00u | ((hsh >> 9) & 0x80FFu);
      a[p][j] = __builtin_bit_cast(__bf16, ua); b[p][j] = __builtin_bit_cast(__bf16, ub);
    }
  }
  float x[32];
  for (int j = 0; j < 32; ++j) x[j] = seed * (j + 1) + threadIdx.x;
  float sink = 0.f;
  for (int it = 0; it < iters; ++it) {
    bf16x8 pc[4][3];
    if (MODE == 0 || MODE == 2 || MODE == 3) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[t], 0, 0, 0);
        }
    }
    if (MODE >= 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float xx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xx[j] = x[8 * q + j];
        split3(xx, pc[q][0], pc[q][1], pc[q][2]);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[8 * q + j] = x[8 * q + j] * 1.0009765625f + (float)pc[q][2][j];
      }
    }
    if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 48; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);  // 7 VALU
      }
    }
  }
  for (int j = 0; j < 32; ++j) sink += x[j];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) sink += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
}

template <int MODE>
static float run(int threads, int iters, float* d, int rnd = 0, int blocks = 256) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.25f, rnd);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.25f, rnd);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  const int iters = 2000;
  for (int threads : {256, 512}) {
    const float t0 = run<0>(threads, iters, d), t1 = run<1>(threads, iters, d), t2 = run<2>(threads, iters, d), t3 = run<3>(threads, iters, d);
    printf("threads/CU %d (waves/SIMD %d): per iteration  mfma48 %.3f us  split32 %.3f us  both(compiler) %.3f us  both(interleaved) %.3f us\n",
           threads, threads / 256, t0 / iters, t1 / iters, t2 / iters, t3 / iters);
  }
  for (int blocks : {256, 128, 64, 16})
    for (int rnd : {0, 1}) {
      const int it2 = 20000;  // long enough (tens of ms) for the power management to settle
      const float t0 = run<0>(256, it2, d, rnd, blocks), t2 = run<2>(256, it2, d, rnd, blocks);
      printf("blocks %3d  %s operands: mfma48 %.3f us/iter   mfma48 + split32 %.3f us/iter\n", blocks, rnd ? "random" : "small-integer", t0 / it2, t2 / it2);
    }
  return 0;
}

// Micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate on all CUs as a function of operand activity.
//   same      every MFMA of the loop multiplies the same two register fragments (operands never toggle)
//   rotating  consecutive MFMAs take different fragments (6 A x 6 B random fragments, as in a real K-loop)
// with small-integer or random significands.  Long runs (tens of ms) so that the power management settles.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip ; run on the GPU box.
// MI355X, round 2:  256 CUs constant 0.666 / 0.667 us, random 0.843 / 0.874 us;  128 CUs 0.666-0.674 us in every case.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int ROT>
__global__ void __launch_bounds__(256) k(float* out, int iters, int rnd) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a[6], b[6];
  unsigned hsh = 0x9E3779B9u * (threadIdx.x + 1) + 0x85EBCA6Bu * (blockIdx.x + 1);
  for (int p = 0; p < 6; ++p) for (int j = 0; j < 8; ++j) {
    hsh = hsh * 1664525u + 1013904223u; const unsigned short ua = rnd ? (0x3F00u | ((hsh >> 9) & 0x80FFu)) : 0x3F80u;
    hsh = hsh * 1664525u + 1013904223u; const unsigned short ub = rnd ? (0x3F00u | ((hsh >> 9) & 0x80FFu)) : 0x4000u;
    a[p][j] = __builtin_bit_cast(__bf16, ua); b[p][j] = __builtin_bit_cast(__bf16, ub);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 48; ++i) {
      const int ia = ROT ? (i % 6) : 0, ib = ROT ? ((i / 6 + i) % 6) : 0;
      acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ia], b[ib], acc[i & 3], 0, 0, 0);
    }
  }
  float sink = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) sink += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
}

template <int ROT>
static float run(int blocks, int iters, float* d, int rnd) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<ROT>, dim3(blocks), dim3(256), 0, 0, d, iters, rnd);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<ROT>, dim3(blocks), dim3(256), 0, 0, d, iters, rnd);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 256 * 4);
  const int iters = 30000;
  for (int blocks : {256, 128})
    for (int rnd : {0, 1})
      printf("CUs %3d  %-13s operands: 48 MFMAs  same fragments %.3f us   rotating fragments %.3f us   (32 clk each at 2.4 GHz = 0.640 us)\n",
             blocks, rnd ? "random" : "constant", run<0>(blocks, iters, d, rnd), run<1>(blocks, iters, d, rnd));
  return 0;
}

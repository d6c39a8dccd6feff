// Diagnostic build of the weights-stationary split product (k_gemm_ws3<.., FAST>) with in-kernel phase stamps (wave 0 of
// every workgroup).  M x 128 x 128, NT weights, plain epilogue.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Iinclude -Ignnepcsaft_amd/csrc \
//              -o tools/ubench/gemm_ws3_stamp tools/ubench/gemm_ws3_stamp.hip        ; run on the GPU box.
#define WS_STAMP 1
#include "../../gnnepcsaft_amd/csrc/gnx_gemm.hip"

#include <cstdarg>
#include <cstdlib>
#include <vector>

void gnx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
extern "C" int32_t gnx_fill(gnx_handle*, float*, int64_t, float) { return GNX_OK; }

int main(int argc, char** argv) {
  const int K = 128, N = 128;
  gnx_handle h;
  h.num_cus = 256;
  for (int i = 0; i < GNX_OPT_COUNT; ++i) h.opt[i] = 0;
  h.opt[GNX_OPT_GEMM_SPLIT] = 1; h.opt[GNX_OPT_GEMM_VEC] = 1; h.opt[GNX_OPT_GEMM_WS] = 1; h.opt[GNX_OPT_GEMM_WS_FAST] = 1;
  for (int64_t M : {81920ll, 327680ll}) {
    std::vector<float> ha((size_t)M * K), hw((size_t)K * N);
    unsigned s = 777u;
    for (auto& v : ha) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) * 1e-3f; }
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) * 1e-4f; }
    float *A, *W, *C;
    (void)hipMalloc(&A, M * K * 4); (void)hipMalloc(&W, K * N * 4); (void)hipMalloc(&C, M * N * 4);
    (void)hipMemcpy(A, ha.data(), M * K * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(W, hw.data(), K * N * 4, hipMemcpyHostToDevice);
    gnx_gemm_seg seg;
    seg.a = A; seg.lda = K; seg.rowscale = nullptr; seg.b = W; seg.ldb = K; seg.k = K;
    unsigned long long* sb;
    (void)hipMalloc(&sb, (size_t)256 * 8 * 8);
    (void)hipMemset(sb, 0, (size_t)256 * 8 * 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(ws_stamp_buf), &sb, sizeof(sb));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i)
      if (gnx_gemm(&h, 1, &seg, M, N, nullptr, nullptr, 0, C, N, GNX_GEMM_B_TRANS, nullptr, 0) != GNX_OK) return 1;
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) (void)gnx_gemm(&h, 1, &seg, M, N, nullptr, nullptr, 0, C, N, GNX_GEMM_B_TRANS, nullptr, 0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> st((size_t)256 * 8);
    (void)hipMemcpy(st.data(), sb, st.size() * 8, hipMemcpyDeviceToHost);
    double sum[5] = {0}, tiles = 0;
    for (int b = 0; b < 256; ++b) {
      for (int i = 0; i < 5; ++i) sum[i] += (double)st[(size_t)b * 8 + i];
      tiles += (double)st[(size_t)b * 8 + 5];
    }
    double tot = 0;
    for (int i = 0; i < 5; ++i) tot += sum[i];
    printf("%lld rows: %.1f us per launch; wave 0 per tile: %.0f cycles = issue of the next tile's loads %.0f, fragment reads + 48 MFMAs %.0f, "
           "epilogue %.0f, load wait + split + LDS stores %.0f, barrier %.0f\n",
           (long long)M, ms * 1e3 / reps, tot / tiles, sum[0] / tiles, sum[1] / tiles, sum[2] / tiles, sum[3] / tiles, sum[4] / tiles);
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(C); (void)hipFree(sb);
  }
  return 0;
}

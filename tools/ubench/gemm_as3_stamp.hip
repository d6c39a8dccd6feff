// Diagnostic build of the activation-stationary split product (k_gemm_as3) with in-kernel phase stamps.
// dA shape of cfg-2: M = 81 920 rows, K = 128 -> N = 512, NN weights.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Iinclude -Ignnepcsaft_amd/csrc \
//              -o /tmp/gemm_as3_stamp tools/ubench/gemm_as3_stamp.hip        ; run on the GPU box.
#define AS_STAMP 1
#include "../../gnnepcsaft_amd/csrc/gnx_gemm.hip"

#include <algorithm>
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
  const int64_t M = argc > 1 ? atoll(argv[1]) : 81920;
  const int K = 128, N = 512;
  gnx_handle h;
  h.num_cus = 256;
  for (int i = 0; i < GNX_OPT_COUNT; ++i) h.opt[i] = 0;
  h.opt[GNX_OPT_GEMM_SPLIT] = 1; h.opt[GNX_OPT_GEMM_VEC] = 1; h.opt[GNX_OPT_GEMM_AS] = 1; h.opt[GNX_OPT_GEMM_PIPE] = 1;
  std::vector<float> ha((size_t)M * K), hw((size_t)K * N);
  unsigned s = 777u;
  for (auto& v : ha) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) * 1e-3f; }
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) * 1e-4f; }
  float *A, *W, *C;
  void* ws;
  (void)hipMalloc(&A, M * K * 4); (void)hipMalloc(&W, K * N * 4); (void)hipMalloc(&C, M * N * 4);
  (void)hipMemcpy(A, ha.data(), M * K * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(W, hw.data(), K * N * 4, hipMemcpyHostToDevice);
  gnx_gemm_seg seg;
  seg.a = A; seg.lda = K; seg.rowscale = nullptr; seg.b = W; seg.ldb = N; seg.k = K;
  const size_t wsb = gnx_gemm_workspace_bytes(&h, 1, &seg, nullptr, 1, M, N, nullptr, 0, 1);
  (void)hipMalloc(&ws, wsb ? wsb : 16);
  const int nwg = (int)((M + 63) / 64);
  unsigned long long* sb;
  (void)hipMalloc(&sb, (size_t)nwg * 8 * 8);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(as_stamp_buf), &sb, sizeof(sb));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int as = 1; as >= 0; --as) {
    h.opt[GNX_OPT_GEMM_AS] = as;
    for (int i = 0; i < 3; ++i)
      if (gnx_gemm(&h, 1, &seg, M, N, nullptr, nullptr, 0, C, N, 0, ws, wsb) != GNX_OK) return 1;
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) (void)gnx_gemm(&h, 1, &seg, M, N, nullptr, nullptr, 0, C, N, 0, ws, wsb);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("activation-stationary = %d: %.1f us per call (k_split_weights + product), %lld rows\n", as, ms * 1e3 / reps, (long long)M);
    if (!as) break;
    std::vector<unsigned long long> st((size_t)nwg * 8);
    (void)hipMemcpy(st.data(), sb, st.size() * 8, hipMemcpyDeviceToHost);
    double sum[5] = {0}, life = 0;
    unsigned long long t0 = ~0ull, t1 = 0;
    std::vector<double> starts;
    for (int b = 0; b < nwg; ++b) {
      for (int i = 0; i < 5; ++i) sum[i] += (double)st[(size_t)b * 8 + i];
      life += (double)(st[(size_t)b * 8 + 6] - st[(size_t)b * 8 + 5]);
      t0 = std::min(t0, st[(size_t)b * 8 + 5]);
      t1 = std::max(t1, st[(size_t)b * 8 + 6]);
    }
    for (int b = 0; b < nwg; ++b) starts.push_back((double)(st[(size_t)b * 8 + 5] - t0));
    std::sort(starts.begin(), starts.end());
    printf("  %d workgroups; mean lifetime %.0f cycles; kernel span %.0f cycles (s_memtime ticks); per workgroup: ids + A issue %.0f, "
           "A wait + split %.0f, barrier %.0f, MFMA loops %.0f (4 column tiles), epilogues %.0f\n",
           nwg, life / nwg, (double)(t1 - t0), sum[0] / nwg, sum[1] / nwg, sum[2] / nwg, sum[3] / nwg, sum[4] / nwg);
    printf("  start offsets: median %.0f, 60th pct %.0f, 75th %.0f, 90th %.0f, max %.0f\n", starts[nwg / 2], starts[nwg * 6 / 10],
           starts[nwg * 3 / 4], starts[nwg * 9 / 10], starts[nwg - 1]);
  }
  return 0;
}

// Diagnostic build of the fused PNA edge kernel (gnx_pna_edge_fwd) with in-kernel phase stamps: where a tile's time goes.
// Includes the product's kernel source with EF_STAMP defined (no stamp executes in the library build).  cfg-2's layer
// shape: N = 81 920 destination rows of in-degree 2 (E = 163 840), F = 128, one tower; sources near their destination.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Iinclude -Ignnepcsaft_amd/csrc \
//              -o /tmp/edge_fwd_stamp tools/ubench/edge_fwd_stamp.hip        ; run on the GPU box.
#define EF_STAMP 1
#include "../../gnnepcsaft_amd/csrc/gnx_fused.hip"

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
  const int64_t N = argc > 1 ? atoll(argv[1]) : 81920;
  const int deg = 2, F = 128, W = 65 - 4;
  const int64_t E = N * deg;
  std::vector<int> rowptr(N + 1), src(E), dst(E), code(E);
  unsigned s = 12345u;
  for (int64_t n = 0; n <= N; ++n) rowptr[n] = (int)(n * deg);
  for (int64_t e = 0; e < E; ++e) {
    s = s * 1664525u + 1013904223u;
    dst[e] = (int)(e / deg);
    int64_t j = dst[e] / 20 * 20 + (s >> 8) % 20;  // a neighbour inside the same 20-atom molecule
    src[e] = (int)(j < N ? j : N - 1);
    code[e] = (int)((s >> 20) % 60);
  }
  std::vector<float> hP((size_t)N * F), hW((size_t)F * F), hb(F);
  for (auto& v : hP) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) * 1e-3f; }
  for (auto& v : hW) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) * 1e-4f; }
  for (auto& v : hb) v = 0.01f;
  gnx_handle h;
  h.num_cus = 256;
  (void)hipMalloc(&h.d_flag, 256);
  (void)hipMemset(h.d_flag, 0, 256);
  float *P, *Q, *Te, *Wd, *bd, *h1, *m, *A;
  int *drp, *dsrc, *ddst, *dcode, *tinfo;
  (void)hipMalloc(&P, N * F * 4); (void)hipMalloc(&Q, N * F * 4); (void)hipMalloc(&Te, 60 * F * 4);
  (void)hipMalloc(&Wd, F * F * 4); (void)hipMalloc(&bd, F * 4);
  (void)hipMalloc(&h1, E * F * 4); (void)hipMalloc(&m, E * F * 4); (void)hipMalloc(&A, N * 4 * F * 4);
  (void)hipMalloc(&drp, (N + 1) * 4); (void)hipMalloc(&dsrc, E * 4); (void)hipMalloc(&ddst, E * 4); (void)hipMalloc(&dcode, E * 4);
  const int ntiles = gnx_edge_tiles_count(E, W);
  (void)hipMalloc(&tinfo, (ntiles + 1) * 8);
  (void)hipMalloc(&ef_stamp_buf, 256 * 8 * 8);
  (void)hipMemcpy(P, hP.data(), N * F * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(Q, hP.data(), N * F * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(Te, hP.data(), 60 * F * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(Wd, hW.data(), F * F * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(bd, hb.data(), F * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(drp, rowptr.data(), (N + 1) * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dsrc, src.data(), E * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(ddst, dst.data(), E * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dcode, code.data(), E * 4, hipMemcpyHostToDevice);
  if (gnx_edge_tiles(&h, drp, N, E, W, tinfo) != GNX_OK) return 1;
  const float* W1[1] = {Wd};
  const float* b1[1] = {bd};
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int keep = 1; keep >= 0; --keep) {
    for (int i = 0; i < 3; ++i)
      if (gnx_pna_edge_fwd(&h, P, Q, Te, dsrc, ddst, dcode, drp, tinfo, W, N, E, 1, F, W1, b1, keep ? h1 : nullptr,
                           keep ? m : nullptr, A) != GNX_OK) return 1;
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i)
      (void)gnx_pna_edge_fwd(&h, P, Q, Te, dsrc, ddst, dcode, drp, tinfo, W, N, E, 1, F, W1, b1, keep ? h1 : nullptr,
                             keep ? m : nullptr, A);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> st(256 * 8);
    (void)hipMemcpy(st.data(), ef_stamp_buf, st.size() * 8, hipMemcpyDeviceToHost);
    double sum[8] = {0};
    for (int b = 0; b < 256; ++b)
      for (int i = 0; i < 8; ++i) sum[i] += (double)st[b * 8 + i];
    double tot = 0;
    for (int i = 0; i < 8; ++i) tot += sum[i];
    printf("keep h1/m = %d: %.1f us per launch (%d tiles, %.2f per CU); wave 0 phase shares: issue %.1f%%  mfma %.1f%%  "
           "barrier-C %.1f%%  C->LDS %.1f%%  gather+split %.1f%%  barrier-A %.1f%%  stores+aggregate %.1f%%   "
           "(cycles per tile %.0f)\n",
           keep, ms * 1e3 / reps, ntiles, ntiles / 256.0, 100 * sum[0] / tot, 100 * sum[1] / tot, 100 * sum[2] / tot,
           100 * sum[3] / tot, 100 * sum[4] / tot, 100 * sum[5] / tot, 100 * sum[6] / tot, tot / ntiles);
  }
  return 0;
}

// Internal helpers shared by the gnx_*.hip translation units (gfx950 only; no dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "gnx.h"

struct gnx_handle {
  int device = 0;
  int num_cus = 0;
  hipStream_t stream = nullptr;
  // sticky device-side range flag + small scratch (handle state, not tensor memory)
  int* d_flag = nullptr;
  float* d_scratch = nullptr;  // 4 KiB
  float* d_zero = nullptr;     // 256 zero bytes, never written: what a masked-out vector load reads
  // side streams (gnx_side_begin/end/join): created on first use; `stream` is swapped to one between begin and end
  static constexpr int kSideStreams = 3;  // 0: weight gradients, 1: bond-table chain, 2: weight-image splits ahead of their products
  hipStream_t side[kSideStreams] = {nullptr, nullptr, nullptr};
  hipStream_t main_saved = nullptr;
  hipEvent_t side_fork[kSideStreams] = {nullptr, nullptr, nullptr}, side_done[kSideStreams] = {nullptr, nullptr, nullptr};
  bool on_side = false;
  // A/B switches (gnx_set_option); initialised ONCE from the GNX_* environment variables in gnx_create
  int opt[GNX_OPT_COUNT] = {};
  // profiling: bit k of prof_mask = record an event pair around every launch group of kernel id k
  unsigned prof_mask = 0;
  std::vector<hipEvent_t> ev;
  std::vector<int> ev_kid;  // kernel id of the event pair starting at ev[2*i]
  std::vector<double> ev_work;  // 3 per pair: algorithmic bytes, algorithmic flops, executed bf16-MFMA flops
  size_t ev_used = 0;
};

void gnx_set_error(const char* fmt, ...);

#define GNX_CHECK_ARG(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      gnx_set_error(__VA_ARGS__);         \
      return GNX_E_INVALID;               \
    }                                     \
  } while (0)

#define GNX_HIP(call)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      gnx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_));      \
      return GNX_E_HIP;                                                                        \
    }                                                                                          \
  } while (0)

#define GNX_LAUNCH_CHECK()                                                                     \
  do {                                                                                         \
    hipError_t e_ = hipGetLastError();                                                         \
    if (e_ != hipSuccess) {                                                                    \
      gnx_set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(e_));  \
      return GNX_E_HIP;                                                                        \
    }                                                                                          \
  } while (0)

// Records an event pair around the launches issued while it is alive, if the handle profiles kernel `kid`.
// `dispatch_timed`: the scope holds exactly ONE kernel launch and its two events are attached to that dispatch
// (GNX_LAUNCH_TIMED -> hipExtLaunchKernelGGL start / stop events): their distance is the kernel's own execution time,
// the quantity rocprofv3's kernel trace reports, without the ~3-4 us of marker packets and dispatch latency that an
// event pair recorded AROUND a launch includes.
struct gnx_prof_scope {
  gnx_handle* h;
  bool on;
  bool dispatch_timed;
  hipEvent_t e_start = nullptr, e_stop = nullptr;
  gnx_prof_scope(gnx_handle* h_, int kid, double bytes = 0.0, double flops = 0.0, double mfma = 0.0,
                 bool dispatch_timed_ = false)
      : h(h_), on(((h_->prof_mask >> kid) & 1u) != 0 && kid != GNX_K_NONE), dispatch_timed(dispatch_timed_) {
    if (on) {
      const size_t i = h->ev_used / 2;
      if (h->ev_kid.size() <= i) h->ev_kid.resize(i + 1);
      if (h->ev_work.size() < 3 * (i + 1)) h->ev_work.resize(3 * (i + 1));
      h->ev_kid[i] = kid;
      h->ev_work[3 * i] = bytes;
      h->ev_work[3 * i + 1] = flops;
      h->ev_work[3 * i + 2] = mfma;
      if (dispatch_timed) {
        e_start = take();
        e_stop = take();
        if (!e_start || !e_stop) on = false;
      } else {
        mark();
      }
    }
  }
  ~gnx_prof_scope() {
    if (on && !dispatch_timed) mark();
  }
  hipEvent_t take() {  // the next event of the handle's pool, not recorded
    if (h->ev_used == h->ev.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      h->ev.push_back(e);
    }
    return h->ev[h->ev_used++];
  }
  void mark() {
    hipEvent_t e = take();
    if (!e) {
      on = false;
      return;
    }
    (void)hipEventRecord(e, h->stream);
  }
};

// One kernel launch inside a dispatch_timed scope: with profiling on, the scope's events are attached to the dispatch;
// otherwise an ordinary launch (also what a stream capture sees).
#define GNX_LAUNCH_TIMED(prof, kernel, grid, block, shmem, stream, ...)                                        \
  do {                                                                                                         \
    if ((prof).on && (prof).dispatch_timed)                                                                    \
      hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, (prof).e_start, (prof).e_stop, 0, __VA_ARGS__); \
    else                                                                                                       \
      hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                     \
  } while (0)

static inline int64_t gnx_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// reads + clears the sticky range flag (synchronises the stream)
int32_t gnx_read_flag(gnx_handle* h, int* value);

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Zero-fill as a kernel launch: hipMemsetAsync costs ~50 us of host time per call on this stack (it showed up as idle
// gaps of that size between the packer's tiny kernels); a launch costs ~5 us.
template <typename T>
__global__ void k_zero_fill(T* __restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = T(0);
}
static inline void gnx_zero_ints(gnx_handle* h, int* p, int64_t n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_zero_fill<int>, dim3((unsigned)gnx_cdiv(n, 256)), dim3(256), 0, h->stream, p, n);
}


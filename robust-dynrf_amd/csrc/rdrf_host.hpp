// rdrf_host.hpp -- host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rdrf_common.hpp"

void rdrf_set_error(const char* fmt, ...);

// A/B switches of tools/ab_*.sh and tools/abl_*.sh.  The product library never reads the caller's environment: a C-ABI
// call's behaviour depends on its arguments (and on explicit rdrf_set_* calls) only.  `make tools` builds
// librodynrf_tools.so with -DRDRF_TOOLS, where these lookups and the RDRF_ABL_* ablation blocks are live.
#ifdef RDRF_TOOLS
#include <stdlib.h>
#define RDRF_ENV(name) getenv(name)
#else
#define RDRF_ENV(name) ((const char*)nullptr)
#endif

#define RDRF_CHECK(cond, code, ...)  \
  do {                               \
    if (!(cond)) {                   \
      rdrf_set_error(__VA_ARGS__);   \
      return (code);                 \
    }                                \
  } while (0)

#define RDRF_HIP(expr)                                                             \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess) {                                                        \
      rdrf_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return -5;                                                                   \
    }                                                                              \
  } while (0)

// byte fill as a kernel launch (rdrf_pack.hip): the launch sequences use it instead of hipMemsetAsync, whose memset nodes do
// not replay reliably inside a captured HIP graph
int rdrf_fill_async(void* p, int byte_value, size_t bytes, hipStream_t stream);
#define RDRF_FILL(p, v, bytes, stream)                          \
  do {                                                          \
    int rc_fill_ = rdrf_fill_async((p), (v), (bytes), (stream)); \
    if (rc_fill_) return rc_fill_;                              \
  } while (0)

// optional per-kernel timing with HIP events on the launch stream (bench.py roofline figure)
void rdrf_prof_begin(const char* name, hipStream_t s);
void rdrf_prof_end(const char* name, hipStream_t s);

#define RDRF_LAUNCH(name, kernel, grid, block, stream, ...)            \
  do {                                                                 \
    rdrf_prof_begin(name, stream);                                     \
    hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);   \
    rdrf_prof_end(name, stream);                                       \
    RDRF_HIP(hipGetLastError());                                       \
  } while (0)

static inline Box make_box(const RdrfFieldCfg* cfg) {
  Box b;
  for (int i = 0; i < 3; ++i) {
    b.lo[i] = cfg->aabb[i];
    b.hi[i] = cfg->aabb[3 + i];
    b.inv[i] = 2.0f / (cfg->aabb[3 + i] - cfg->aabb[i]);  // models/tensorBase.py:377
  }
  return b;
}

// workspace carving (all offsets 256-byte aligned)
struct WsCarver {
  char* base;
  size_t off, cap;
  WsCarver(void* p, size_t c) : base((char*)p), off(0), cap(c) {}
  template <typename T>
  T* take(size_t n) {
    size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
    T* r = (T*)(base + off);
    off += bytes;
    return r;
  }
  bool ok() const { return off <= cap; }
};

// component counts + the three planes / lines span ONE grid (plane XY <-> line Z, XZ <-> Y, YZ <-> X): the forward
// kernels compute one tap per grid axis and share it between the planes (rdrf_common.hpp, shared-tap gather)
static inline bool vm_one_grid(const RdrfVM& v) {
  return v.W[0] == v.W[1] && v.W[0] == v.L[2] && v.H[0] == v.W[2] && v.H[0] == v.L[1] && v.H[1] == v.H[2] && v.H[1] == v.L[0];
}
static inline bool vm_same_grid(const RdrfVM& a, const RdrfVM& b) {
  return a.W[0] == b.W[0] && a.H[0] == b.H[0] && a.L[0] == b.L[0];
}
static inline bool vm_ok(const RdrfVM& v, int c0, int c1) {
  return v.C[0] == c0 && v.C[1] == c1 && v.C[2] == c1 && vm_one_grid(v);
}

// device-wide stable key sort (rdrf_sort.hip)
size_t rdrf_sort_temp_bytes(unsigned n, int bits);
int rdrf_sort_positions(const unsigned* keys_in, unsigned* keys_out, unsigned* vals_out, unsigned n, int bits, void* temp,
                        size_t temp_bytes, hipStream_t stream, const int* n_dev = nullptr, unsigned n_mul = 0);

int rdrf_sort_ints_inplace(int* data, unsigned n, hipStream_t stream);   // deterministic build only

// pack-job builders (rdrf_pack.hip)
void pack_add(PackJobs& J, const float* src, int ld, int out_dim, int in_dim, int seg, int mode,
              int nb, int kk, int dst);
void pack_add_b3(PackJobs& J, const float* src, int ld, int out_dim, int in_dim, int seg, int nb, int kk, int seg_kk0, int kk_off,
                 int kk_tot, int dst);
void pack_add_b3s(PackJobs& J, const float* src, int ld, int out_dim, int in_dim, int seg, int nb, int kk, int seg_kk0, int kk_off,
                  int kk_tot, int dst, int dst_lo);
void pack_add_b3s_t(PackJobs& J, const float* src, int ld, int out_dim, int in_dim, int seg, int nbi, int kk, int dst, int dst_lo);
void pack_add_from(PackJobs& J, const float* src, int ld, int out_dim, int in_dim, int seg, int nb, int kk, int seg_kk0, int dst);
int pack_launch(const PackJobs& J, float* dst, hipStream_t stream);

// Does the in-order vmcnt counter throttle a gather + atomic loop?  Each wave loops over "iterations" of
// NL random 16-byte loads (from a 2 MB L2-resident table) whose values feed NA no-return fp32 atomic
// instructions (random 64-byte lines of a 4 MB buffer, quads of 4 lanes).
//   serial   : loads(i) -> wait -> atomics(i)            the wait for loads(i+1) drains atomics(i) (in order)
//   pipelined: loads(i+1) issued BEFORE atomics(i)       the wait for loads(i+1) leaves atomics(i) in flight
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NL = 6, NA = 12;

__device__ inline unsigned long long rnd(unsigned long long& st) { st = st * 6364136223846793005ull + 1442695040888963407ull; return st >> 20; }

template <int PIPE>
__global__ __launch_bounds__(256) void k_loop(const f32x4* __restrict__ table, long tn, float* __restrict__ grad, long lines, int iters, int quads) {
  const int lane = threadIdx.x & 63;
  unsigned long long st = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 99;
  f32x4 cur[NL], nxt[NL];
  auto issue = [&](f32x4 (&v)[NL]) {
#pragma unroll
    for (int i = 0; i < NL; ++i) v[i] = table[rnd(st) % (unsigned long long)tn];
  };
  auto atomics = [&](const f32x4 (&v)[NL], unsigned long long& qs) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NL; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      // a quad of 4 lanes shares the line (same qs in the 4 lanes): 16 lines per instruction
      const long line = (long)(rnd(qs) % (unsigned long long)lines);
      if ((lane >> 2) < quads) __hip_atomic_fetch_add(grad + line * 16 + (lane & 3), s * 1e-9f + 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  unsigned long long qs = ((unsigned long long)blockIdx.x * blockDim.x + (threadIdx.x & ~3)) * 0xD1B54A32D192ED03ull + 7;
  if (PIPE) {
    issue(cur);
    for (int it = 0; it < iters; ++it) {
      issue(nxt);
      atomics(cur, qs);
#pragma unroll
      for (int i = 0; i < NL; ++i) cur[i] = nxt[i];
    }
  } else {
    for (int it = 0; it < iters; ++it) {
      issue(cur);
      atomics(cur, qs);
    }
  }
}

int main() {
  const long tn = 131072;      // 2 MB of float4
  const long lines = 65536;    // 4 MB of gradient lines
  f32x4* table; float* grad;
  CK(hipMalloc(&table, tn * 16)); CK(hipMemset(table, 0, tn * 16));
  CK(hipMalloc(&grad, lines * 64)); CK(hipMemset(grad, 0, lines * 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 200;
  for (int quads : {16, 4, 2})
  for (int wpc : {12, 16}) {
    const int blocks = 256 * wpc / 4;
    for (int pipe = 0; pipe < 2; ++pipe) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (pipe) k_loop<1><<<blocks, 256>>>(table, tn, grad, lines, iters, quads);
        else k_loop<0><<<blocks, 256>>>(table, tn, grad, lines, iters, quads);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      const double req = (double)blocks * 4 * iters * NA * quads;
      printf("%2d quads/instr %2d waves/CU  %-9s %.3f ms  %.1f G line-requests/s\n", quads, wpc, pipe ? "pipelined" : "serial", best, req / best / 1e6);
    }
  }
  return 0;
}

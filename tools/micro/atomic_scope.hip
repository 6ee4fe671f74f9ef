// Microbenchmark: fp32 quad atomics (4 lanes -> one 16-byte texel) on gfx950 by memory scope.
//   A  agent scope, one buffer            (what k_scatter does today)
//   B  agent scope, per-XCD private copy  (isolates the effect of splitting the buffer)
//   C  workgroup scope, per-XCD private copy chosen by HW_REG_XCC_ID (L2-local atomics)
// Each variant adds the same deterministic pseudo-random texel sequence; the sums are integers so the
// reduced result must be bit-identical between variants.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ inline int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7; }  // HW_REG_XCC_ID[3:0]

template <int MODE>
__global__ __launch_bounds__(512) void k_atom(float* __restrict__ buf, long texels, int iters, int* __restrict__ xcc_seen) {
  const long quad = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const int comp = threadIdx.x & 3;
  const int xcc = xcc_id();
  if (threadIdx.x == 0) atomicOr(xcc_seen + (blockIdx.x & 1023), 1 << xcc);
  float* base = buf + (MODE == 0 ? 0 : (size_t)xcc * texels * 4);
  unsigned long long st = quad * 0x9E3779B97F4A7C15ull + 12345;
  for (int i = 0; i < iters; ++i) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    const long t = (long)((st >> 20) % (unsigned long long)texels);
    float* p = base + t * 4 + comp;
    if (MODE == 2) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// G lanes add G consecutive floats (one G*4-byte block, naturally aligned): which span is one request?
template <int G>
__global__ __launch_bounds__(512) void k_width(float* __restrict__ buf, long blocks_of_g, int iters) {
  const long grp = ((long)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const int comp = threadIdx.x % G;
  unsigned long long st = grp * 0x9E3779B97F4A7C15ull + 777;
  for (int i = 0; i < iters; ++i) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    const long t = (long)((st >> 20) % (unsigned long long)blocks_of_g);
    __hip_atomic_fetch_add(buf + t * G + comp, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void k_reduce(const float* __restrict__ buf, float* __restrict__ out, long n, int copies) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int c = 0; c < copies; ++c) s += buf[(size_t)c * n + i];
  out[i] = s;
}

int main(int argc, char** argv) {
  const long texels = argc > 1 ? atol(argv[1]) : 50000;
  const int iters = argc > 2 ? atoi(argv[2]) : 64;
  const int blocks = argc > 3 ? atoi(argv[3]) : 768;
  const long n = texels * 4;
  float *buf, *outA, *outB, *outC;
  int* seen;
  CK(hipMalloc(&buf, n * 8 * sizeof(float)));
  CK(hipMalloc(&outA, n * sizeof(float))); CK(hipMalloc(&outB, n * sizeof(float))); CK(hipMalloc(&outC, n * sizeof(float)));
  CK(hipMalloc(&seen, 1024 * sizeof(int)));
  CK(hipMemset(seen, 0, 1024 * sizeof(int)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double reqs = (double)blocks * 512 / 4 * iters;
  float* outs[3] = {outA, outB, outC};
  const char* names[3] = {"A agent/one-buffer", "B agent/xcd-copies", "C workgroup/xcd-copies"};
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemset(buf, 0, n * 8 * sizeof(float)));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      if (mode == 0) k_atom<0><<<blocks, 512>>>(buf, texels, iters, seen);
      if (mode == 1) k_atom<1><<<blocks, 512>>>(buf, texels, iters, seen);
      if (mode == 2) k_atom<2><<<blocks, 512>>>(buf, texels, iters, seen);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    k_reduce<<<(n + 255) / 256, 256>>>(buf, outs[mode], n, mode == 0 ? 1 : 8);
    CK(hipDeviceSynchronize());
    printf("%-26s texels=%ld  %.3f ms  %.1f G quad-req/s\n", names[mode], texels, best, reqs / best / 1e6);
  }
  {
    const long floats = n;  // same footprint for every width
    auto run = [&](int G) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(buf, 0, floats * sizeof(float)));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (G == 1) k_width<1><<<blocks, 512>>>(buf, floats / 1, iters);
        if (G == 2) k_width<2><<<blocks, 512>>>(buf, floats / 2, iters);
        if (G == 4) k_width<4><<<blocks, 512>>>(buf, floats / 4, iters);
        if (G == 8) k_width<8><<<blocks, 512>>>(buf, floats / 8, iters);
        if (G == 16) k_width<16><<<blocks, 512>>>(buf, floats / 16, iters);
        if (G == 32) k_width<32><<<blocks, 512>>>(buf, floats / 32, iters);
        if (G == 64) k_width<64><<<blocks, 512>>>(buf, floats / 64, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      const double lanes = (double)blocks * 512 * iters;
      printf("width %3d B per group: %.3f ms  %.1f G groups/s  %.1f G lane-adds/s\n", G * 4, best, lanes / G / best / 1e6, lanes / best / 1e6);
    };
    for (int G : {1, 2, 4, 8, 16, 32, 64}) run(G);
  }
  std::vector<float> a(n), b(n), c(n);
  CK(hipMemcpy(a.data(), outA, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), outB, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(c.data(), outC, n * 4, hipMemcpyDeviceToHost));
  double sa = 0, sb = 0, sc = 0; long db = 0, dc = 0;
  for (long i = 0; i < n; ++i) { sa += a[i]; sb += b[i]; sc += c[i]; db += a[i] != b[i]; dc += a[i] != c[i]; }
  printf("sums: A %.0f B %.0f C %.0f expected %.0f ; entries differing from A: B %ld C %ld\n", sa, sb, sc, reqs * 4, db, dc);
  std::vector<int> hs(1024);
  CK(hipMemcpy(hs.data(), seen, 1024 * 4, hipMemcpyDeviceToHost));
  int multi = 0, mism = 0;
  for (int i = 0; i < 1024 && i < blocks; ++i) { multi += __builtin_popcount(hs[i]) > 1; mism += !(hs[i] & (1 << (i % 8))); }
  printf("block->xcc: %d of %d block slots saw >1 xcc, %d slots not on xcc b%%8\n", multi, blocks < 1024 ? blocks : 1024, mism);
  return 0;
}

// Microbenchmark: memory-side atomic request rate on gfx950 by DATA TYPE (after tools/micro/lds_atomic_rate.hip found
// ds_add_f32 11x slower than ds_add_f64): groups of 4 adjacent lanes add 4 adjacent elements of a pseudo-random texel
// of a ~3 MB gradient plane (the scatter kernels' pattern), device scope, no return.
//   f32 global_atomic_add_f32   u32 global_atomic_add (32-bit)   u64 global_atomic_add_x2   f64 global_atomic_add_f64
// Prints G requests/s (one request = one 4-lane group) and G lane-updates/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename T>
__global__ __launch_bounds__(512) void k_atom(T* __restrict__ buf, long texels, int iters) {
  const long quad = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const int comp = threadIdx.x & 3;
  unsigned long long st = quad * 0x9E3779B97F4A7C15ull + 12345;
  for (int i = 0; i < iters; ++i) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    const long t = (long)((st >> 20) % (unsigned long long)texels);
    __hip_atomic_fetch_add(buf + t * 4 + comp, (T)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <typename T>
static void run(const char* name, void* buf, long texels, int iters, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(buf, 0, texels * 4 * 8));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_atom<T><<<blocks, 512>>>((T*)buf, texels, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double reqs = (double)blocks * 512 / 4 * iters;
  printf("%-4s %8.3f ms  %6.2f G requests/s  %6.2f G lane-updates/s\n", name, best, reqs / (best * 1e-3) / 1e9, 4 * reqs / (best * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
  const long texels = argc > 1 ? atol(argv[1]) : 50000;
  const int iters = argc > 2 ? atoi(argv[2]) : 64;
  const int blocks = argc > 3 ? atoi(argv[3]) : 768;
  void* buf;
  CK(hipMalloc(&buf, texels * 4 * 8));
  printf("%ld texels of 4 elements, %d workgroups of 512, %d atomics per lane\n", texels, blocks, iters);
  run<float>("f32", buf, texels, iters, blocks);
  run<unsigned>("u32", buf, texels, iters, blocks);
  run<unsigned long long>("u64", buf, texels, iters, blocks);
  run<double>("f64", buf, texels, iters, blocks);
  return 0;
}

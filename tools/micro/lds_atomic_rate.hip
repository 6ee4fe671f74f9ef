// Microbenchmark: what does an LDS atomic add cost on gfx950 by data type?  The sorted scatter's XY pass spends ~40 % of
// its time in ds_add_f32 on the line accumulators (~150 cycles per 64-lane instruction).  Variants, all with
// the kernel's own access pattern (groups of 4 adjacent lanes add 4 adjacent elements of a random line entry, entry stride
// 20 elements, 512 threads x 3 workgroups per CU):
//   f32  ds_add_f32          (what k_scatter / k_scatter_sorted issue today)
//   u32  ds_add_u32          (integer: native in the LDS ALU?)
//   u64  ds_add_u64          (64-bit fixed point, the deterministic build's format)
//   f64  ds_add_f64          (gfx90a+: double accumulators)
//   f32r ds_add_rtn_f32      (with return, for reference)
// and, as the floor, plain ds_write_b32 to the same addresses.  Prints cycles per wave instruction (at 2.4 GHz) and
// lane-updates per clock per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ENTRIES = 94, STRIDE = 20;   // one density line: 94 entries of 16 components, padded stride 20

template <int MODE>
__global__ __launch_bounds__(512, 3) void k_lds(float* __restrict__ out, int iters) {
  __shared__ double lacc64[ENTRIES * STRIDE];
  float* lacc = reinterpret_cast<float*>(lacc64);
  for (int i = threadIdx.x; i < ENTRIES * STRIDE; i += blockDim.x) lacc64[i] = 0.0;
  __syncthreads();
  const int c = threadIdx.x & 3;
  unsigned st = (blockIdx.x * blockDim.x + (threadIdx.x >> 2)) * 2654435761u + 12345u;
  for (int i = 0; i < iters; ++i) {
    st = st * 1664525u + 1013904223u;
    const int e = (st >> 10) % ENTRIES;
    const int q = (st >> 4) & 3;
    const int a = e * STRIDE + 4 * q + c;
    if (MODE == 0) __hip_atomic_fetch_add(lacc + a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 1) __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(lacc) + a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 2) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(lacc64) + a, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 3) __hip_atomic_fetch_add(lacc64 + a, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 4) {
      const float r = __hip_atomic_fetch_add(lacc + a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (r == -1.0f) out[0] = r;
    }
    if (MODE == 5) reinterpret_cast<volatile float*>(lacc)[a] = 1.0f;
  }
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < ENTRIES * STRIDE; i += blockDim.x) s += (MODE == 2 || MODE == 3) ? (float)lacc64[i] : lacc[i];
  if (s == -1.0f) out[1] = s;
}

template <int MODE>
static void run(const char* name, float* out, int blocks, int iters, int cus) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_lds<MODE><<<blocks, 512>>>(out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double wave_instr = (double)blocks * 8 * iters, per_cu = wave_instr / cus;
  const double cyc = best * 1e-3 * 2.4e9 / per_cu;
  printf("%-5s %8.3f ms  %7.1f cycles per wave instruction per CU   %6.2f lane-updates / clk / CU\n", name, best, cyc, 64.0 / cyc);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4096;
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount, blocks = cus * 3;
  float* out; CK(hipMalloc(&out, 16));
  printf("%s: %d CUs, %d workgroups of 512, %d atomics per lane\n", p.name, cus, blocks, iters);
  run<0>("f32", out, blocks, iters, cus);
  run<1>("u32", out, blocks, iters, cus);
  run<2>("u64", out, blocks, iters, cus);
  run<3>("f64", out, blocks, iters, cus);
  run<4>("f32r", out, blocks, iters, cus);
  run<5>("store", out, blocks, iters, cus);
  return 0;
}

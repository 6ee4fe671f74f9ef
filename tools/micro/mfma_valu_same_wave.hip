// Microbenchmark (round 6, companion of mfma_valu_overlap.hip): are independent fp32 VALU instructions hidden when they sit in
// the SAME wave's instruction stream between its fp32 MFMAs?  One or two waves per SIMD; per loop iteration 4 MFMAs
// (v_mfma_f32_32x32x2_f32, four accumulators = 256 pipe cycles) and F independent v_fma_f32 placed between them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int F>
__global__ __launch_bounds__(512) void k_same(float* __restrict__ out, int n) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
  float a = lane * 1e-3f, b = lane * 2e-3f;
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = lane * 1e-3f + j;
  const float m = 1.0000001f, c = 1e-7f;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[nb], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < F / 4; ++j) v[(nb * (F / 4) + j) & 15] = __builtin_fmaf(v[(nb * (F / 4) + j) & 15], m, c);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[nb][r];
#pragma unroll
  for (int j = 0; j < 16; ++j) s += v[j];
  if (s == -1.f) out[0] = s;
}

template <int F>
static void run(float* out, int block) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int n = 20000;
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_same<F><<<256, block>>>(out, n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("waves/SIMD %d, %2d v_fma per 4 MFMAs: %7.3f ms  = %6.1f cycles per iteration per wave at 2.4 GHz (MFMA alone: 256)\n", block / 256, F, best,
         best * 1e-3 * 2.4e9 / n / (block / 256));
}

int main() {
  float* out;
  CK(hipMalloc(&out, 64));
  for (int block : {256, 512}) {
    run<0>(out, block); run<8>(out, block); run<16>(out, block); run<32>(out, block); run<48>(out, block); run<64>(out, block);
  }
  return 0;
}

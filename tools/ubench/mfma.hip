// fp32 MFMA issue microbenchmark for gfx950: v_mfma_f32_32x32x2_f32 in (a) one dependent chain per
// wave, (b) 2 / (c) 4 independent accumulator chains interleaved, at 1 / 2 / 4 waves per SIMD.
// Reports TFLOP/s against the 157.3 peak.  hipcc -O3 --offload-arch=gfx950 mfma.hip -o mfma.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(const char* name, int blocks_per_cu) {
  const int blocks = 256 * blocks_per_cu, iters = 2048 / NACC;
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double flops = (double)blocks * 4 * iters * 16 * NACC * 4096.0;
  printf("%-22s waves/SIMD %d  %.3f ms  %.1f TFLOP/s (%.0f %% of 157.3)\n", name, blocks_per_cu, best,
         flops / best / 1e9, flops / best / 1e9 / 157.3 * 100);
  hipFree(out);
}
int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run<1>("1 chain / wave", w);
    run<2>("2 chains interleaved", w);
    run<4>("4 chains interleaved", w);
  }
  return 0;
}

// Microbenchmark (round 6): what does the fp32 matrix pipe of one SIMD sustain for the two f32 MFMA shapes, in the operand
// patterns of the MLP kernels, as a function of the waves per SIMD that share it?
//   32x32x2, 4 accumulators  (k_static_app: four 32-neuron blocks of a 32-sample tile, weights streamed from LDS)
//   16x16x4, 8 accumulators  (k_static_app16: eight 16-neuron blocks of a 16-sample tile)
// variants: operands in registers only ("reg") or the A operand streamed from LDS exactly like mfma_seg / mfma16_seg ("lds").
// Prints cycles per MFMA per SIMD from s_memtime (shader clock) and the TFLOP/s the wall clock gives.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: 32x32x2 reg, 1: 16x16x4 reg, 2: 32x32x2 lds, 3: 16x16x4 lds
template <int MODE>
__global__ __launch_bounds__(1024) void k_mfma(float* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float w[8 * 32 * 64 * 2];   // 128 KB: one layer's fragments, both layouts
  for (int i = threadIdx.x; i < 8 * 32 * 64 * 2; i += blockDim.x) w[i] = (float)((i * 2654435761u) >> 20) * 1e-6f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float in[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) in[i] = (float)(lane + i) * 1e-3f;
  unsigned long long t0 = __builtin_readcyclecounter();
  if constexpr (MODE == 0 || MODE == 2) {
    f32x16 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      // one 128 -> 128 layer of a 32-sample tile: 4 blocks x 64 k-steps = 256 MFMAs
      f32x4 wc[4], wn[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) wc[nb] = MODE == 2 ? *(const f32x4*)(w + (((nb * 16) * 64 + lane) << 2)) : f32x4{in[nb], in[nb + 1], in[nb + 2], in[nb + 3]};
#pragma unroll
      for (int k4 = 0; k4 < 16; ++k4) {
        if (k4 + 1 < 16) {
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) wn[nb] = MODE == 2 ? *(const f32x4*)(w + (((nb * 16 + k4 + 1) * 64 + lane) << 2)) : f32x4{in[nb + k4], in[nb + 1], in[nb + 2], in[nb + 3]};
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[nb].x, in[(k4 * 4 + 0) & 31], acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[nb].y, in[(k4 * 4 + 1) & 31], acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[nb].z, in[(k4 * 4 + 2) & 31], acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[nb].w, in[(k4 * 4 + 3) & 31], acc[nb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) wc[nb] = wn[nb];
      }
    }
    float s = 0.f;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[nb][r];
    if (s == -1.f) out[0] = s;
  } else {
    f32x4 acc[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
      // one 128 -> 128 layer of a 16-sample tile: 8 blocks x 32 k-steps = 256 MFMAs
      f32x2 wc[8], wn[8];
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) wc[nb] = MODE == 3 ? *(const f32x2*)(w + (((nb * 16) * 64 + lane) << 1)) : f32x2{in[nb], in[nb + 1]};
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) {
        if (k2 + 1 < 16) {
#pragma unroll
          for (int nb = 0; nb < 8; ++nb) wn[nb] = MODE == 3 ? *(const f32x2*)(w + (((nb * 16 + k2 + 1) * 64 + lane) << 1)) : f32x2{in[nb + k2], in[nb + 1]};
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[nb].x, in[(k2 * 2 + 0) & 31], acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[nb].y, in[(k2 * 2 + 1) & 31], acc[nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) wc[nb] = wn[nb];
      }
    }
    float s = 0.f;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) s += acc[nb].x + acc[nb].y + acc[nb].z + acc[nb].w;
    if (s == -1.f) out[0] = s;
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
static void run(const char* name, float* out, unsigned long long* cyc, int waves_per_simd, int iters) {
  const int block = 256 * waves_per_simd, blocks = 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_mfma<MODE><<<blocks, block>>>(out, cyc, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const int nw = blocks * (block / 64);
  unsigned long long* h = (unsigned long long*)malloc(nw * 8);
  CK(hipMemcpy(h, cyc, nw * 8, hipMemcpyDeviceToHost));
  double mean = 0; for (int i = 0; i < nw; ++i) mean += (double)h[i]; mean /= nw;
  free(h);
  const double mf_per_wave = 256.0 * iters;
  const double flop = (MODE == 0 || MODE == 2 ? 4096.0 : 2048.0) * mf_per_wave * nw;
  // s_memtime / readcyclecounter ticks at a fixed 100 MHz on gfx9: report wall-clock figures only when the tick is not the shader clock
  printf("%-14s waves/SIMD %d: %7.3f ms  %6.1f TFLOP/s  (%.1f ns per MFMA per SIMD)  timer ticks per wave %.0f\n", name, waves_per_simd, best,
         flop / (best * 1e-3) / 1e12, best * 1e6 / (mf_per_wave * waves_per_simd), mean);
}

int main() {
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 256 * 16 * 8));
  const int iters = 400;
  for (int w = 1; w <= 4; ++w) {
    run<0>("32x32x2 reg", out, cyc, w, iters);
    run<2>("32x32x2 lds", out, cyc, w, iters);
    run<1>("16x16x4 reg", out, cyc, w, iters);
    run<3>("16x16x4 lds", out, cyc, w, iters);
  }
  return 0;
}

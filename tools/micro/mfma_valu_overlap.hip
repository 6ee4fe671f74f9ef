// Microbenchmark (round 6): do fp32 MFMAs and ordinary fp32 VALU instructions of DIFFERENT waves on one SIMD overlap?
// 512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run a pure v_mfma_f32_32x32x2_f32 chain (4 accumulators),
// waves 4-7 (their SIMD partners) a pure v_fma_f32 chain (8 independent accumulators).  Three launches: MFMA waves only,
// VALU waves only, both.  If the two pipes were independent, `both` would take max(a, b); if the f32 MFMA executes on the
// same lanes as the f32 VALU, a + b.  Also the same with a bf16 MFMA (v_mfma_f32_32x32x16_bf16) for contrast.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int BF16>
__global__ __launch_bounds__(512) void k_mix(float* __restrict__ out, int n_mfma, int n_valu, int which) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < 4) {
    if (!(which & 1)) return;
    f32x16 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    float a = lane * 1e-3f, b = lane * 2e-3f;
    bf16x8 ab, bb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(lane * 1e-3f); bb[i] = (__bf16)(lane * 2e-3f); }
    for (int i = 0; i < n_mfma; ++i) {
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        if constexpr (BF16) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[nb], 0, 0, 0);
        else acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[nb], 0, 0, 0);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[nb][r];
    if (s == -1.f) out[0] = s;
  } else {
    if (!(which & 2)) return;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = lane * 1e-3f + j;
    const float m = 1.0000001f, c = 1e-7f;
    for (int i = 0; i < n_valu; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], m, c);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    if (s == -1.f) out[1] = s;
  }
}

template <int BF16>
static float run(float* out, int n_mfma, int n_valu, int which) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_mix<BF16><<<256, 512>>>(out, n_mfma, n_valu, which);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  float* out;
  CK(hipMalloc(&out, 64));
  const int n_mfma = 20000;   // x 4 MFMAs
  for (int nv : {20000, 80000, 160000, 320000}) {   // x 8 FMAs
    const float a = run<0>(out, n_mfma, nv, 1), b = run<0>(out, n_mfma, nv, 2), ab = run<0>(out, n_mfma, nv, 3);
    printf("f32  MFMA 32x32x2 : mfma only %7.3f ms (%d MFMAs/wave), valu only %7.3f ms (%d FMAs/wave), both %7.3f ms  [max %7.3f  sum %7.3f]\n",
           a, 4 * n_mfma, b, 8 * nv, ab, a > b ? a : b, a + b);
  }
  for (int nv : {20000, 80000, 160000}) {
    const float a = run<1>(out, n_mfma, nv, 1), b = run<1>(out, n_mfma, nv, 2), ab = run<1>(out, n_mfma, nv, 3);
    printf("bf16 MFMA 32x32x16: mfma only %7.3f ms (%d MFMAs/wave), valu only %7.3f ms (%d FMAs/wave), both %7.3f ms  [max %7.3f  sum %7.3f]\n",
           a, 4 * n_mfma, b, 8 * nv, ab, a > b ? a : b, a + b);
  }
  return 0;
}

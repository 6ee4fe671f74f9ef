// Microbenchmark (round 6): an fp32-grade MLP layer on the bf16 matrix pipe.
//   x = x_hi + x_mid + x_lo  (three bf16 pieces: 8 + 8 + 8 mantissa bits = the whole fp32 significand, exact by truncation)
//   W x ~= Wh xh + Wh xm + Wm xh + Wh xl + Wl xh + Wm xm          (the six products above 2^-24 relative)
// on v_mfma_f32_32x32x16_bf16 (32 cycles per SIMD for K = 16: the six cost 192 cycles against 8 x 64 = 512 for the same K on
// v_mfma_f32_32x32x2_f32), activations split in registers (VALU), weights pre-split in LDS (1.5 x the fp32 bytes).
// Checks (1) the operand layout -- the lane's `in[kk]` slots of the TRANSPOSED form used by every MLP kernel
// (rdrf_common.hpp: element e of a sample's vector in lane half h = (e>>2)&1, slot kk = (e>>3)*4 + (e&3)) feed the bf16
// instruction as eight consecutive slots per K = 16 step --, (2) the error of both forms against an fp64 product,
// (3) cycles per layer per wave at 1 / 2 waves per SIMD, weights streamed from LDS as in mfma_seg,
// (4) that repeated launches return the same bits (an earlier form with the lo pieces consumed by the third MFMA of a step did not).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/bf16x3_layer.hip -o tools/micro/bf16x3_layer && tools/micro/bf16x3_layer
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef OUT_N
#define OUT_N 64
#endif
constexpr int OUT = OUT_N, NBO = OUT / 32;   // neurons (two 32-row blocks)
constexpr int KK = 80, K = 2 * KK;        // slots per lane half, input elements (10 K = 16 steps)
constexpr int K8 = KK / 8, K4 = KK / 4;

__host__ __device__ inline int elem_of(int kk, int h) { return ((kk >> 2) << 3) + (h << 2) + (kk & 3); }

// ---- fp32 form: exactly mfma_seg<NBO, KK> ----------------------------------------------------------------------
__device__ __forceinline__ void seg_f32(f32x16 (&acc)[NBO], const float (&in)[KK], const float* __restrict__ wp, int lane) {
  f32x4 wc[NBO], wn[NBO];
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) wc[nb] = *(const f32x4*)(wp + (((nb * K4) * 64 + lane) << 2));
#pragma unroll
  for (int k4 = 0; k4 < K4; ++k4) {
    if (k4 + 1 < K4) {
#pragma unroll
      for (int nb = 0; nb < NBO; ++nb) wn[nb] = *(const f32x4*)(wp + (((nb * K4 + k4 + 1) * 64 + lane) << 2));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nb = 0; nb < NBO; ++nb) {
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[nb].x, in[k4 * 4 + 0], acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[nb].y, in[k4 * 4 + 1], acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[nb].z, in[k4 * 4 + 2], acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[nb].w, in[k4 * 4 + 3], acc[nb], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (k4 + 1 < K4) {
#pragma unroll
      for (int nb = 0; nb < NBO; ++nb) wc[nb] = wn[nb];
    }
  }
}

// ---- bf16 x 3 form ----------------------------------------------------------------------------------------------
// split by truncation: hi = top 16 bits of x; r = x - hi (exact); mid = top 16 bits of r; lo = top 16 bits of r - mid
__device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned xb = __float_as_uint(x);
  hi = xb & 0xffff0000u;
  const float r = x - __uint_as_float(hi);
  const unsigned rb = __float_as_uint(r);
  mid = rb & 0xffff0000u;
  lo = __float_as_uint(r - __uint_as_float(mid));   // the instruction reads its top 16 bits
}
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) {   // (a >> 16) | (b & 0xffff0000): one v_perm_b32
  return __builtin_amdgcn_perm(b, a, 0x07060302u);
}
// weights: [NBO][K8][3 pieces][64 lanes][4 dwords] (8 bf16 per lane and piece: one ds_read_b128)
__device__ __forceinline__ void seg_b3(f32x16 (&acc)[NBO], const float (&in)[KK], const unsigned* __restrict__ wp, int lane) {
  // the form of rdrf_common.hpp mfma_seg_b3: the step's weight pieces requested first, the split in three sweeps (hi, mid, lo),
  // the MFMAs in the order the pieces were produced; scheduling barriers keep hipcc from hoisting every split of the fully
  // unrolled layer to the top (148 spilled registers without them)
#pragma unroll
  for (int k8 = 0; k8 < K8; ++k8) {
    u32x4 wc[NBO][3];
#pragma unroll
    for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
      for (int p = 0; p < 3; ++p) wc[nb][p] = *(const u32x4*)(wp + ((size_t)((nb * K8 + k8) * 3 + p) * 64 + lane) * 4);
    __builtin_amdgcn_sched_barrier(0);
    unsigned hi[8], r1[8], mid[8];
    u32x4 bh, bm, bl;
#pragma unroll
    for (int e = 0; e < 8; ++e) hi[e] = __float_as_uint(in[k8 * 8 + e]) & 0xffff0000u;
#pragma unroll
    for (int q = 0; q < 4; ++q) bh[q] = pack_hi16(hi[2 * q], hi[2 * q + 1]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      r1[e] = __float_as_uint(in[k8 * 8 + e] - __uint_as_float(hi[e]));
      mid[e] = r1[e] & 0xffff0000u;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) bm[q] = pack_hi16(mid[2 * q], mid[2 * q + 1]);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bl[q] = pack_hi16(__float_as_uint(__uint_as_float(r1[2 * q]) - __uint_as_float(mid[2 * q])),
                        __float_as_uint(__uint_as_float(r1[2 * q + 1]) - __uint_as_float(mid[2 * q + 1])));
    const bf16x8 xh = __builtin_bit_cast(bf16x8, bh), xm = __builtin_bit_cast(bf16x8, bm), xl = __builtin_bit_cast(bf16x8, bl);
    __builtin_amdgcn_sched_barrier(0);
#define B3_STEP(WP, XP)                                                                                                   \
    _Pragma("unroll") for (int nb = 0; nb < NBO; ++nb)                                                                    \
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wc[nb][WP]), XP, acc[nb], 0, 0, 0);
    B3_STEP(0, xh) B3_STEP(1, xh) B3_STEP(2, xh) B3_STEP(0, xm) B3_STEP(1, xm) B3_STEP(0, xl)
#undef B3_STEP
    __builtin_amdgcn_sched_barrier(0);
  }
}

// x: [32 samples][K] row major; out: [OUT][32 samples]; one wave per launch block-wave, `iters` layers back to back
template <int MODE>
__global__ __launch_bounds__(512) void k_layer(const float* __restrict__ wf32, const unsigned* __restrict__ wb3, const float* __restrict__ x,
                                               float* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  const int nw = MODE == 0 ? NBO * K4 * 64 * 4 : NBO * K8 * 3 * 64 * 4;
  for (int i = threadIdx.x; i < nw; i += blockDim.x) lds[i] = MODE == 0 ? __float_as_uint(wf32[i]) : wb3[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, h = lane >> 5, s = lane & 31;
  float in[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) in[kk] = x[s * K + elem_of(kk, h)];
  f32x16 acc[NBO];
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) seg_f32(acc, in, (const float*)lds, lane);
    else seg_b3(acc, in, lds, lane);
    if (it + 1 < iters) {   // keep the next layer dependent on this one without changing the values much
#pragma unroll
      for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] *= 0.0f;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  if (blockIdx.x == 0 && threadIdx.x < 64) {
#pragma unroll
    for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[(nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + s] = acc[nb][r];
  }
}

static unsigned short bf16_trunc(float v) { unsigned b; memcpy(&b, &v, 4); return (unsigned short)(b >> 16); }
static float bf16_val(unsigned short v) { unsigned b = (unsigned)v << 16; float f; memcpy(&f, &b, 4); return f; }

int main() {
  std::vector<float> W(OUT * K), X(32 * K);
  srand(3);
  for (auto& v : W) v = ((rand() % 20001) - 10000) * 1e-4f * (1.0f + 0.37f * (rand() % 7));   // asymmetric, mixed magnitudes
  for (auto& v : X) v = ((rand() % 20001) - 10000) * 1e-4f * (rand() % 5 == 0 ? 30.f : 1.f);
  // pack: fp32 [NBO][K4][64][4]; bf16 x 3 [NBO][K8][3][64][8 bf16]
  std::vector<float> wf(NBO * K4 * 64 * 4);
  std::vector<unsigned> wb(NBO * K8 * 3 * 64 * 4);
  for (int nb = 0; nb < NBO; ++nb)
    for (int lane = 0; lane < 64; ++lane) {
      const int h = lane >> 5, neuron = nb * 32 + (lane & 31);
      for (int kk = 0; kk < KK; ++kk) {
        const float w = W[neuron * K + elem_of(kk, h)];
        wf[(((nb * K4 + kk / 4) * 64 + lane) << 2) + (kk & 3)] = w;
        const unsigned short hi = bf16_trunc(w);
        const float r = w - bf16_val(hi);
        const unsigned short mid = bf16_trunc(r);
        const unsigned short lo = bf16_trunc(r - bf16_val(mid));
        const unsigned short pc[3] = {hi, mid, lo};
        for (int p = 0; p < 3; ++p) {
          unsigned& d = wb[((size_t)((nb * K8 + kk / 8) * 3 + p) * 64 + lane) * 4 + (kk & 7) / 2];
          if (kk & 1) d = (d & 0x0000ffffu) | ((unsigned)pc[p] << 16);
          else d = (d & 0xffff0000u) | pc[p];
        }
      }
    }
  std::vector<double> ref(OUT * 32), mag(OUT * 32);
  for (int o = 0; o < OUT; ++o)
    for (int s = 0; s < 32; ++s) {
      double a = 0, m = 0;
      for (int k = 0; k < K; ++k) { a += (double)W[o * K + k] * X[s * K + k]; m += fabs((double)W[o * K + k] * X[s * K + k]); }
      ref[o * 32 + s] = a; mag[o * 32 + s] = m;
    }
  float *dwf, *dx, *dout;
  unsigned* dwb;
  unsigned long long* dcyc;
  CK(hipMalloc(&dwf, wf.size() * 4)); CK(hipMalloc(&dwb, wb.size() * 4)); CK(hipMalloc(&dx, X.size() * 4));
  CK(hipMalloc(&dout, OUT * 32 * 4)); CK(hipMalloc(&dcyc, 8));
  CK(hipMemcpy(dwf, wf.data(), wf.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dwb, wb.data(), wb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dx, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> out(OUT * 32);
  for (int mode = 0; mode < 2; ++mode) {
    const size_t lds = (mode == 0 ? wf.size() : wb.size()) * 4;
    if (mode == 0) hipLaunchKernelGGL(k_layer<0>, dim3(1), dim3(64), lds, 0, dwf, dwb, dx, dout, dcyc, 1);
    else hipLaunchKernelGGL(k_layer<1>, dim3(1), dim3(64), lds, 0, dwf, dwb, dx, dout, dcyc, 1);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), dout, OUT * 32 * 4, hipMemcpyDeviceToHost));
    double worst = 0, worst_abs = 0;
    for (int i = 0; i < OUT * 32; ++i) {
      worst = fmax(worst, fabs(out[i] - ref[i]) / mag[i]);
      worst_abs = fmax(worst_abs, fabs(out[i] - ref[i]) / fmax(fabs(ref[i]), 1e-30));
    }
    printf("%s: max |err| / sum|w x| = %.3e (fp32 eps 5.96e-08), max |err| / |result| = %.3e\n", mode == 0 ? "fp32 32x32x2 " : "bf16x3 6 prod", worst, worst_abs);
    int mismatches = 0;
    std::vector<float> again(OUT * 32);
    for (int rep = 0; rep < 200; ++rep) {   // 8 waves per workgroup on every CU, wave 0 of workgroup 0 writes: same bits every time?
      if (mode == 0) hipLaunchKernelGGL(k_layer<0>, dim3(256), dim3(512), lds, 0, dwf, dwb, dx, dout, dcyc, 1);
      else hipLaunchKernelGGL(k_layer<1>, dim3(256), dim3(512), lds, 0, dwf, dwb, dx, dout, dcyc, 1);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(again.data(), dout, OUT * 32 * 4, hipMemcpyDeviceToHost));
      mismatches += memcmp(again.data(), out.data(), OUT * 32 * 4) != 0;
    }
    printf("               200 repeated launches (8 waves per CU): %d differ from the first\n", mismatches);
  }
  // rate: 8 or 4 waves per workgroup on one CU (2 / 1 per SIMD), one workgroup per CU over the chip
  for (int waves : {4, 8}) {
    for (int mode = 0; mode < 2; ++mode) {
      const size_t lds = (mode == 0 ? wf.size() : wb.size()) * 4;
      const int iters = 2000;
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(k_layer<0>, dim3(256), dim3(64 * waves), lds, 0, dwf, dwb, dx, dout, dcyc, iters);
        else hipLaunchKernelGGL(k_layer<1>, dim3(256), dim3(64 * waves), lds, 0, dwf, dwb, dx, dout, dcyc, iters);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
      }
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long c = 0;
      CK(hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost));
      const double flop = 2.0 * OUT * K * 32 * (double)iters * waves * 256;
      printf("%s  %d waves/SIMD: %7.0f cycles per layer (K = %d -> %d neurons, 32 samples) per wave, %6.1f fp32-equivalent TFLOP/s\n",
             mode == 0 ? "fp32 32x32x2 " : "bf16x3 6 prod", waves / 4, (double)c / iters, K, OUT, flop / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}

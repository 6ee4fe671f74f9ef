// rdrf_common.hpp -- device-side building blocks shared by every kernel of the RoDynRF hot path.
//
// Design (gfx950 / CDNA4, wave64):
//  * MLPs run on the fp32 matrix cores (v_mfma_f32_32x32x2_f32) in the TRANSPOSED form
//        out[neuron][sample] = sum_k W[neuron][k] * in[k][sample]
//    with i = neuron (A operand = weights), j = sample (B operand = activations).  One wave owns a
//    tile of 32 samples; lane l = (h = l>>5, s = l&31) holds HALF of sample s' activation vector.
//  * Canonical activation layout: element e of any per-sample vector lives in lane half
//    h = (e>>2)&1, register slot kk = (e>>3)*4 + (e&3)  (elem_of() below is the inverse).  This is
//    exactly the C/D layout the MFMA produces (row = (r&3) + 8*(r>>2) + 4*(lane>>5)), so the
//    accumulator of one layer is consumed as the B operand of the next with NO transpose and no
//    LDS round trip: k-step kk multiplies elements elem_of(kk,0) / elem_of(kk,1), and the weight
//    columns are permuted to match when they are packed (rdrf_pack.hip).
//  * VM factors are channel-last, so one bilinear tap of 4 components is one 16-byte load.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rodynrf.h"

// compile-time ablation blocks (tools/abl_*.sh, tools/prof_names.py) exist in the tools build only
#ifndef RDRF_TOOLS
#undef RDRF_ABL_NOATOM
#undef RDRF_ABL_NOGLOBAL
#undef RDRF_ABL_NOSCAN
#undef RDRF_ABL_NOLDS
#undef RDRF_ABL_NOGBWD
#undef RDRF_ABL_SC_NLV
#undef RDRF_ABL_SC_NOXY
#undef RDRF_ABL_SC_NOZ
#undef RDRF_ABL_DW_NOLOAD
#undef RDRF_ABL_DW_NOMFMA
#undef RDRF_ABL_DW_NOFLUSH
#undef RDRF_ABL_DW_FULLROW
#undef RDRF_ABL_NOGATHER
#undef RDRF_ABL_OCML_SINCOS
#undef RDRF_ABL_NOSAVE
#undef RDRF_ABL_APP_NOPE
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// waves per persistent workgroup of the fused MLP kernels: 8 -> 2 waves/SIMD with a 256-VGPR
// budget, 4 -> 1 wave/SIMD with 512 VGPRs (A/B-tested on MI355X: 8 wins overall)
#ifndef RDRF_MAXW
#define RDRF_MAXW 8
#endif
#define RDRF_HD __host__ __device__ __forceinline__
#define RDRF_D __device__ __forceinline__

// ---------------------------------------------------------------------------------------------
// gradient accumulation.  Product build: hardware fp32 atomics (the order of the additions, and with it the last
// bits of every gradient, varies from run to run).  Deterministic build (-DRDRF_DETERMINISTIC -> librodynrf_det.so,
// selected with RDRF_DETERMINISTIC=1): every addition into a field's flat gradient buffer goes to a 64-bit FIXED-POINT
// shadow of that buffer (value * 2^40 rounded to an integer; integer addition is associative, so the sum does not
// depend on the order) and rdrf_det_finish folds the shadow into the fp32 gradients once: bit-reproducible gradients
// to diff a suspected race against (SURVEY.md 5 / 7 hard part 1).  Additions outside the bound buffers (pose / ray
// gradients) stay fp32 atomics.
// ---------------------------------------------------------------------------------------------
#ifdef RDRF_DETERMINISTIC
struct DetMap {
  const float* base;
  unsigned long long n;
  unsigned long long* shadow;
};
// one copy per translation unit, written only from the host (rdrf_det_bind): `volatile` keeps the optimiser from
// treating the never-stored-to internal global as a zero constant and folding every lookup away
static __device__ volatile DetMap g_det[2];
#define RDRF_DET_SCALE 1099511627776.0f   // 2^40: quantum 9.1e-13, range +-8.4e6
RDRF_D void grad_add(float* p, float v) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float* base = g_det[k].base;
    unsigned long long* shadow = g_det[k].shadow;
    const unsigned long long n = g_det[k].n;
    if (shadow != nullptr && p >= base && (unsigned long long)(p - base) < n) {
      atomicAdd(shadow + (p - base), (unsigned long long)__float2ll_rn(v * RDRF_DET_SCALE));
      return;
    }
  }
  atomicAdd(p, v);
}
#else
RDRF_D void grad_add(float* p, float v) { atomicAdd(p, v); }
#endif

// ---------------------------------------------------------------------------------------------
// canonical layout
// ---------------------------------------------------------------------------------------------
RDRF_HD int elem_of(int kk, int h) { return ((kk >> 2) << 3) + (h << 2) + (kk & 3); }

// ---------------------------------------------------------------------------------------------
// input segments of the first layers: element index (our order) -> column of the reference weight
// matrix (or -1 = structural zero).  Shared by the pack kernel and documented in DESIGN.md.
// X0 = [xn0,xn1,xn2,t | 15 quads (sin q_2k, cos q_2k, sin q_2k+1, cos q_2k+1)], q_j = xn[j/10]*2^(j%10)
// X1 = 8 pairs (sin t2^f, cos t2^f)
// ---------------------------------------------------------------------------------------------
enum SegId {
  SEG_IDENT = 0,     // hidden layer / plain feature block: e -> e (bounded by in_dim)
  SEG_WARP3_X0,      // layer3 (64,93): [xn3, sin30, cos30, tout30]
  SEG_WARP3_T,       // tout block -> cols 63..92
  SEG_DEN1_X0,       // density/blending layer1 (64,152): [feat72, xn3, sin30, cos30, t, sin8, cos8]
  SEG_DEN1_X1,
  SEG_RGB1_F,        // dyn rgb layer1 (128,107): [feat27, xn3, sin30, cos30, t, sin8, cos8]
  SEG_RGB1_X0,
  SEG_RGB1_X1,
  SEG_STAT1_F_FEA,   // static MLP_Fea layer1 (128,138): [feat27, view3, sin54, cos54]
  SEG_STAT1_P_FEA,
  SEG_STAT1_F_TE,    // static MLP_Fea_TimeEmbedding layer1 (128,135): [feat27, sin54, cos54]
  SEG_STAT1_P_TE,
  SEG_SF_X,          // scene flow layer0 (64,36): [xn3, sin12, cos12, t, sin4, cos4]
  SEG_IDENT72,       // the 72 VM features of density/blending layer1 (cols 0..71)
  SEG_VIEW3,         // view-direction columns 128..130 of a (3,131) output layer
  SEG16_APP_G,       // 16-sample layout: the 72 gathered appearance components, slot (j, c) of lane group g (app16_quad)
  SEG_COUNT
};

RDRF_HD int x0_col(int e, int base_xn, int base_sin, int base_cos, int col_t) {
  if (e < 3) return base_xn + e;
  if (e == 3) return col_t;
  int k = (e - 4) >> 2, m = (e - 4) & 3;
  int j = 2 * k + (m >> 1);
  if (j >= 30) return -1;
  return ((m & 1) ? base_cos : base_sin) + j;
}

RDRF_HD int seg_imap(int seg, int e, int in_dim) {
  switch (seg) {
    case SEG_IDENT: return e < in_dim ? e : -1;
    case SEG_IDENT72: return e < 72 ? e : -1;
    case SEG_VIEW3: return e < 3 ? 128 + e : -1;
    case SEG_WARP3_X0: return e < 64 ? x0_col(e, 0, 3, 33, -1) : -1;
    case SEG_WARP3_T: return e < 30 ? 63 + e : -1;
    case SEG_DEN1_X0: return e < 64 ? x0_col(e, 72, 75, 105, 135) : -1;
    case SEG_DEN1_X1: return e < 16 ? ((e & 1) ? 144 : 136) + (e >> 1) : -1;
    case SEG_RGB1_F: return e < 27 ? e : -1;
    case SEG_RGB1_X0: return e < 64 ? x0_col(e, 27, 30, 60, 90) : -1;
    case SEG_RGB1_X1: return e < 16 ? ((e & 1) ? 99 : 91) + (e >> 1) : -1;
    case SEG_STAT1_F_FEA: return e < 30 ? e : -1;
    case SEG_STAT1_F_TE: return e < 27 ? e : -1;
    case SEG_STAT1_P_FEA:
    case SEG_STAT1_P_TE: {
      // PE block: lane half h, slot 4r+m holds (sin f, cos f, sin 2f, cos 2f)[m] of feature
      // w = elem_of(r, h);  element index e = elem_of(4r+m, h) = 8r + 4h + m
      if (e >= 128) return -1;
      int r = e >> 3, h = (e >> 2) & 1, m = e & 3;
      int w = elem_of(r, h);
      if (w >= 27) return -1;
      int base = (seg == SEG_STAT1_P_FEA) ? 30 : 27;
      // reference order: sin(feat*2^k) at d*2+k, then cos
      return base + ((m & 1) ? 54 : 0) + w * 2 + (m >> 1);
    }
    case SEG_SF_X: {
      if (e < 3) return e;
      if (e == 3) return 27;
      int p = 2 * ((e - 4) >> 2) + (((e - 4) & 3) >> 1);  // pair index
      int c = (e - 4) & 1;
      if (p < 12) return (c ? 15 : 3) + p;
      if (p < 16) return (c ? 32 : 28) + (p - 12);
      return -1;
    }
  }
  return -1;
}

// ---------------------------------------------------------------------------------------------
// 16-sample tiles on v_mfma_f32_16x16x4_f32 (round 6, k_static_app16): lane l = (g = l>>4, s = l&15) holds a QUARTER of
// sample s' vector; element e lives in lane group g = (e>>2)&3, slot kk = (e>>4)*4 + (e&3) -- the C/D layout of the
// 16x16x4 MFMA in the transposed product (row i = 4*(lane>>4) + r of block nb = neuron nb*16 + 4g + r), so that an
// accumulator is again the B operand of the next layer.  Half the activation registers of the 32-sample layout.
// ---------------------------------------------------------------------------------------------
RDRF_HD int elem16_of(int kk, int g) { return ((kk >> 2) << 4) + (g << 2) + (kk & 3); }
// quad of a {48,12,12}-component set that lane group g gathers into slots 4 j .. 4 j + 3 (16-sample layout): j < 3: XY quad
// 4 j + g; j = 3: XZ quad g; j = 4: YZ quad g (groups 0..2; group 3 holds zeros there) -- every lane of the wave reads the
// SAME plane at step j, so no per-lane select between two planes' pointers and strides (which costs 14 VGPRs, spilled).
// Returned in the natural quad order (0..11 XY, 12..14 XZ, 15..17 YZ); -1 = padding.
RDRF_HD int app16_quad(int j, int g) { return j < 3 ? 4 * j + g : (g < 3 ? (j == 3 ? 12 : 15) + g : -1); }
// k-step kk of lane group g -> column of the reference weight matrix for the first-layer segments of the static head
RDRF_HD int seg_imap16(int seg, int kk, int g, int in_dim) {
  if (seg == SEG_STAT1_P_FEA || seg == SEG_STAT1_P_TE) {
    // PE block: slot 4r+m of group g holds (sin f, cos f, sin 2f, cos 2f)[m] of feature w = elem16_of(r, g)
    const int r = kk >> 2, m = kk & 3, w = elem16_of(r, g);
    if (r >= 8 || w >= 27) return -1;
    const int base = (seg == SEG_STAT1_P_FEA) ? 30 : 27;
    return base + ((m & 1) ? 54 : 0) + w * 2 + (m >> 1);   // reference order: sin(feat*2^k) at d*2+k, then cos
  }
  if (seg == SEG16_APP_G) {
    const int w = (kk >> 2) < 5 ? app16_quad(kk >> 2, g) : -1;
    return w < 0 ? -1 : 4 * w + (kk & 3);
  }
  return seg_imap(seg, elem16_of(kk, g), in_dim);
}

// ---------------------------------------------------------------------------------------------
// packed-weight addressing.  MFMA segment: [NBO][KK/4][64 lanes][4];  small (VALU) layer:
// [OUT][2 halves][KK].
// ---------------------------------------------------------------------------------------------
RDRF_HD int pk_mfma_size(int nbo, int kk) { return nbo * kk * 64; }
RDRF_HD int pk_small_size(int out, int kk) { return out * 2 * kk; }

// ---------------------------------------------------------------------------------------------
// MFMA layer segment: acc[nb] += W[nb-block][seg cols] * in
// ---------------------------------------------------------------------------------------------
template <int NBO, int KK>
RDRF_D void mfma_seg(f32x16 (&acc)[NBO], const float (&in)[KK], const float* __restrict__ wp,
                     int lane) {
  static_assert(KK % 4 == 0, "segments are octet padded");
  constexpr int K4 = KK / 4;
  // explicit 1-deep software pipeline on the weight fetch: the scheduling barriers stop hipcc
  // from hoisting every load of the (fully unrolled) layer to the top and spilling.
  f32x4 wc[NBO], wn[NBO];
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) wc[nb] = *(const f32x4*)(wp + (((nb * K4) * 64 + lane) << 2));
#ifdef RDRF_MFMA_PRIO
  __builtin_amdgcn_s_setprio(RDRF_MFMA_PRIO);   // experiment: the wave in its MFMA chain wins the issue arbitration
#endif
#pragma unroll
  for (int k4 = 0; k4 < K4; ++k4) {
    if (k4 + 1 < K4) {
#pragma unroll
      for (int nb = 0; nb < NBO; ++nb)
        wn[nb] = *(const f32x4*)(wp + (((nb * K4 + k4 + 1) * 64 + lane) << 2));
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
#ifdef RDRF_MFMA_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

// ---------------------------------------------------------------------------------------------
// fp32-grade layer segment on the BF16 matrix pipe (round 6).  x = xh + xm + xl: three bf16 pieces, 8 + 8 + 8 significand
// bits -- the whole fp32 significand, exact by truncation -- and
//     W x  ~=  Wl xh + Wh xl + Wm xm + Wm xh + Wh xm + Wh xh           (the six products above 2^-24 relative)
// on v_mfma_f32_32x32x16_bf16: 32 cycles per SIMD for K = 16, i.e. 192 cycles for the six against 8 x 64 = 512 for the same K on
// v_mfma_f32_32x32x2_f32.  Activations are split in registers right before their MFMAs -- 44 VALU instructions per K = 16
// step that issue in the gaps of the twelve MFMAs of the step (tools/micro/bf16x3_layer.hip: 4164 cycles per 160 -> 64 layer and
// wave against 10 444 for mfma_seg, error 2.1e-7 of sum |w x| against 2.7e-7) --, weights arrive pre-split from the pack
// kernel: [NBO][KK/8][3 pieces][64 lanes][4 dwords] = one ds_read_b128 per piece (1.5 x the fp32 bytes, which is why only the
// layers whose image still fits the 160 KB LDS take this form).  The C/D layout of the instruction is that of 32x32x2, and a
// lane's eight consecutive `in` slots are its eight K values of the step, so the canonical activation layout is unchanged.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
RDRF_HD int pk_b3_size(int nbo, int kk) { return nbo * kk * 96; }
// hi = the top 16 bits of x; r = x - hi (exact); mid = the top 16 bits of r; lo = r - mid (the instruction reads its top 16 bits)
RDRF_D void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = __float_as_uint(x) & 0xffff0000u;
  const float r = x - __uint_as_float(hi);
  mid = __float_as_uint(r) & 0xffff0000u;
  lo = __float_as_uint(r - __uint_as_float(mid));
}
RDRF_D unsigned pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }   // (a >> 16) | (b & 0xffff0000)
template <int NBO, int KK>
RDRF_D void mfma_seg_b3(f32x16 (&acc)[NBO], const float (&in)[KK], const float* __restrict__ wpf, int lane) {
  static_assert(KK % 8 == 0, "one K = 16 step takes eight slots per lane half");
  constexpr int K8 = KK / 8;
  const unsigned* __restrict__ wp = reinterpret_cast<const unsigned*>(wpf);
#pragma unroll
  for (int k8 = 0; k8 < K8; ++k8) {
    // the step's weight pieces are requested first: the ~44 VALU instructions of the split below cover the LDS latency, so no
    // step-ahead copy of the fragments is kept (24 registers at NBO = 2).  The scheduling barriers stop hipcc from hoisting
    // every split of the fully unrolled layer to the top (148 spilled registers in the microbenchmark).
    // NO inline asm in this function: hipcc counts an `asm` statement as an instruction when it pads the VALU-write ->
    // MFMA-read wait states, an empty one (used to defeat common-subexpression elimination) left the last piece register
    // short of them -- one sample in ~15 000 came out with a stale lo piece (tools/graph/det_fwd.py).
    u32x4 wc[NBO][3];
#pragma unroll
    for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
      for (int p = 0; p < 3; ++p) wc[nb][p] = *(const u32x4*)(wp + ((size_t)((nb * K8 + k8) * 3 + p) * 64 + lane) * 4);
    // The loads stay FIRST after the previous step's MFMAs: the split below reuses the registers of the previous step's pieces,
    // and a v_sub_f32 that overwrote the B operand two instructions after the last MFMA of the step (the fused render kernel's
    // schedule) gave one ray in ~700 a stale piece, run to run -- a write-after-read window on the 4-register operands of
    // v_mfma_f32_32x32x16_bf16 that hipcc does not pad.  3 NBO ds_read_b128 issue slots sit between the two now.
    __builtin_amdgcn_sched_barrier(0);
    // Split in three sweeps -- hi pieces, mid pieces, lo pieces -- and consume them in that order: the MFMAs that read a
    // piece then sit as far behind the VALU instructions that wrote it as the step allows.  With the lo pieces packed last
    // and read by the THIRD MFMA of the step (small terms first), the fused render kernel returned a stale lo piece for one
    // ray in a few hundred, run to run (tools/graph/fused_diff.py; a full s_waitcnt before the MFMAs did not change it, the
    // order below did: 80 of 80 runs identical): independent MFMAs issue a few cycles apart, and hipcc's wait-state
    // accounting between a v_perm_b32 and the MFMA that reads its result does not hold across them.
    unsigned hi[8], r1[8], mid[8];
    u32x4 bh, bm, bl;
#pragma unroll
    for (int e = 0; e < 8; ++e) hi[e] = __float_as_uint(in[k8 * 8 + e]) & 0xffff0000u;
#pragma unroll
    for (int q = 0; q < 4; ++q) bh[q] = pack_hi16(hi[2 * q], hi[2 * q + 1]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      r1[e] = __float_as_uint(in[k8 * 8 + e] - __uint_as_float(hi[e]));   // exact
      mid[e] = r1[e] & 0xffff0000u;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) bm[q] = pack_hi16(mid[2 * q], mid[2 * q + 1]);
#pragma unroll
    for (int q = 0; q < 4; ++q)   // the instruction reads the top 16 bits of r1 - mid
      bl[q] = pack_hi16(__float_as_uint(__uint_as_float(r1[2 * q]) - __uint_as_float(mid[2 * q])),
                        __float_as_uint(__uint_as_float(r1[2 * q + 1]) - __uint_as_float(mid[2 * q + 1])));
    const bf16x8 xh = __builtin_bit_cast(bf16x8, bh), xm = __builtin_bit_cast(bf16x8, bm), xl = __builtin_bit_cast(bf16x8, bl);
    __builtin_amdgcn_sched_barrier(0);
    // consecutive MFMAs alternate between the accumulators; pieces in the order they were produced (hi, mid, lo last)
#define RDRF_B3_STEP(WP, XP)                                                                                              \
    _Pragma("unroll") for (int nb = 0; nb < NBO; ++nb)                                                                    \
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wc[nb][WP]), XP, acc[nb], 0, 0, 0);
    RDRF_B3_STEP(0, xh) RDRF_B3_STEP(1, xh) RDRF_B3_STEP(2, xh) RDRF_B3_STEP(0, xm) RDRF_B3_STEP(1, xm) RDRF_B3_STEP(0, xl)
#undef RDRF_B3_STEP
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Two groups of output blocks that consume ONE input vector (the backward of a layer with two input segments: d(features) and
// d(X0) from the same dz): the split of a step is done once, group A's MFMAs run while group B's weight pieces arrive.
// Same rules as mfma_seg_b3: loads first, pieces produced and consumed in the order hi, mid, lo, no inline asm.
template <int NA, int NB, int KK>
RDRF_D void mfma_seg_b3_pair(f32x16 (&accA)[NA], f32x16 (&accB)[NB], const float (&in)[KK], const float* __restrict__ wpfA,
                             const float* __restrict__ wpfB, int lane) {
  static_assert(KK % 8 == 0, "one K = 16 step takes eight slots per lane half");
  constexpr int K8 = KK / 8;
  const unsigned* __restrict__ wpA = reinterpret_cast<const unsigned*>(wpfA);
  const unsigned* __restrict__ wpB = reinterpret_cast<const unsigned*>(wpfB);
#pragma unroll
  for (int k8 = 0; k8 < K8; ++k8) {
    u32x4 wa[NA][3], wb[NB][3];
#pragma unroll
    for (int nb = 0; nb < NA; ++nb)
#pragma unroll
      for (int p = 0; p < 3; ++p) wa[nb][p] = *(const u32x4*)(wpA + ((size_t)((nb * K8 + k8) * 3 + p) * 64 + lane) * 4);
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
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int p = 0; p < 3; ++p) wb[nb][p] = *(const u32x4*)(wpB + ((size_t)((nb * K8 + k8) * 3 + p) * 64 + lane) * 4);
#define RDRF_B3_STEP(ACC, W, N, WP, XP)                                                                                   \
    _Pragma("unroll") for (int nb = 0; nb < N; ++nb)                                                                      \
      ACC[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, W[nb][WP]), XP, ACC[nb], 0, 0, 0);
    RDRF_B3_STEP(accA, wa, NA, 0, xh) RDRF_B3_STEP(accA, wa, NA, 1, xh) RDRF_B3_STEP(accA, wa, NA, 2, xh)
    RDRF_B3_STEP(accA, wa, NA, 0, xm) RDRF_B3_STEP(accA, wa, NA, 1, xm) RDRF_B3_STEP(accA, wa, NA, 0, xl)
    __builtin_amdgcn_sched_barrier(0);
    RDRF_B3_STEP(accB, wb, NB, 0, xh) RDRF_B3_STEP(accB, wb, NB, 1, xh) RDRF_B3_STEP(accB, wb, NB, 2, xh)
    RDRF_B3_STEP(accB, wb, NB, 0, xm) RDRF_B3_STEP(accB, wb, NB, 1, xm) RDRF_B3_STEP(accB, wb, NB, 0, xl)
#undef RDRF_B3_STEP
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 x 3 with SPLIT STORAGE (round 6): the appearance kernels' images fill the LDS in fp32 already, so their weights cannot
// grow 1.5 x.  The hi and mid pieces (4 bytes per weight -- the fp32 footprint) stay in the LDS image,
// [NBO][KK/8][2 pieces][64 lanes][4 dwords]; the lo pieces -- read by ONE of the six products of a step -- are streamed from
// the pack buffer in global memory, [NBO][KK/8][64 lanes][4 dwords] (L2-resident: every wave of the launch reads the same
// 70 KB), one global_load_dwordx4 per output block and step, requested one step ahead: `lo` holds the pieces of this call's
// step 0 on entry and those of `glo_next`'s step 0 on return (the stream runs across the segments and layers of a tile).
// Same piece order (hi, mid, lo), same no-asm / loads-first rules as mfma_seg_b3.
// ---------------------------------------------------------------------------------------------
RDRF_HD int pk_b3s_size(int nbo, int kk) { return nbo * kk * 64; }      // LDS dwords (hi + mid)
RDRF_HD int pk_b3s_lo_size(int nbo, int kk) { return nbo * kk * 32; }   // global dwords (lo)
// The stream is read with buffer loads: one scalar descriptor for the whole lo region, the lane's byte offset (16 lane) in ONE
// register, the fragment's offset as the scalar operand -- with flat 64-bit addresses hipcc kept a register pair per
// fragment address alive across the tile loop (96 spilled registers in k_static_app).
struct B3sLo {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff;   // 16 * lane
};
RDRF_D B3sLo b3s_lo_stream(const float* __restrict__ base, int lane) {
  return B3sLo{__builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000), (unsigned)lane * 16u};
}
// fragment (nb, k8) of the lo image that starts `off` dwords into the region
RDRF_D u32x4 b3s_lo_frag(const B3sLo& st, int off, int k8n, int nb, int k8) {
  return __builtin_amdgcn_raw_buffer_load_b128(st.rsrc, st.voff, (off + (nb * k8n + k8) * 256) * 4, 0);
}
template <int NBO>
RDRF_D void b3s_lo_load(u32x4 (&lo)[NBO], const B3sLo& st, int off, int k8n, int k8) {
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) lo[nb] = b3s_lo_frag(st, off, k8n, nb, k8);
}
template <int NBO, int KK, int KK_NEXT>
RDRF_D void mfma_seg_b3s(f32x16 (&acc)[NBO], const float (&in)[KK], const float* __restrict__ wpf, const B3sLo& st, int off,
                         int off_next, u32x4 (&lo)[NBO], int lane) {
  static_assert(KK % 8 == 0, "one K = 16 step takes eight slots per lane half");
  static_assert(NBO >= 2, "output blocks are processed in two groups");
  constexpr int K8 = KK / 8, NA = NBO / 2, NB = NBO - NA;
  // The step follows mfma_seg_b3_pair (the form the reproducibility tests pin): group A's hi / mid fragments are requested
  // FIRST -- with the lo requests that end the previous step, at least six load issue slots separate the previous step's
  // last MFMA from the VALU instructions of the split, which reuse its operand registers (narrow layers request group B's
  // fragments there too) --, the split is done once, group B's fragments arrive while group A's MFMAs run.  The product
  // that reads the streamed lo pieces comes LAST in its group: a whole step lies between their request and their use.
  constexpr bool EARLY_B = NA * 2 + NBO < 6;
  const unsigned* __restrict__ wp = reinterpret_cast<const unsigned*>(wpf);
#pragma unroll
  for (int k8 = 0; k8 < K8; ++k8) {
    u32x4 wa[NA][2], wb[NB][2];
#pragma unroll
    for (int nb = 0; nb < NA; ++nb)
#pragma unroll
      for (int p = 0; p < 2; ++p) wa[nb][p] = *(const u32x4*)(wp + ((size_t)((nb * K8 + k8) * 2 + p) * 64 + lane) * 4);
    if (EARLY_B) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int p = 0; p < 2; ++p) wb[nb][p] = *(const u32x4*)(wp + ((size_t)(((NA + nb) * K8 + k8) * 2 + p) * 64 + lane) * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned hi[8], r1[8], mid[8];
    u32x4 bh, bm, bl;
#pragma unroll
    for (int e = 0; e < 8; ++e) hi[e] = __float_as_uint(in[k8 * 8 + e]) & 0xffff0000u;
#pragma unroll
    for (int q = 0; q < 4; ++q) bh[q] = pack_hi16(hi[2 * q], hi[2 * q + 1]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      r1[e] = __float_as_uint(in[k8 * 8 + e] - __uint_as_float(hi[e]));   // exact
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
    if (!EARLY_B) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int p = 0; p < 2; ++p) wb[nb][p] = *(const u32x4*)(wp + ((size_t)(((NA + nb) * K8 + k8) * 2 + p) * 64 + lane) * 4);
    }
#define RDRF_B3S_STEP(N, NB0, WV, XP)                                                                                     \
    _Pragma("unroll") for (int nb = 0; nb < N; ++nb)                                                                      \
      acc[NB0 + nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WV), XP, acc[NB0 + nb], 0, 0, 0);
    RDRF_B3S_STEP(NA, 0, wa[nb][0], xh) RDRF_B3S_STEP(NA, 0, wa[nb][1], xh) RDRF_B3S_STEP(NA, 0, wa[nb][0], xm)
    RDRF_B3S_STEP(NA, 0, wa[nb][1], xm) RDRF_B3S_STEP(NA, 0, wa[nb][0], xl) RDRF_B3S_STEP(NA, 0, lo[nb], xh)
    __builtin_amdgcn_sched_barrier(0);
    RDRF_B3S_STEP(NB, NA, wb[nb][0], xh) RDRF_B3S_STEP(NB, NA, wb[nb][1], xh) RDRF_B3S_STEP(NB, NA, wb[nb][0], xm)
    RDRF_B3S_STEP(NB, NA, wb[nb][1], xm) RDRF_B3S_STEP(NB, NA, wb[nb][0], xl) RDRF_B3S_STEP(NB, NA, lo[NA + nb], xh)
#undef RDRF_B3S_STEP
    __builtin_amdgcn_sched_barrier(0);
    // the lo pieces are consumed: request the next step's (this segment's, or step 0 of the segment that follows in the tile)
    if (k8 + 1 < K8) b3s_lo_load<NBO>(lo, st, off, K8, k8 + 1);
    else if (KK_NEXT > 0) b3s_lo_load<NBO>(lo, st, off_next, KK_NEXT / 8, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// accumulator init from a PACKED bias ([2 halves][NBO*16] in canonical order; nullptr = zero)
template <int NBO>
RDRF_D void acc_bias(f32x16 (&acc)[NBO], const float* __restrict__ bpk, int h) {
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) {
    // one 64-byte vector load per block: the four ds_read_b128 write the accumulator's registers directly (element-wise
    // inserts of four f32x4 compiled to 16 v_mov per block and layer -- and a VALU instruction costs its four cycles of the
    // SIMD whatever the matrix pipe does, tools/micro/mfma_valu_overlap.hip)
    if (bpk != nullptr) acc[nb] = *(const f32x16*)(bpk + h * (NBO * 16) + nb * 16);
    else acc[nb] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  }
}

// cooperative copy of one kernel's weight image into LDS
// (121-159 KB per workgroup: 15-20 rounds of one 16-byte load per thread at 8 waves; issued one at a time each round is
// an L2 round trip, ~15 us per launch -- a quarter of a 512-ray eval chunk's kernel -- so four are kept in flight)
RDRF_D void lds_fill(float* __restrict__ lds, const float* __restrict__ src, int nfloats) {
  const int stride = blockDim.x * 4;
  int i = threadIdx.x * 4;
  for (; i + 3 * stride < nfloats; i += 4 * stride) {
    const f32x4 v0 = *(const f32x4*)(src + i);
    const f32x4 v1 = *(const f32x4*)(src + i + stride);
    const f32x4 v2 = *(const f32x4*)(src + i + 2 * stride);
    const f32x4 v3 = *(const f32x4*)(src + i + 3 * stride);
    *(f32x4*)(lds + i) = v0;
    *(f32x4*)(lds + i + stride) = v1;
    *(f32x4*)(lds + i + 2 * stride) = v2;
    *(f32x4*)(lds + i + 3 * stride) = v3;
  }
  for (; i < nfloats; i += stride) *(f32x4*)(lds + i) = *(const f32x4*)(src + i);
  __syncthreads();
}

// max(x, 0) in ONE instruction.  fmaxf(x, 0.0f) compiles to TWO: `v_max_f32 t, x, x` (LLVM canonicalises the operand because
// it cannot prove the MFMA result is not a signalling NaN) and `v_max_f32 out, 0, t` -- 160 extra VALU instructions per
// 32-sample tile of k_dyn_density, and VALU time adds to MFMA time on a SIMD (tools/micro/mfma_valu_overlap.hip).  The
// hardware instruction already implements maxNum (a NaN operand yields the other one), i.e. fmaxf's result bit for bit.
RDRF_D float relu1(float x) {
  float o;
  asm("v_max_f32 %0, 0, %1" : "=v"(o) : "v"(x));
  return o;
}
template <int NBO>
RDRF_D void acc_relu(float (&out)[NBO * 16], const f32x16 (&acc)[NBO]) {
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[nb * 16 + r] = relu1(acc[nb][r]);
}

template <int NBO>
RDRF_D void acc_copy(float (&out)[NBO * 16], const f32x16 (&acc)[NBO]) {
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[nb * 16 + r] = acc[nb][r];
}

// small output layer on the VALU: returns sum_e W[o][e]*in[e] over BOTH lane halves (no bias)
template <int KK>
RDRF_D float dot_small(const float (&in)[KK], const float* __restrict__ ws /* [2][KK] */, int h) {
  const float* w = ws + h * KK;
  float a = 0.f;
#pragma unroll
  for (int k4 = 0; k4 < KK / 4; ++k4) {
    f32x4 v = *(const f32x4*)(w + k4 * 4);
    a = fmaf(v.x, in[k4 * 4 + 0], a);
    a = fmaf(v.y, in[k4 * 4 + 1], a);
    a = fmaf(v.z, in[k4 * 4 + 2], a);
    a = fmaf(v.w, in[k4 * 4 + 3], a);
  }
  return a + __shfl_xor(a, 32, 64);
}

// ---------------------------------------------------------------------------------------------
// coordinates.  Written with contraction OFF so that they round exactly like the reference's
// separate ATen ops (models/tensorBase.py:425-433): valid / app masks depend on these bits.
// ---------------------------------------------------------------------------------------------
struct Box {
  float lo[3], hi[3], inv[3];
};

RDRF_D float norm_c(float x, float lo, float inv) {
#pragma clang fp contract(off)
  float a = x - lo;
  float b = a * inv;
  return b - 1.0f;
}
RDRF_D float unnorm_c(float xn, float lo, float inv) {
#pragma clang fp contract(off)
  float a = xn + 1.0f;
  float b = a / inv;
  return b + lo;
}
RDRF_D float mul_add_nc(float a, float b, float c) {  // a*b + c, two roundings
#pragma clang fp contract(off)
  float m = a * b;
  return m + c;
}

// ---------------------------------------------------------------------------------------------
// VM gather: one quad (4 consecutive components) of plane_tap * line_tap, align_corners=True,
// zero padding, stride level `level` (sub-array plane[::s, ::s], s = 1<<level), following
// models/tensoRF.py:118-154 / 646-723 and ATen's grid_sampler_2d.
// ---------------------------------------------------------------------------------------------
struct Tap1 {
  int i0;
  float w0, w1;
  bool ok0, ok1;
};
template <bool UNI = false>
RDRF_D Tap1 tap1d(float c, int Ls) {
#pragma clang fp contract(off)
  Tap1 t;
  // Ls is wave-uniform and loop-invariant: left alone, the optimiser hoists (float)(Ls-1), (float)Ls+1,
  // (float)(Ls-2) of every (axis, level, factor set) out of the tile loop -- as VECTOR registers (gfx950 has no
  // scalar float ALU) -- and spills them; each reload is a scratch load + `s_waitcnt vmcnt(0)` in the middle of
  // the gather sequence, which also waits for every store and gather in flight.  Opaque here = three v_cvt per call.
  // UNI (the 128-VGPR kernels): Ls is wave-uniform at the call site and stays in a SCALAR register -- with "+v" the
  // v_mov from the SGPR is itself hoisted out of the tile loop and spilled (18 of them in k_static_app16)
  if constexpr (UNI) asm volatile("" : "+s"(Ls));
  else asm volatile("" : "+v"(Ls));   // ("v": Ls differs between the half-waves of some callers)
  float f = ((c + 1.0f) / 2.0f) * (float)(Ls - 1);
  float fl = floorf(f);
  t.w1 = f - fl;
  t.w0 = (fl + 1.0f) - f;
  // clamp before the int conversion so wild coordinates cannot overflow
  // (one v_med3_f32: fminf(fmaxf(..)) compiled to two canonicalising v_max + max + min; same value, NaN -> -2 as before)
  float flc = __builtin_amdgcn_fmed3f(fl, -2.0f, (float)Ls + 1.0f);
  t.i0 = (int)flc;
  t.ok0 = (fl >= 0.0f) && (fl <= (float)(Ls - 1));
  t.ok1 = (fl >= -1.0f) && (fl <= (float)(Ls - 2));
  return t;
}

template <int C0Q, int C1Q>
struct QuadSel {
  int level, pi, q, C;
};
template <int C0Q, int C1Q>
RDRF_D QuadSel<C0Q, C1Q> quad_sel(int g) {
  constexpr int QPL = C0Q + 2 * C1Q;
  QuadSel<C0Q, C1Q> s;
  s.level = g / QPL;
  int w = g - s.level * QPL;
  s.pi = w < C0Q ? 0 : (w < C0Q + C1Q ? 1 : 2);
  s.q = w - (s.pi == 0 ? 0 : (s.pi == 1 ? C0Q : C0Q + C1Q));
  s.C = 4 * (s.pi == 0 ? C0Q : C1Q);
  return s;
}

struct QuadTaps {  // everything the backward needs as well
  f32x4 pv;        // interpolated plane quad
  f32x4 lv;        // interpolated line quad
};

RDRF_D f32x4 ld4(const float* p) { return *(const f32x4*)p; }

template <int C0Q, int C1Q, int ABL = 0>
RDRF_D QuadTaps gather_quad_taps(const RdrfVM& vm, int g, float x0, float x1, float x2) {
  QuadSel<C0Q, C1Q> s = quad_sel<C0Q, C1Q>(g);
  const int pi = s.pi;
  const float cx = pi == 2 ? x1 : x0;
  const float cy = pi == 0 ? x1 : x2;
  const float cl = pi == 0 ? x2 : (pi == 1 ? x1 : x0);
  const float* P = pi == 0 ? vm.plane[0] : (pi == 1 ? vm.plane[1] : vm.plane[2]);
  const float* Lp = pi == 0 ? vm.line[0] : (pi == 1 ? vm.line[1] : vm.line[2]);
  const int H = pi == 0 ? vm.H[0] : (pi == 1 ? vm.H[1] : vm.H[2]);
  const int W = pi == 0 ? vm.W[0] : (pi == 1 ? vm.W[1] : vm.W[2]);
  const int L = pi == 0 ? vm.L[0] : (pi == 1 ? vm.L[1] : vm.L[2]);
  const int sH = pi == 0 ? vm.sH[0] : (pi == 1 ? vm.sH[1] : vm.sH[2]);
  const int sW = pi == 0 ? vm.sW[0] : (pi == 1 ? vm.sW[1] : vm.sW[2]);
  const int lv = s.level, st = 1 << lv;
  const int Ws = (W + st - 1) >> lv, Hs = (H + st - 1) >> lv, Ls = (L + st - 1) >> lv;
  Tap1 tx = tap1d(cx, Ws), ty = tap1d(cy, Hs), tl = tap1d(cl, Ls);
  const int qo = 4 * s.q;
  const int C = s.C;
  QuadTaps r;
  // All six taps are loaded UNCONDITIONALLY from clamped (always valid) addresses and validity is
  // folded into the weights: a predicated tap (`if (ok) acc += load * w`) compiles to a branch +
  // load + wait per tap, i.e. six serialized memory round trips per quad.
  const int x0c = min(max(tx.i0, 0), Ws - 1) << lv, x1c = min(max(tx.i0 + 1, 0), Ws - 1) << lv;
  const int y0c = min(max(ty.i0, 0), Hs - 1) << lv, y1c = min(max(ty.i0 + 1, 0), Hs - 1) << lv;
  const int l0c = min(max(tl.i0, 0), Ls - 1) << lv, l1c = min(max(tl.i0 + 1, 0), Ls - 1) << lv;
  f32x4 v00, v01, v10, v11, a0, a1;
  if constexpr (ABL == 1) {   // the addresses are still formed, nothing is fetched
    const float f0 = (float)(y0c * sH + x0c * sW + qo) * 1e-9f, f1 = (float)(y0c * sH + x1c * sW + qo) * 1e-9f;
    const float f2 = (float)(y1c * sH + x0c * sW + qo) * 1e-9f, f3 = (float)(y1c * sH + x1c * sW + qo) * 1e-9f;
    const float f4 = (float)(l0c * C + qo) * 1e-9f, f5 = (float)(l1c * C + qo) * 1e-9f;
    v00 = f32x4{f0, f1, f2, f3}; v01 = f32x4{f1, f2, f3, f0}; v10 = f32x4{f2, f3, f0, f1}; v11 = f32x4{f3, f0, f1, f2};
    a0 = f32x4{f4, f5, f4, f5}; a1 = f32x4{f5, f4, f5, f4};
  } else {
    v00 = ld4(P + (size_t)(y0c * sH + x0c * sW) + qo);
    v01 = ld4(P + (size_t)(y0c * sH + x1c * sW) + qo);
    v10 = ld4(P + (size_t)(y1c * sH + x0c * sW) + qo);
    v11 = ld4(P + (size_t)(y1c * sH + x1c * sW) + qo);
    a0 = ld4(Lp + (size_t)l0c * C + qo);
    a1 = ld4(Lp + (size_t)l1c * C + qo);
  }
  const float wx0 = tx.ok0 ? tx.w0 : 0.f, wx1 = tx.ok1 ? tx.w1 : 0.f;
  const float wy0 = ty.ok0 ? ty.w0 : 0.f, wy1 = ty.ok1 ? ty.w1 : 0.f;
  const float wl0 = tl.ok0 ? tl.w0 : 0.f, wl1 = tl.ok1 ? tl.w1 : 0.f;
  r.pv = v00 * (wx0 * wy0) + v01 * (wx1 * wy0) + v10 * (wx0 * wy1) + v11 * (wx1 * wy1);
  r.lv = a0 * wl0 + a1 * wl1;
  return r;
}

template <int C0Q, int C1Q, int ABL = 0>
RDRF_D f32x4 gather_quad(const RdrfVM& vm, int g, float x0, float x1, float x2) {
#ifdef RDRF_ABL_NOGATHER
  return f32x4{x0, x1, x2, (float)g};
#endif
  if constexpr (ABL == 2) return f32x4{x0 * 0.01f, x1 * 0.01f, x2 * 0.01f, (float)g * 1e-3f};
  QuadTaps t = gather_quad_taps<C0Q, C1Q, ABL>(vm, g, x0, x1, x2);
  return t.pv * t.lv;
}

// ---------------------------------------------------------------------------------------------
// VM gather with SHARED taps (forward kernels).  The three planes of a factor set are spanned by the same three grid
// axes (plane XY <-> line Z, XZ <-> Y, YZ <-> X; the host checks vm_one_grid), so at one stride level there are only
// three distinct 1-D taps -- one per axis -- whatever the number of planes, quads and factor sets read at that point.
// gather_quad above recomputes all three for EVERY quad (81 tap1d + 162 address computations per lane and tile in
// k_dyn_app: measured by ablation, the tap arithmetic was 17 % of that kernel's time -- more than the waits on its
// loads); here a lane forms the axis taps once per level, the four texel pointers / two line pointers / bilinear weights
// once per (level, plane), and every quad of that plane is six 16-byte loads at immediate offsets + the interpolation.
// Same arithmetic per element (tap1d, clamped addresses, validity folded into the weights,
// (v00 w00 + v01 w01 + v10 w10 + v11 w11) * (a0 wl0 + a1 wl1)) as gather_quad_taps.
// ---------------------------------------------------------------------------------------------
struct AxisTap {
  int i0, i1;     // clamped tap indices in level-0 texels (already << level)
  float w0, w1;   // interpolation weights, 0 where the tap is out of range (zero padding)
};
template <bool UNI = false>
RDRF_D AxisTap axis_tap(float c, int L, int lv) {
  const int st = 1 << lv, Ls = (L + st - 1) >> lv;
  const Tap1 t = tap1d<UNI>(c, Ls);
  AxisTap a;
  a.i0 = min(max(t.i0, 0), Ls - 1) << lv;
  a.i1 = min(max(t.i0 + 1, 0), Ls - 1) << lv;
  a.w0 = t.ok0 ? t.w0 : 0.f;
  a.w1 = t.ok1 ? t.w1 : 0.f;
  return a;
}
struct PlaneTaps {
  const float *p00, *p01, *p10, *p11, *l0, *l1;
  float w00, w01, w10, w11, wl0, wl1;
};
// plane `pi` of `vm` at the taps (ax, ay) of its two axes, its line at `al`; `qoff` floats are added to every pointer
// (the lane's first quad)
RDRF_D PlaneTaps plane_taps(const RdrfVM& vm, int pi, const AxisTap& ax, const AxisTap& ay, const AxisTap& al, int qoff) {
  const float* P = pi == 0 ? vm.plane[0] : (pi == 1 ? vm.plane[1] : vm.plane[2]);
  const float* Lp = pi == 0 ? vm.line[0] : (pi == 1 ? vm.line[1] : vm.line[2]);
  const int sH = pi == 0 ? vm.sH[0] : (pi == 1 ? vm.sH[1] : vm.sH[2]);
  const int sW = pi == 0 ? vm.sW[0] : (pi == 1 ? vm.sW[1] : vm.sW[2]);
  const int C = pi == 0 ? vm.C[0] : (pi == 1 ? vm.C[1] : vm.C[2]);
  PlaneTaps t;
  t.p00 = P + (size_t)(ay.i0 * sH + ax.i0 * sW) + qoff;
  t.p01 = P + (size_t)(ay.i0 * sH + ax.i1 * sW) + qoff;
  t.p10 = P + (size_t)(ay.i1 * sH + ax.i0 * sW) + qoff;
  t.p11 = P + (size_t)(ay.i1 * sH + ax.i1 * sW) + qoff;
  t.l0 = Lp + (size_t)al.i0 * C + qoff;
  t.l1 = Lp + (size_t)al.i1 * C + qoff;
  t.w00 = ax.w0 * ay.w0; t.w01 = ax.w1 * ay.w0; t.w10 = ax.w0 * ay.w1; t.w11 = ax.w1 * ay.w1;
  t.wl0 = al.w0; t.wl1 = al.w1;
  return t;
}
// the three axis taps of the normalised point (x0, x1, x2) at stride level lv (axis sizes: the XY plane and its line)
struct PointTaps { AxisTap x, y, z; };
template <bool UNI = false>
RDRF_D PointTaps point_taps(const RdrfVM& vm, float x0, float x1, float x2, int lv) {
  PointTaps p;
  p.x = axis_tap<UNI>(x0, vm.W[0], lv);
  p.y = axis_tap<UNI>(x1, vm.H[0], lv);
  p.z = axis_tap<UNI>(x2, vm.L[0], lv);
  return p;
}
RDRF_D AxisTap select_axis(bool first, const AxisTap& a, const AxisTap& b) {
  AxisTap t;
  t.i0 = first ? a.i0 : b.i0; t.i1 = first ? a.i1 : b.i1; t.w0 = first ? a.w0 : b.w0; t.w1 = first ? a.w1 : b.w1;
  // opaque scalars: left alone, the SLP vectoriser packs the selected weights into vectors and then shuffles them
  // through SCRATCH memory (32 bytes of stack + a vmcnt wait in the middle of every gather sequence)
  asm volatile("" : "+v"(t.w0), "+v"(t.w1), "+v"(t.i0), "+v"(t.i1));
  return t;
}
// the XZ plane for half 0, the YZ plane for half 1 (lane-dependent plane: the INPUTS of plane_taps are selected -- a
// select between two finished PlaneTaps structs was lowered through scratch memory)
RDRF_D PlaneTaps plane_taps_xz_or_yz(const RdrfVM& vm, const PointTaps& pt, int h, int qoff) {
  const bool yz = h != 0;
  return plane_taps(vm, yz ? 2 : 1, select_axis(yz, pt.y, pt.x), pt.z, select_axis(yz, pt.x, pt.y), qoff);
}
RDRF_D void shift_taps(PlaneTaps& t, int off) {
  t.p00 += off; t.p01 += off; t.p10 += off; t.p11 += off; t.l0 += off; t.l1 += off;
}
// one quad of plane_tap * line_tap at float offset `off` (a compile-time constant at the call sites: an immediate)
RDRF_D f32x4 taps_combine(const PlaneTaps& t, f32x4 v00, f32x4 v01, f32x4 v10, f32x4 v11, f32x4 a0, f32x4 a1) {
  const f32x4 pv = v00 * t.w00 + v01 * t.w01 + v10 * t.w10 + v11 * t.w11;
  const f32x4 lv = a0 * t.wl0 + a1 * t.wl1;
  return pv * lv;
}
RDRF_D f32x4 taps_quad(const PlaneTaps& t, int off) {
  return taps_combine(t, ld4(t.p00 + off), ld4(t.p01 + off), ld4(t.p10 + off), ld4(t.p11 + off), ld4(t.l0 + off), ld4(t.l1 + off));
}
// NQ quads of one plane, `stride` floats apart: ALL 6 NQ loads are issued before the first one is used.  Left to itself hipcc
// schedules the gathers for register pressure -- four to six loads, s_waitcnt vmcnt(0), interpolate, the next quad: ~30
// dependent memory round trips per 32-sample tile of k_dyn_app, ~20 us of the ~40 us a wave spends outside its MFMA
// chains per tile, which two waves per SIMD cannot cover (matrix pipe 60 % busy).  One batch = one round trip.
template <int NQ>
RDRF_D void taps_quads(const PlaneTaps& t, int stride, f32x4 (&out)[NQ]) {
  f32x4 v00[NQ], v01[NQ], v10[NQ], v11[NQ], a0[NQ], a1[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    v00[q] = ld4(t.p00 + q * stride); v01[q] = ld4(t.p01 + q * stride);
    v10[q] = ld4(t.p10 + q * stride); v11[q] = ld4(t.p11 + q * stride);
    a0[q] = ld4(t.l0 + q * stride); a1[q] = ld4(t.l1 + q * stride);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < NQ; ++q) out[q] = taps_combine(t, v00[q], v01[q], v10[q], v11[q], a0[q], a1[q]);
}
// one quad of each of three (plane, offset) pairs in one batch
RDRF_D void taps_quads3(const PlaneTaps& ta, const PlaneTaps& tb, const PlaneTaps& tc, f32x4 (&out)[3]) {
  const f32x4 a00 = ld4(ta.p00), a01 = ld4(ta.p01), a10 = ld4(ta.p10), a11 = ld4(ta.p11), al0 = ld4(ta.l0), al1 = ld4(ta.l1);
  const f32x4 b00 = ld4(tb.p00), b01 = ld4(tb.p01), b10 = ld4(tb.p10), b11 = ld4(tb.p11), bl0 = ld4(tb.l0), bl1 = ld4(tb.l1);
  const f32x4 c00 = ld4(tc.p00), c01 = ld4(tc.p01), c10 = ld4(tc.p10), c11 = ld4(tc.p11), cl0 = ld4(tc.l0), cl1 = ld4(tc.l1);
  __builtin_amdgcn_sched_barrier(0);
  out[0] = taps_combine(ta, a00, a01, a10, a11, al0, al1);
  out[1] = taps_combine(tb, b00, b01, b10, b11, bl0, bl1);
  out[2] = taps_combine(tc, c00, c01, c10, c11, cl0, cl1);
}
// half h's nine quads of one stride level of a {48,12,12}-component set, canonical order: out[4 j + c], j = 0..8 = quad
// w = 2 j + h of the level (w < 12: XY quad w, 12..14: XZ quad w - 12, 15..17: YZ quad w - 15)
template <int OFF, int NOUT>
RDRF_D void gather_level_app(const RdrfVM& vm, const PointTaps& pt, int h, float (&out_)[NOUT]) {
  {
    const PlaneTaps xy = plane_taps(vm, 0, pt.x, pt.y, pt.z, 4 * h);
    f32x4 v[6];
    taps_quads<6>(xy, 8, v);   // 36 loads in flight
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      out_[OFF + 4 * j + 0] = v[j].x; out_[OFF + 4 * j + 1] = v[j].y; out_[OFF + 4 * j + 2] = v[j].z; out_[OFF + 4 * j + 3] = v[j].w;
    }
  }
  // j = 6: XZ quad h;  j = 7: h ? YZ quad 0 : XZ quad 2;  j = 8: YZ quad 1 + h   (lane-dependent quads: pointer shifts)
  PlaneTaps t6 = plane_taps(vm, 1, pt.x, pt.z, pt.y, 0), t8 = plane_taps(vm, 2, pt.y, pt.z, pt.x, 0);
  const PlaneTaps t7 = plane_taps_xz_or_yz(vm, pt, h, h ? 0 : 8);
  shift_taps(t6, 4 * h);
  shift_taps(t8, 4 + 4 * h);
  f32x4 w[3];
  taps_quads3(t6, t7, t8, w);   // 18 loads in flight
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    out_[OFF + 24 + 4 * j + 0] = w[j].x; out_[OFF + 24 + 4 * j + 1] = w[j].y; out_[OFF + 24 + 4 * j + 2] = w[j].z; out_[OFF + 24 + 4 * j + 3] = w[j].w;
  }
}
// half h's three quads of one stride level of a {16,4,4}-component set: out[4 j + c], j = 0..2 = quad w = 2 j + h
// (w < 4: XY quad w, 4: XZ, 5: YZ)
template <int OFF, int NOUT>
RDRF_D void gather_level_den(const RdrfVM& vm, const PointTaps& pt, int h, float (&out)[NOUT]) {
  const PlaneTaps xy = plane_taps(vm, 0, pt.x, pt.y, pt.z, 4 * h);
  PlaneTaps xy2 = xy;
  shift_taps(xy2, 8);
  const PlaneTaps t2 = plane_taps_xz_or_yz(vm, pt, h, 0);
  f32x4 v[3];
  taps_quads3(xy, xy2, t2, v);   // 18 loads in flight
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    out[OFF + 4 * j + 0] = v[j].x; out[OFF + 4 * j + 1] = v[j].y; out[OFF + 4 * j + 2] = v[j].z; out[OFF + 4 * j + 3] = v[j].w;
  }
}

// ---------------------------------------------------------------------------------------------
// shared input blocks X0 (xn, t, PE10(xn)) and X1 (PE8(t)) in canonical layout
// (models/tensorBase.py:13-19 positional_encoding: q[d*F+k] = p[d]*2^k; [sin(q), cos(q)])
// ---------------------------------------------------------------------------------------------
// sin / cos of a positional-encoding argument: two-constant Cody-Waite reduction by pi/2 (exact products inside
// the fma, so the reduced argument carries one rounding for |a| <= 1e5: n <= 6.4e4 and n * (pi/2 - C1 - C2) < 4e-10),
// then the Cephes single-precision minimax polynomials on [-pi/4, pi/4].  Max abs error 9.2e-8 over the encodings'
// argument range (libm: 7e-8; tests/test_abi_cpu.py::test_sincos_pe_formula restates and bounds it) in ~24 VALU
// instructions; OCML's sincosf costs ~40 plus a large-argument branch, and the encodings were 28 % of
// k_dyn_density's vector instructions.  Callers route |a| > 1e5 (wild feature-mode coordinates) to OCML.
#define RDRF_PE_FAST_MAX 1.0e5f
RDRF_D void sincos_pe(float a, float& s, float& c) {
  const float n = rintf(a * 0.63661977236758134f);
  float r = fmaf(n, -1.57079625129699707031e+00f, a);
  r = fmaf(n, -7.54978941586159635335e-08f, r);
  const int q = (int)n;
  const float r2 = r * r;
  float sp = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  sp = fmaf(sp, r2, -1.6666654611e-1f);
  const float sr = fmaf(sp * r2, r, r);
  float cp = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  cp = fmaf(cp, r2, 4.166664568298827e-2f);
  const float cr = fmaf(cp * r2, r2, fmaf(r2, -0.5f, 1.0f));
  const bool sw = (q & 1) != 0;
  const float S = sw ? cr : sr, C = sw ? sr : cr;
  s = __int_as_float(__float_as_int(S) ^ ((q & 2) << 30));
  c = __int_as_float(__float_as_int(C) ^ (((q + 1) & 2) << 30));
}
template <bool FAST>
RDRF_D void sincos_sel(float a, float& s, float& c) {
#ifdef RDRF_ABL_OCML_SINCOS
  sincosf(a, &s, &c);
#else
  if constexpr (FAST) sincos_pe(a, s, c);
  else sincosf(a, &s, &c);
#endif
}

// (sin, cos) of the doubled angle from (sin, cos) of the angle: sin 2a = 2 s c, cos 2a = 1 - 2 s^2 -- three VALU
// instructions instead of the ~26 of sincos_pe.  Round 6: fp32 MFMAs and VALU instructions of the waves of a SIMD do not
// overlap (tools/micro/mfma_valu_overlap.hip: the times ADD), so every VALU instruction of the MLP kernels costs its four
// cycles of the step; the encodings are their largest VALU item.  Used for the static head's feature encoding (sin 2F, cos 2F)
// only: the absolute error of the doubled pair is <= 2 x that of the exact one + 1.2e-7 (~3e-7 worst,
// tests/test_abi_cpu.py::test_sincos_double_formula), which the coordinate encodings of the dynamic field did not tolerate
// (fill_x0_impl).
RDRF_D void sincos_double(float s, float c, float& s2, float& c2) {
  const float t = s + s;
  s2 = t * c;
  c2 = fmaf(-t, s, 1.0f);
}

template <bool FAST>
RDRF_D void fill_x0_impl(float (&X0)[32], float xn0, float xn1, float xn2, float t, int h) {
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    if (o == 0 && h == 0) {
      X0[0] = xn0; X0[1] = xn1; X0[2] = xn2; X0[3] = t;
    } else {
      const int k = 2 * o + h - 1;  // quad index 0..14: the pair of octaves (f, f + 1), f even, of one coordinate
      const int j = 2 * k;
      const int d = j / 10, f = j - d * 10;
      const float x = d == 0 ? xn0 : (d == 1 ? xn1 : xn2);
      float sv, cv;
      sincos_sel<FAST>(ldexpf(x, f), sv, cv);
      X0[o * 4 + 0] = sv;
      X0[o * 4 + 1] = cv;
      // (the odd octave exactly too: with sincos_double here -- 3 VALU instead of 26, -170 per tile -- the app-mask decisions
      // `weight > 1e-4` of a few samples flipped against the oracle's and test_trainer_step_gradient_matches_oracle_step
      // [nvidia-30000] moved by 2.6e-3 of the appearance gradients: the coordinates' encodings feed sigma, the features' do not)
      sincos_sel<FAST>(ldexpf(x, f + 1), X0[o * 4 + 2], X0[o * 4 + 3]);
    }
  }
}
RDRF_D void fill_x0(float (&X0)[32], float xn0, float xn1, float xn2, float t, int h) {
  const bool wild = !(fmaxf(fmaxf(fabsf(xn0), fabsf(xn1)), fabsf(xn2)) * 512.0f <= RDRF_PE_FAST_MAX);
  if (__builtin_expect(__any(wild), 0)) fill_x0_impl<false>(X0, xn0, xn1, xn2, t, h);
  else fill_x0_impl<true>(X0, xn0, xn1, xn2, t, h);
}
template <bool FAST>
RDRF_D void fill_x1_impl(float (&X1)[8], float t, int h) {
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const int f = 4 * o + 2 * h;   // octaves (f, f + 1), f even
    float sv, cv;
    sincos_sel<FAST>(ldexpf(t, f), sv, cv);
    X1[o * 4 + 0] = sv;
    X1[o * 4 + 1] = cv;
    sincos_sel<FAST>(ldexpf(t, f + 1), X1[o * 4 + 2], X1[o * 4 + 3]);
  }
}
RDRF_D void fill_x1(float (&X1)[8], float t, int h) {
  if (__builtin_expect(__any(!(fabsf(t) * 128.0f <= RDRF_PE_FAST_MAX)), 0)) fill_x1_impl<false>(X1, t, h);
  else fill_x1_impl<true>(X1, t, h);
}

// ---------------------------------------------------------------------------------------------
// 16-sample tiles: MFMA layer segment on v_mfma_f32_16x16x4_f32.  Weights: [NBO][KK/2][64 lanes][2] (one ds_read_b64 per
// block and pair of k-steps); consecutive MFMAs go to DIFFERENT accumulators (32-cycle issue, 40-cycle dependent latency).
// ---------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NBO, int KK>
RDRF_D void mfma16_seg(f32x4 (&acc)[NBO], const float (&in)[KK], const float* __restrict__ wp, int lane) {
  static_assert(KK % 2 == 0, "k-steps come in pairs");
  constexpr int K2 = KK / 2;
  f32x2 wc[NBO], wn[NBO];
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) wc[nb] = *(const f32x2*)(wp + (((nb * K2) * 64 + lane) << 1));
#pragma unroll
  for (int k2 = 0; k2 < K2; ++k2) {
    if (k2 + 1 < K2) {
#pragma unroll
      for (int nb = 0; nb < NBO; ++nb) wn[nb] = *(const f32x2*)(wp + (((nb * K2 + k2 + 1) * 64 + lane) << 1));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nb = 0; nb < NBO; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[nb].x, in[k2 * 2 + 0], acc[nb], 0, 0, 0);
#pragma unroll
    for (int nb = 0; nb < NBO; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[nb].y, in[k2 * 2 + 1], acc[nb], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (k2 + 1 < K2) {
#pragma unroll
      for (int nb = 0; nb < NBO; ++nb) wc[nb] = wn[nb];
    }
  }
}
// accumulator init from a packed bias ([4 groups][NBO*4]; nullptr = zero)
template <int NBO>
RDRF_D void acc16_bias(f32x4 (&acc)[NBO], const float* __restrict__ bpk, int g) {
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) acc[nb] = bpk != nullptr ? *(const f32x4*)(bpk + g * (NBO * 4) + nb * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
}
template <int NBO>
RDRF_D void acc16_relu(float (&out)[NBO * 4], const f32x4 (&acc)[NBO]) {
#pragma unroll
  for (int nb = 0; nb < NBO; ++nb) {
    out[nb * 4 + 0] = relu1(acc[nb].x); out[nb * 4 + 1] = relu1(acc[nb].y);
    out[nb * 4 + 2] = relu1(acc[nb].z); out[nb * 4 + 3] = relu1(acc[nb].w);
  }
}
// small output layer: sum_e W[o][e] * in[e] over all FOUR lane groups (no bias); ws: [4][KK]
template <int KK>
RDRF_D float dot_small16(const float (&in)[KK], const float* __restrict__ ws, int g) {
  const float* w = ws + g * KK;
  float a = 0.f;
#pragma unroll
  for (int k4 = 0; k4 < KK / 4; ++k4) {
    const f32x4 v = *(const f32x4*)(w + k4 * 4);
    a = fmaf(v.x, in[k4 * 4 + 0], a);
    a = fmaf(v.y, in[k4 * 4 + 1], a);
    a = fmaf(v.z, in[k4 * 4 + 2], a);
    a = fmaf(v.w, in[k4 * 4 + 3], a);
  }
  a += __shfl_xor(a, 16, 64);
  return a + __shfl_xor(a, 32, 64);
}
// two quads of two (plane, offset) pairs in one batch (12 loads in flight)
RDRF_D void taps_quads2(const PlaneTaps& ta, const PlaneTaps& tb, f32x4 (&out)[2]) {
  const f32x4 a00 = ld4(ta.p00), a01 = ld4(ta.p01), a10 = ld4(ta.p10), a11 = ld4(ta.p11), al0 = ld4(ta.l0), al1 = ld4(ta.l1);
  const f32x4 b00 = ld4(tb.p00), b01 = ld4(tb.p01), b10 = ld4(tb.p10), b11 = ld4(tb.p11), bl0 = ld4(tb.l0), bl1 = ld4(tb.l1);
  __builtin_amdgcn_sched_barrier(0);
  out[0] = taps_combine(ta, a00, a01, a10, a11, al0, al1);
  out[1] = taps_combine(tb, b00, b01, b10, b11, bl0, bl1);
}
// lane group g's five quads of a {48,12,12}-component set at stride level 0, 16-sample layout: out[4 j + c] = component c
// of quad app16_quad(j, g) (zeros where that is padding)
RDRF_D void gather_level_app16(const RdrfVM& vm, const PointTaps& pt, int g, float (&out_)[20]) {
  {
    const PlaneTaps xy = plane_taps(vm, 0, pt.x, pt.y, pt.z, 4 * g);
    f32x4 v[3];
    taps_quads<3>(xy, 16, v);   // quads g, g + 4, g + 8: 18 loads in flight
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      out_[4 * j + 0] = v[j].x; out_[4 * j + 1] = v[j].y; out_[4 * j + 2] = v[j].z; out_[4 * j + 3] = v[j].w;
    }
  }
  __builtin_amdgcn_sched_barrier(0);   // the second batch's pointers are formed AFTER the first batch's 72 load registers are free
  const int q = 4 * (g < 3 ? g : 2);   // group 3 reads quad 2 again (a valid address) and drops the result
  const PlaneTaps t3 = plane_taps(vm, 1, pt.x, pt.z, pt.y, q);
  const PlaneTaps t4 = plane_taps(vm, 2, pt.y, pt.z, pt.x, q);
  f32x4 w[2];
  taps_quads2(t3, t4, w);   // 12 loads in flight
  if (g == 3) { w[0] = f32x4{0.f, 0.f, 0.f, 0.f}; w[1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    out_[12 + 4 * j + 0] = w[j].x; out_[12 + 4 * j + 1] = w[j].y; out_[12 + 4 * j + 2] = w[j].z; out_[12 + 4 * j + 3] = w[j].w;
  }
}

// raw2alpha's per-sample factor 1 - alpha + 1e-10 (models/tensorBase.py:28, renderer.py:220)
RDRF_D float one_minus_alpha_eps(float alpha) {
#pragma clang fp contract(off)
  float a = 1.0f - alpha;
  return a + 1e-10f;
}
RDRF_D float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
RDRF_D float softplusf_(float x) { return x > 20.0f ? x : log1pf(expf(x)); }  // F.softplus, beta 1, threshold 20

// inclusive product scan over each 32-lane half (both halves carry identical data)
RDRF_D float scan_mul32(float v, int s) {
#pragma unroll
  for (int d = 1; d < 32; d <<= 1) {
    float o = __shfl_up(v, d, 32);
    if (s >= d) v *= o;
  }
  return v;
}
RDRF_D float scan_mul64(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float o = __shfl_up(v, d, 64);
    if (lane >= d) v *= o;
  }
  return v;
}
RDRF_D float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------
// packed-weight plan: constant offsets (floats) into the pack buffer.  Dimensions are fixed by
// the architecture every shipped config uses (featureC=128, app_dim=27, comps 16/4/4, 48/12/12).
// ---------------------------------------------------------------------------------------------
namespace pk {
// Every kernel's weights (MFMA fragments, small layers, biases) form ONE contiguous region of the
// pack buffer, copied verbatim into LDS by the persistent workgroup that uses it.
// ---- dynamic field, density/blending phase (k_dyn_density) --------------------------------
constexpr int K1_W3_X0 = 0;                                   // 2 x 32
constexpr int K1_W3_T = K1_W3_X0 + 2 * 32 * 64;               // 2 x 16
constexpr int K1_W4 = K1_W3_T + 2 * 16 * 64;                  // 2 x 32
constexpr int K1_W5 = K1_W4 + 2 * 32 * 64;                    // small 3 x 32
#ifdef RDRF_HEADS_F32   // A/B builds (tools/build_variant.sh): the heads' first layers on the fp32 matrix pipe, as up to round 5
constexpr int K1_DEN1_F = K1_W5 + 3 * 2 * 32;                 // 2 x 36
constexpr int K1_DEN1_X0 = K1_DEN1_F + 2 * 36 * 64;           // 2 x 32
constexpr int K1_DEN1_X1 = K1_DEN1_X0 + 2 * 32 * 64;          // 2 x 8
constexpr int K1_DEN2 = K1_DEN1_X1 + 2 * 8 * 64;              // small 1 x 32
constexpr int K1_BLE1_F = K1_DEN2 + 1 * 2 * 32;
constexpr int K1_BLE1_X0 = K1_BLE1_F + 2 * 36 * 64;
constexpr int K1_BLE1_X1 = K1_BLE1_X0 + 2 * 32 * 64;
constexpr int K1_BLE2 = K1_BLE1_X1 + 2 * 8 * 64;
#else
// the heads' first layers (152 -> 64 each, two thirds of the kernel's matrix work) as bf16 x 3 fragments (mfma_seg_b3): slots
// 0..35 = the 72 VM features, 36..67 = X0, 68..71 = X1[0..3] -> nine K = 16 steps; X1[4..7] stays an fp32 segment (a tenth
// step would be half padding, and 2 x 80 x 96 dwords per head would not fit the LDS next to the warp MLP)
constexpr int K1_HEAD_KK = 72;
constexpr int K1_DEN1 = K1_W5 + 3 * 2 * 32;                   // b3 2 x 72
constexpr int K1_DEN1_X1T = K1_DEN1 + 2 * K1_HEAD_KK * 96;    // fp32 2 x 4
constexpr int K1_DEN2 = K1_DEN1_X1T + 2 * 4 * 64;             // small 1 x 32
constexpr int K1_BLE1 = K1_DEN2 + 1 * 2 * 32;
constexpr int K1_BLE1_X1T = K1_BLE1 + 2 * K1_HEAD_KK * 96;
constexpr int K1_BLE2 = K1_BLE1_X1T + 2 * 4 * 64;
#endif
constexpr int K1_B3 = K1_BLE2 + 1 * 2 * 32;                   // biases [2][32]
constexpr int K1_B4 = K1_B3 + 64;
constexpr int K1_BD1 = K1_B4 + 64;
constexpr int K1_BB1 = K1_BD1 + 64;
constexpr int K1_SIZE = K1_BB1 + 64;
// ---- dynamic field, appearance phase (k_dyn_app) -------------------------------------------
constexpr int K3_BASIS = 0;                                   // 1 x 108
constexpr int K3_RGB1_F = K3_BASIS + 1 * 108 * 64;            // 4 x 16
constexpr int K3_RGB1_X0 = K3_RGB1_F + 4 * 16 * 64;           // 4 x 32
constexpr int K3_RGB1_X1 = K3_RGB1_X0 + 4 * 32 * 64;          // 4 x 8
constexpr int K3_RGB2 = K3_RGB1_X1 + 4 * 8 * 64;              // 4 x 64
constexpr int K3_RGBV = K3_RGB2 + 4 * 64 * 64;                // small 3 x 64
constexpr int K3_B1 = K3_RGBV + 3 * 2 * 64;                   // biases [2][64]
constexpr int K3_B2 = K3_B1 + 128;
constexpr int K3_SIZE = K3_B2 + 128;
// ---- scene flow (k_scene_flow) -------------------------------------------------------------
constexpr int SF_W0 = 0;                                      // 2 x 20
constexpr int SF_W2 = SF_W0 + 2 * 20 * 64;                    // 2 x 32
constexpr int SF_W4 = SF_W2 + 2 * 32 * 64;
constexpr int SF_W6 = SF_W4 + 2 * 32 * 64;                    // small 6 x 32
constexpr int SF_B0 = SF_W6 + 6 * 2 * 32;
constexpr int SF_B2 = SF_B0 + 64;
constexpr int SF_B4 = SF_B2 + 64;
constexpr int SF_SIZE = SF_B4 + 64;
// ---- static field, appearance phase (k_static_app) -----------------------------------------
constexpr int S3_BASIS = 0;                                   // 1 x 36
constexpr int S3_W1_F = S3_BASIS + 1 * 36 * 64;               // 4 x 16
constexpr int S3_W1_P = S3_W1_F + 4 * 16 * 64;                // 4 x 64
constexpr int S3_W2 = S3_W1_P + 4 * 64 * 64;                  // 4 x 64
constexpr int S3_W3 = S3_W2 + 4 * 64 * 64;                    // small 3 x 64
constexpr int S3_B1 = S3_W3 + 3 * 2 * 64;
constexpr int S3_B2 = S3_B1 + 128;
constexpr int S3_SIZE = S3_B2 + 128;
// region bases inside the pack buffer (floats)
constexpr int REG_K1 = 0;
constexpr int REG_K3 = REG_K1 + K1_SIZE;
constexpr int REG_SF = REG_K3 + K3_SIZE;
constexpr int REG_DYN_END = REG_SF + SF_SIZE;
constexpr int REG_S3 = 0;
constexpr int REG_STAT_END = REG_S3 + S3_SIZE;
// lo pieces of k_dyn_app's bf16 x 3 layers with split storage (mfma_seg_b3s): behind the dynamic images, never copied to LDS
constexpr int K3_LO_RGB1_F = 0;                               // 4 x 16 slots
constexpr int K3_LO_RGB1_X0 = K3_LO_RGB1_F + 4 * 16 * 32;     // 4 x 32
constexpr int K3_LO_RGB1_X1 = K3_LO_RGB1_X0 + 4 * 32 * 32;    // 4 x 8
constexpr int K3_LO_RGB2 = K3_LO_RGB1_X1 + 4 * 8 * 32;        // 4 x 64
constexpr int K3_LO_SIZE = K3_LO_RGB2 + 4 * 64 * 32;
constexpr int REG_K3_LO = REG_DYN_END;
constexpr int REG_DYN_END_LO = REG_K3_LO + K3_LO_SIZE;
// ---- static field, appearance phase on 16-sample tiles (k_static_app16): 16x16x4 fragments [nb][kk/2][64 lanes][2]
constexpr int S16_BASIS = 0;                                  // 2 x 20
constexpr int S16_W1_F = S16_BASIS + 2 * 20 * 64;             // 8 x 8
constexpr int S16_W1_P = S16_W1_F + 8 * 8 * 64;               // 8 x 32
constexpr int S16_W2 = S16_W1_P + 8 * 32 * 64;                // 8 x 32
constexpr int S16_W3 = S16_W2 + 8 * 32 * 64;                  // small 3 x [4 groups][32]
constexpr int S16_B1 = S16_W3 + 3 * 4 * 32;                   // biases [4 groups][32]
constexpr int S16_B2 = S16_B1 + 128;
constexpr int S16_SIZE = S16_B2 + 128;
constexpr int REG_S16 = REG_STAT_END;                         // follows the 32-sample image in the static pack area
constexpr int REG_STAT_END16 = REG_S16 + S16_SIZE;
// ---- lo pieces of the bf16 x 3 layers with split storage (mfma_seg_b3s): streamed from the pack buffer, never copied to LDS
constexpr int S3_LO_W1_F = 0;                                 // 4 x 16 slots
constexpr int S3_LO_W1_P = S3_LO_W1_F + 4 * 16 * 32;          // 4 x 64
constexpr int S3_LO_W2 = S3_LO_W1_P + 4 * 64 * 32;            // 4 x 64
constexpr int S3_LO_SIZE = S3_LO_W2 + 4 * 64 * 32;
constexpr int REG_S3_LO = REG_STAT_END16;
constexpr int REG_STAT_END_LO = REG_S3_LO + S3_LO_SIZE;
static_assert(S16_SIZE * 4 <= 160 * 1024, "the 16-sample image must fit the 160 KiB LDS");
static_assert(K1_SIZE * 4 <= 160 * 1024 && K3_SIZE * 4 <= 160 * 1024 && S3_SIZE * 4 <= 160 * 1024,
              "each kernel's weight image must fit the 160 KiB LDS");
}  // namespace pk

// descriptor of one packing job (rdrf_pack.hip)
struct PackJob {
  const float* src;  // natural [out_dim][ld]
  int ld, out_dim, in_dim;
  int seg;    // SegId of the input segment
  int mode;   // 0 = MFMA forward, 1 = small forward, 2 = MFMA transposed (backward data), 3 = bias, 7 / 8 = bf16 x 3 MFMA forward / transposed, 9 / 10 = 7 / 8 with split storage
  int nb;     // NBO (mode 0) / OUT (mode 1) / NBI (mode 2)
  int kk;     // k-steps of the segment (mode 0/1) or of the OUT dimension (mode 2)
  int dst;    // float offset into the pack buffer
  int seg_kk0;  // modes 0 / 7: first slot of the segment this job covers (a segment may be cut between two images)
  int kk_off, kk_tot;   // modes 7 / 9 (bf16 x 3 fragments): the job's first slot inside the image, the image's slots per lane half
  int dst2;             // modes 9 / 10 (bf16 x 3, split storage): float offset of the lo pieces (outside the LDS image); dst = hi + mid
};
#define RDRF_MAX_PACK_JOBS 48
struct PackJobs {
  PackJob j[RDRF_MAX_PACK_JOBS];
  int n;
};

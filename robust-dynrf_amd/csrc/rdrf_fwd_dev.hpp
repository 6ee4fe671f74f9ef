// rdrf_fwd_dev.hpp -- device bodies of the forward field kernels (TensorBase.forward, /root/reference/models/
// tensorBase.py:704-850), shared by the per-phase kernels of rdrf_fwd.hip and the single-launch fused render of
// rdrf_render.hip.  A body sees its position in the launch through GridCtx (workgroup id / count, thread id / count)
// and gets its LDS weight image as a pointer, so the same code runs as its own kernel or as one phase of a larger one.
#pragma once
#include "rdrf_kernels.hpp"

struct GridCtx {
  int bid, nblk, tid, nthr;
};
RDRF_D GridCtx grid_ctx() { return GridCtx{(int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x, (int)blockDim.x}; }
// a body that IS the kernel (FUSED = false) reads the launch geometry from the hardware registers, exactly as the
// kernels did before they were split into bodies (same code generation); as a phase of the fused render it reads gc
#define GC_BID (FUSED ? gc.bid : (int)blockIdx.x)
#define GC_NBLK (FUSED ? gc.nblk : (int)gridDim.x)
#define GC_TID (FUSED ? gc.tid : (int)threadIdx.x)
#define GC_NTHR (FUSED ? gc.nthr : (int)blockDim.x)

// App-mask compaction (models/tensorBase.py:773-790): the masked samples are appended to one list through one counter -- an
// atomic with return on ONE address for the whole chip (~3.6 ns each, serialised at the memory side: 65 k appends of a
// 32 400-ray frame are 0.23 ms during which every appending wave waits).  So the density kernels count first, append once
// per workgroup (here) or per ray (dyn_density_body), and write their entries in a second sweep over the stored weights.
// cnt is wave-uniform; returns the wave's base in the list.  All threads of the workgroup must call it.
RDRF_D int block_append_base(int* counter, int cnt, int* s_cnt, int tid, int nthr) {   // cnt: wave-uniform; returns the wave's base
  const int wave = tid >> 6, nwaves = nthr >> 6;
  __syncthreads();   // s_cnt may still be read by the previous use
  if ((tid & 63) == 0) s_cnt[wave] = cnt;
  __syncthreads();
  if (tid == 0) {
    int tot = 0;
    for (int w = 0; w < nwaves; ++w) {
      const int c = s_cnt[w];
      s_cnt[w] = tot;
      tot += c;
    }
    const int b = tot ? atomicAdd(counter, tot) : 0;
    for (int w = 0; w < nwaves; ++w) s_cnt[w] += b;
  }
  __syncthreads();
  return s_cnt[wave];
}

// Distinct static issue priorities for the waves that share a SIMD (round 6).  The waves of a persistent workgroup start
// together and do identical work per tile, and the SIMD's arbitration is fair between waves of equal priority -- so they stay
// in LOCKSTEP: all of them gather, then all of them want the matrix pipe.  The measured wave time per tile is exactly
// (waves per SIMD) x (MFMA cycles of a tile) + (non-MFMA time of ONE tile): 2 x 39.2 k + 17 k = 95.8 k cycles in
// k_static_app, 4 x 19.7 k + 24 k = 102.5 k in k_static_app16 -- the non-MFMA phase is never hidden.  With a different
// priority per co-resident wave (waves i, i + 4, i + 8, ... share a SIMD: priority 3 - (wave >> 2)) the favoured wave runs its
// MFMA chain at full rate and is in its gather phase while the next one computes; the tile queue evens out the tile counts.
RDRF_D void set_wave_priority(int wave) {
  switch ((wave >> 2) & 3) {
    case 0: __builtin_amdgcn_s_setprio(3); break;
    case 1: __builtin_amdgcn_s_setprio(2); break;
    case 2: __builtin_amdgcn_s_setprio(1); break;
    default: __builtin_amdgcn_s_setprio(0); break;
  }
}

// Start-up stagger (round 6): the k-th wave of a SIMD (k = wave >> 2) sleeps k * n * 8128 cycles once before its first tile.
// Two waves that share the matrix pipe fairly keep whatever phase difference they start with (the one that entered its
// MFMA chain first also leaves it first, by the same margin), and they start with none -- so their gather phases coincide
// for the whole launch (see set_wave_priority; s_setprio does not change the MFMA arbitration, measured twice).  An initial
// offset of about half a tile puts one wave's gathers under the other's MFMA chain.
RDRF_D void stagger_start(int wave, int n) {
  const int k = (wave >> 2) & 3;
  for (int i = 0; i < k * n; ++i) __builtin_amdgcn_s_sleep(127);
}

template <bool FEAT, bool FUSED = false>
RDRF_D void static_density_body(const FieldArgs a, const StaticW w, const GridCtx gc) {
  __shared__ int s_cnt[16];
  const int lane = GC_TID & 63;
  const int wave_ = GC_TID >> 6, nwaves_ = GC_NTHR >> 6;
  for (int nb = GC_BID * nwaves_; nb < a.N; nb += GC_NBLK * nwaves_) {   // uniform over the workgroup (block_append_base syncs)
  const bool ray = nb + wave_ < a.N;
  const int n = ray ? nb + wave_ : 0;
  float vx, vy, vz;
  float nrm = 1.0f;
  if constexpr (!FEAT) nrm = ray_norm(a.rays, n, a.ray_type, vx, vy, vz);
  float carry = 1.0f;
  int cnt = 0;   // app-masked samples of the ray (wave-uniform)
  for (int j0 = 0; j0 < a.S; j0 += 64) {
    const int j = j0 + lane;
    const bool act = ray && j < a.S && (!FEAT || n * a.S + j < a.M);
    const int idx = n * a.S + (act ? j : 0);
    const bool vld = act && (FEAT || a.valid[idx] != 0);
    if constexpr (!FEAT) {
      if (ray && a.rgb != nullptr) {   // the appearance phase writes the masked samples only: zero the round's colours (coalesced)
        float* r = a.rgb + ((size_t)n * a.S + j0) * 3 + lane;
        const int lim = (a.S - j0 < 64 ? a.S - j0 : 64) * 3;
        if (lane < lim) r[0] = 0.f;
        if (lane + 64 < lim) r[64] = 0.f;
        if (lane + 128 < lim) r[128] = 0.f;
      }
    }
    float f = 0.0f;
    if (vld) {
      float x0, x1, x2;
      if (FEAT && a.in_norm) {
        x0 = a.xyz[idx * 3 + 0]; x1 = a.xyz[idx * 3 + 1]; x2 = a.xyz[idx * 3 + 2];
      } else {
        x0 = norm_c(a.xyz[idx * 3 + 0], a.box.lo[0], a.box.inv[0]);
        x1 = norm_c(a.xyz[idx * 3 + 1], a.box.lo[1], a.box.inv[1]);
        x2 = norm_c(a.xyz[idx * 3 + 2], a.box.lo[2], a.box.inv[2]);
      }
      {  // quads 0..3 plane XY, 4 plane XZ, 5 plane YZ; one set of axis taps for all six (shared-tap gather)
        const PointTaps pt = point_taps(w.density, x0, x1, x2, 0);
        const PlaneTaps xy = plane_taps(w.density, 0, pt.x, pt.y, pt.z, 0);
        float sp = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = taps_quad(xy, 4 * q);
          sp += v.x + v.y + v.z + v.w;
        }
        f += sp;
        const f32x4 v1 = taps_quad(plane_taps(w.density, 1, pt.x, pt.z, pt.y, 0), 0);
        f += v1.x + v1.y + v1.z + v1.w;
        const f32x4 v2 = taps_quad(plane_taps(w.density, 2, pt.y, pt.z, pt.x, 0), 0);
        f += v2.x + v2.y + v2.z + v2.w;
      }
    }
    if (a.raw != nullptr && act) a.raw[idx] = f;
    if constexpr (FEAT) {  // compute_densityfeature: the raw feature (models/tensoRF.py:118-154)
      if (act && a.sigma != nullptr) a.sigma[idx] = f;
    } else {
      const float sigma = vld ? density_act(f, a.act, a.density_shift) : 0.0f;
      const float zj = act ? a.z[idx] : 0.f;
      const float zn = (j + 1 < a.S) ? a.z[idx + 1] : zj;
      const float ds = ((j + 1 < a.S) ? (zn - zj) : 0.0f) * nrm * a.distance_scale;
      const float alpha = 1.0f - expf(-sigma * ds);
      const float p = act ? one_minus_alpha_eps(alpha) : 1.0f;
      const float incl = scan_mul64(p, lane);
      float excl = __shfl_up(incl, 1, 64);
      if (lane == 0) excl = 1.0f;
      const float T = carry * excl;
      const float wt = alpha * T;
      carry *= __shfl(incl, 63, 64);
      if (act) {
        a.sigma[idx] = sigma;
        a.weight[idx] = wt;
        a.dists[idx] = ds;
      }
      cnt += __popcll(__ballot(act && wt > a.weight_thres));
    }
  }
  if constexpr (!FEAT) {
    int base = block_append_base(a.counter, cnt, s_cnt, GC_TID, GC_NTHR);
    if (cnt) {
      for (int j0 = 0; j0 < a.S; j0 += 64) {
        const int j = j0 + lane;
        const bool act = ray && j < a.S;
        const int idx = n * a.S + (act ? j : 0);
        const bool m = act && a.weight[idx] > a.weight_thres;   // the lane's own store above
        const unsigned long long bal = __ballot(m);
        if (m) a.list[base + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = idx;   // rank among the set lanes below
        base += __popcll(bal);
      }
    }
  }
  }  // ray loop
}

template <int HEAD, bool FEAT, bool SAVE = true, bool FUSED = false>
RDRF_D void static_app_body(const FieldArgs a, const StaticW w, float* lds_fused, const GridCtx gc) {
  float* lds;
  if constexpr (FUSED) lds = lds_fused;
  else {   // the standalone kernel owns its image as a named LDS array (constant addresses in every ds_read)
    __shared__ __attribute__((aligned(16))) float lds_own[pk::S3_SIZE];
    lds = lds_own;
  }
  __shared__ int s_next;
  if (GC_TID == 0) s_next = GC_NTHR >> 6;
  lds_fill(lds, a.pk + pk::REG_S3, pk::S3_SIZE);
  const int lane = GC_TID & 63, h = lane >> 5, s = lane & 31;
  const int wave = GC_TID >> 6, nwaves = GC_NTHR >> 6;
  const int count = FEAT ? a.M : *a.counter;
  const int ntiles = (count + 31) >> 5;
  const float* pkw = lds;
  const bool dynq = !FUSED && (a.dynq & 1) != 0;
  if (!FUSED && (a.dynq & 2)) set_wave_priority(wave);
  if (!FUSED) stagger_start(wave, a.dynq >> 8);
  for (int k = wave, tile; (tile = GC_BID + k * GC_NBLK) < ntiles; k = tile_queue_next(&s_next, k, nwaves, dynq)) {
    const int li = tile * 32 + s;
    const bool act = li < count;
    const int idx = act ? (FEAT ? li : a.list[li]) : 0;
    const int n = idx / a.S;
    float* svb = (SAVE && a.act3) ? a.act3 + (size_t)tile * sv::S3_ROWS * 32 : nullptr;
    float vx = 0.f, vy = 0.f, vz = 0.f;
    if constexpr (!FEAT) ray_norm(a.rays, n, a.ray_type, vx, vy, vz);
    float x0, x1, x2;
    if (FEAT && a.in_norm) {
      x0 = a.xyz[idx * 3 + 0]; x1 = a.xyz[idx * 3 + 1]; x2 = a.xyz[idx * 3 + 2];
    } else {
      x0 = norm_c(a.xyz[idx * 3 + 0], a.box.lo[0], a.box.inv[0]);
      x1 = norm_c(a.xyz[idx * 3 + 1], a.box.lo[1], a.box.inv[1]);
      x2 = norm_c(a.xyz[idx * 3 + 2], a.box.lo[2], a.box.inv[2]);
    }
    float G[36];
    gather_level_app<0>(w.app, point_taps(w.app, x0, x1, x2, 0), h, G);
    if (!act) {
#pragma unroll
      for (int i = 0; i < 36; ++i) G[i] = 0.f;
    }
    f32x16 accF[1];
    acc_bias<1>(accF, nullptr, h);
    mfma_seg<1, 36>(accF, G, pkw + pk::S3_BASIS, lane);
    float F[16];
    acc_copy<1>(F, accF);
    if constexpr (FEAT) {  // compute_appfeature: basis_mat output (models/tensoRF.py:156-196)
      save_rows<36>(svb, sv::S3_G, G, s, h);
      if (act) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
          if (elem_of(kk, h) < 27) a.feat[(size_t)idx * 27 + elem_of(kk, h)] = F[kk];
      }
      continue;
    }
    float P[64];
    {
      float fmax_ = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) fmax_ = fmaxf(fmax_, fabsf(F[r]));
      if (__builtin_expect(__any(!(fmax_ * 2.0f <= RDRF_PE_FAST_MAX)), 0)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float s1, c1, s2, c2;
          sincosf(F[r], &s1, &c1);
          sincosf(F[r] * 2.0f, &s2, &c2);
          P[4 * r + 0] = s1; P[4 * r + 1] = c1; P[4 * r + 2] = s2; P[4 * r + 3] = c2;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float s1, c1, s2, c2;
          sincos_sel<true>(F[r], s1, c1);
          sincos_double(s1, c1, s2, c2);   // (sin 2F, cos 2F): rdrf_common.hpp
          P[4 * r + 0] = s1; P[4 * r + 1] = c1; P[4 * r + 2] = s2; P[4 * r + 3] = c2;
        }
      }
    }
    if (HEAD == RDRF_HEAD_MLP_FEA) {  // viewdirs ride in the pad slots 27..29 of the feature block
      if (h == 0) F[15] = vx;
      else { F[12] = vy; F[13] = vz; }
    }
    if (svb != nullptr && h == 0) {
      svb[(size_t)(sv::S3_VD + 0) * 32 + s] = vx; svb[(size_t)(sv::S3_VD + 1) * 32 + s] = vy;
      svb[(size_t)(sv::S3_VD + 2) * 32 + s] = vz;
    }
    save_rows<36>(svb, sv::S3_G, G, s, h);
    save_rows<16>(svb, sv::S3_F, F, s, h);
    save_rows<64>(svb, sv::S3_P, P, s, h);
    f32x16 acc[4];
    acc_bias<4>(acc, pkw + pk::S3_B1, h);
    float H1[64];
#ifdef RDRF_APP_F32
    mfma_seg<4, 16>(acc, F, pkw + pk::S3_W1_F, lane);
    mfma_seg<4, 64>(acc, P, pkw + pk::S3_W1_P, lane);
    acc_relu<4>(H1, acc);
    save_rows<64>(svb, sv::S3_H1, H1, s, h);
    acc_bias<4>(acc, pkw + pk::S3_B2, h);
    mfma_seg<4, 64>(acc, H1, pkw + pk::S3_W2, lane);
#else
    {  // the two hidden layers on the bf16 matrix pipe (fp32-grade bf16 x 3, lo pieces streamed from the pack buffer)
      const B3sLo st = b3s_lo_stream(a.pk + pk::REG_S3_LO, lane);
      u32x4 lo[4];
      b3s_lo_load<4>(lo, st, pk::S3_LO_W1_F, 2, 0);
      mfma_seg_b3s<4, 16, 64>(acc, F, pkw + pk::S3_W1_F, st, pk::S3_LO_W1_F, pk::S3_LO_W1_P, lo, lane);
      mfma_seg_b3s<4, 64, 64>(acc, P, pkw + pk::S3_W1_P, st, pk::S3_LO_W1_P, pk::S3_LO_W2, lo, lane);
      acc_relu<4>(H1, acc);
      save_rows<64>(svb, sv::S3_H1, H1, s, h);
      acc_bias<4>(acc, pkw + pk::S3_B2, h);
      mfma_seg_b3s<4, 64, 0>(acc, H1, pkw + pk::S3_W2, st, pk::S3_LO_W2, 0, lo, lane);
    }
#endif
    acc_relu<4>(H1, acc);
    save_rows<64>(svb, sv::S3_H2, H1, s, h);
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float v = dot_small<64>(H1, pkw + pk::S3_W3 + o * 128, h) + w.b3[o];
      if (HEAD == RDRF_HEAD_MLP_FEA_TIMEEMBEDDING)
        v += w.w3[o * 131 + 128] * vx + w.w3[o * 131 + 129] * vy + w.w3[o * 131 + 130] * vz;
      if (act && h == 0) a.rgb[(size_t)idx * 3 + o] = sigmoidf_(v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The static appearance phase on 16-SAMPLE tiles (round 6).  static_app_body holds a 32-sample tile per
// wave on v_mfma_f32_32x32x2_f32: 64 + 64 + 64 activation / accumulator registers, 256 VGPRs, two waves per SIMD -- which
// do not cover the gather round trips of a tile (64 % of the wave cycles in SQ_WAIT_INST_ANY).  Here a wave owns 16 samples
// on v_mfma_f32_16x16x4_f32 (same FLOP per cycle): lane (g = l>>4, s = l&15) holds a quarter of sample s' vector, every
// activation / accumulator array is half as long, the kernel fits 128 VGPRs and runs sixteen waves per workgroup = four
// per SIMD.  Same arithmetic per element as static_app_body (the MFMA k order differs: sums of the same products in
// another order), same saved rows ([32-sample tile][row][32]: tile16 t writes columns 16 (t & 1) .. + 15 of tile t >> 1),
// so the backward kernels are unchanged.  Ray path only (FEAT / FUSED stay on static_app_body).
// ------------------------------------------------------------------------------------------------
// tile_base is WAVE-UNIFORM (a scalar register pair), voff = 128 g + s the lane's float offset inside a row block: every
// store is `global_store_dword voff, data, s[base] offset:imm` -- no 64-bit per-lane row pointers to keep alive or spill
template <int KK>
RDRF_D void save_rows16(float* __restrict__ tile_base, int row0, const float (&v)[KK], unsigned voff) {
  if (tile_base == nullptr) return;
#ifdef RDRF_ABL_NOSAVE
  return;
#endif
#pragma unroll
  for (int kk = 0; kk < KK; ++kk)
    __builtin_nontemporal_store(v[kk], tile_base + (row0 + ((kk >> 2) << 4) + (kk & 3)) * 32 + voff);
}

template <int HEAD, bool SAVE>
RDRF_D void static_app16_body(const FieldArgs a, const StaticW w) {
  __shared__ __attribute__((aligned(16))) float lds[pk::S16_SIZE];
  __shared__ int s_next;
  if (threadIdx.x == 0) s_next = blockDim.x >> 6;
  lds_fill(lds, a.pk + pk::REG_S16, pk::S16_SIZE);
  const int lane = threadIdx.x & 63, g = lane >> 4, s = lane & 15;
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int count = *a.counter;
  const int ntiles = ((count + 31) >> 5) * 2;   // both halves of the last 32-sample tile: its saved rows are read whole
  const float* pkw = lds;
  const unsigned voff = 128 * g + s;                                 // rows 4 g + r of a 16-row block, column s
  const unsigned voffP = (32 * (g >> 1) + 4 * (g & 1)) * 32 + s;     // PE rows: the row of element (feature w, m) in the 32-sample order
  const bool dynq = (a.dynq & 1) != 0;
  if (a.dynq & 2) set_wave_priority(wave);
  stagger_start(wave, a.dynq >> 8);
  for (int k = wave, tile_; (tile_ = blockIdx.x + k * gridDim.x) < ntiles; k = tile_queue_next(&s_next, k, nwaves, dynq)) {
    const int tile = __builtin_amdgcn_readfirstlane(tile_);   // wave-uniform: the saved-row base stays in scalar registers
    const int li = tile * 16 + s;
    const bool act = li < count;
    const int idx = act ? a.list[li] : 0;
    const int n = idx / a.S;
    float* svb = (SAVE && a.act3) ? a.act3 + (size_t)(tile >> 1) * sv::S3_ROWS * 32 + 16 * (tile & 1) : nullptr;
    float vx, vy, vz;
    ray_norm(a.rays, n, a.ray_type, vx, vy, vz);
    const float x0 = norm_c(a.xyz[idx * 3 + 0], a.box.lo[0], a.box.inv[0]);
    const float x1 = norm_c(a.xyz[idx * 3 + 1], a.box.lo[1], a.box.inv[1]);
    const float x2 = norm_c(a.xyz[idx * 3 + 2], a.box.lo[2], a.box.inv[2]);
    float F[8];
    {
      float G[20];
      // g made opaque per tile: left visible, the optimiser hoists every lane-dependent pointer (plane / line base + quad
      // offset of the lane group, eleven 64-bit values) out of the tile loop and spills them
      int gq = g;
      asm volatile("" : "+v"(gq));
      gather_level_app16(w.app, point_taps<true>(w.app, x0, x1, x2, 0), gq, G);
      if (!act) {
#pragma unroll
        for (int i = 0; i < 20; ++i) G[i] = 0.f;
      }
      if (svb != nullptr) {   // rows in the natural component order: XY quads 4 j + g, XZ quads 12 + g, YZ quads 15 + g
#pragma unroll
        for (int kk = 0; kk < 12; ++kk)
          __builtin_nontemporal_store(G[kk], svb + (sv::S3_G + ((kk >> 2) << 4) + (kk & 3)) * 32 + voff);
        if (g < 3) {
#pragma unroll
          for (int kk = 12; kk < 20; ++kk)
            __builtin_nontemporal_store(G[kk], svb + (sv::S3_G + (kk < 16 ? 48 : 60) + (kk & 3)) * 32 + voff);
        }
      }
      f32x4 accF[2];
      acc16_bias<2>(accF, nullptr, g);
      mfma16_seg<2, 20>(accF, G, pkw + pk::S16_BASIS, lane);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        F[nb * 4 + 0] = accF[nb].x; F[nb * 4 + 1] = accF[nb].y; F[nb * 4 + 2] = accF[nb].z; F[nb * 4 + 3] = accF[nb].w;
      }
    }
    float P[32];
    {
      float fmax_ = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) fmax_ = fmaxf(fmax_, fabsf(F[r]));
      if (__builtin_expect(__any(!(fmax_ * 2.0f <= RDRF_PE_FAST_MAX)), 0)) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          float s1, c1, s2, c2;
          sincosf(F[r], &s1, &c1);
          sincosf(F[r] * 2.0f, &s2, &c2);
          P[4 * r + 0] = s1; P[4 * r + 1] = c1; P[4 * r + 2] = s2; P[4 * r + 3] = c2;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          float s1, c1, s2, c2;
          sincos_sel<true>(F[r], s1, c1);
          sincos_double(s1, c1, s2, c2);
          P[4 * r + 0] = s1; P[4 * r + 1] = c1; P[4 * r + 2] = s2; P[4 * r + 3] = c2;
        }
      }
    }
    if (HEAD == RDRF_HEAD_MLP_FEA) {  // viewdirs ride in the pad elements 27..29 of the feature block
      if (g == 2) F[7] = vx;
      if (g == 3) { F[4] = vy; F[5] = vz; }
    }
    if (svb != nullptr) {
      if (g == 0) {
        svb[(size_t)(sv::S3_VD + 0) * 32 + s] = vx; svb[(size_t)(sv::S3_VD + 1) * 32 + s] = vy;
        svb[(size_t)(sv::S3_VD + 2) * 32 + s] = vz;
      }
      save_rows16<8>(svb, sv::S3_F, F, voff);
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          __builtin_nontemporal_store(P[4 * r + m], svb + (sv::S3_P + 64 * (r >> 2) + 8 * (r & 3) + m) * 32 + voffP);
    }
    f32x4 acc[8];
    float H[32];
    __builtin_amdgcn_sched_barrier(0);   // the 32 accumulator registers are claimed after the encodings' temporaries are dead
    acc16_bias<8>(acc, pkw + pk::S16_B1, g);
    mfma16_seg<8, 8>(acc, F, pkw + pk::S16_W1_F, lane);
    mfma16_seg<8, 32>(acc, P, pkw + pk::S16_W1_P, lane);
    acc16_relu<8>(H, acc);
    save_rows16<32>(svb, sv::S3_H1, H, voff);
    acc16_bias<8>(acc, pkw + pk::S16_B2, g);
    mfma16_seg<8, 32>(acc, H, pkw + pk::S16_W2, lane);
    acc16_relu<8>(H, acc);
    save_rows16<32>(svb, sv::S3_H2, H, voff);
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float v = dot_small16<32>(H, pkw + pk::S16_W3 + o * 128, g) + w.b3[o];
      if (HEAD == RDRF_HEAD_MLP_FEA_TIMEEMBEDDING)
        v += w.w3[o * 131 + 128] * vx + w.w3[o * 131 + 129] * vy + w.w3[o * 131 + 130] * vz;
      if (act && g == 0) a.rgb[(size_t)idx * 3 + o] = sigmoidf_(v);
    }
  }
}

#ifndef RDRF_HEADS_F32
// first layer of the density / blending head (152 -> 64) on the bf16 matrix pipe: [features 36 | X0 32 | X1 0..3] as nine
// K = 16 steps of mfma_seg_b3, X1[4..7] as an fp32 segment (pk::K1_DEN1)
RDRF_D void head_layer1(f32x16 (&acc)[2], const float (&Fv)[36], const float (&X0)[32], const float (&X1)[8],
                        const float* __restrict__ wb3, const float* __restrict__ wtail, int lane) {
  float U[pk::K1_HEAD_KK];
#pragma unroll
  for (int i = 0; i < 36; ++i) U[i] = Fv[i];
#pragma unroll
  for (int i = 0; i < 32; ++i) U[36 + i] = X0[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) U[68 + i] = X1[i];
  mfma_seg_b3<2, pk::K1_HEAD_KK>(acc, U, wb3, lane);
  const float V[4] = {X1[4], X1[5], X1[6], X1[7]};
  mfma_seg<2, 4>(acc, V, wtail, lane);
}
#endif

// FLAT (inference only): the unit of work is a 32-sample tile of the flattened [N * S] sample array instead of a ray, so a
// 512-ray eval chunk is 1840 tiles for the 2048 resident waves instead of 512 rays, and no tile is padded to the ray's
// end (S = 115: 3.6 tiles per ray instead of 4).  The per-ray constants (time, tout, PE8(t)) are fetched per lane through
// the sample's ray index; sigma and blending are written per sample and the transmittance scan / app-mask compaction run
// afterwards in ray_scan_body (same arithmetic in the same order: results are bit-identical to the wave-per-ray form).
template <bool FEAT, bool SAVE = true, bool FUSED = false, bool FLAT = false>
RDRF_D void dyn_density_body(const FieldArgs a, const DynW w, float* lds_fused, const GridCtx gc) {
  static_assert(!FLAT || (!FEAT && !SAVE && !FUSED), "the flat-tile form is the inference kernel");
  float* lds;
  if constexpr (FUSED) lds = lds_fused;
  else {   // the standalone kernel owns its image as a named LDS array (constant addresses in every ds_read)
    __shared__ __attribute__((aligned(16))) float lds_own[pk::K1_SIZE];
    lds = lds_own;
  }
  lds_fill(lds, a.pk + pk::REG_K1, pk::K1_SIZE);
  const int lane = GC_TID & 63, h = lane >> 5, s = lane & 31;
  const int wave = GC_TID >> 6, nwaves = GC_NTHR >> 6;
  const float* pkw = lds;
  const int nunits = FLAT ? (int)(((long)a.N * a.S + 31) >> 5) : a.N;
  for (int n = GC_BID * nwaves + wave; n < nunits; n += GC_NBLK * nwaves) {
  float t = 0.f, nrm = 1.0f;
  float T[16];
  float X1[8];
  if constexpr (!FEAT && !FLAT) {   // time is per ray: tout / PE8(t) are per-ray constants
    t = a.ts[n];
    float vx, vy, vz;
    nrm = ray_norm(a.rays, n, a.ray_type, vx, vy, vz);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = ld4(a.tout + n * 32 + 8 * q + 4 * h);
      T[q * 4 + 0] = v.x; T[q * 4 + 1] = v.y; T[q * 4 + 2] = v.z; T[q * 4 + 3] = v.w;
    }
    fill_x1(X1, t, h);
  }
  float carry = 1.0f;
  int nmask = 0;   // app-masked samples of the ray (wave-uniform)
  for (int j0 = 0; j0 < (FLAT ? 32 : a.S); j0 += 32) {
    const int j = j0 + s;
    bool act;
    int idx;
    if constexpr (FLAT) {   // n = tile of the flattened sample array
      const long i = (long)n * 32 + s;
      act = i < (long)a.N * a.S;
      idx = act ? (int)i : 0;
    } else {
      act = j < a.S && (!FEAT || n * a.S + j < a.M);
      idx = n * a.S + (act ? j : 0);
    }
    const bool vld = act && (FEAT || a.valid[idx] != 0);
    if constexpr (FEAT || FLAT) {  // time is per point (FEAT) / looked up through the sample's ray (FLAT)
      const int ti = FLAT ? idx / a.S : idx;
      t = a.ts[ti];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = ld4(a.tout + (size_t)ti * 32 + 8 * q + 4 * h);
        T[q * 4 + 0] = v.x; T[q * 4 + 1] = v.y; T[q * 4 + 2] = v.z; T[q * 4 + 3] = v.w;
      }
      fill_x1(X1, t, h);
    }
#ifndef RDRF_T_RESIDENT   // the per-ray time-branch outputs are re-read per tile (L1 hits) instead of living in 16 registers across the
    // tiles: the training instantiation loses its 29 spilled registers (104 B of scratch per lane): kernel -6 % / -7 % (stage 0 / final,
    // profiles/r06_ab_t_reload.txt; -DRDRF_T_RESIDENT builds the old form).  Same loads, same bits.
    if constexpr (!FEAT && !FLAT) {
      int nn = n;
      asm volatile("" : "+v"(nn));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = ld4(a.tout + nn * 32 + 8 * q + 4 * h);
        T[q * 4 + 0] = v.x; T[q * 4 + 1] = v.y; T[q * 4 + 2] = v.z; T[q * 4 + 3] = v.w;
      }
    }
#endif
    const float px = a.xyz[idx * 3 + 0], py = a.xyz[idx * 3 + 1], pz = a.xyz[idx * 3 + 2];
    const bool raw_in = FEAT && a.in_norm;   // compute_*: the caller hands normalised coordinates
    const float xn0 = raw_in ? px : norm_c(px, a.box.lo[0], a.box.inv[0]);
    const float xn1 = raw_in ? py : norm_c(py, a.box.lo[1], a.box.inv[1]);
    const float xn2 = raw_in ? pz : norm_c(pz, a.box.lo[2], a.box.inv[2]);
    float X0[32];
    fill_x0(X0, xn0, xn1, xn2, t, h);
    float* svb = (SAVE && a.act1) ? a.act1 + ((size_t)n * ((a.S + 31) >> 5) + (j0 >> 5)) * sv::K1_ROWS * 32 : nullptr;
    save_rows<32>(svb, sv::K1_X0, X0, s, h);
    save_rows<8>(svb, sv::K1_X1, X1, s, h);
    save_rows<16>(svb, sv::K1_T, T, s, h);
    // ---- warp MLP: [xn, PE10(xn), tout] -> 64 -> 64 -> 3  (models/tensoRF.py:521-541)
    float d0, d1, d2;
    {
      f32x16 acc[2];
      acc_bias<2>(acc, pkw + pk::K1_B3, h);
      mfma_seg<2, 32>(acc, X0, pkw + pk::K1_W3_X0, lane);
      mfma_seg<2, 16>(acc, T, pkw + pk::K1_W3_T, lane);
      float H3[32];
      acc_relu<2>(H3, acc);
      save_rows<32>(svb, sv::K1_H3, H3, s, h);
      acc_bias<2>(acc, pkw + pk::K1_B4, h);
      mfma_seg<2, 32>(acc, H3, pkw + pk::K1_W4, lane);
      acc_relu<2>(H3, acc);
      save_rows<32>(svb, sv::K1_H4, H3, s, h);
      d0 = dot_small<32>(H3, pkw + pk::K1_W5 + 0 * 64, h) + w.l5b[0];
      d1 = dot_small<32>(H3, pkw + pk::K1_W5 + 1 * 64, h) + w.l5b[1];
      d2 = dot_small<32>(H3, pkw + pk::K1_W5 + 2 * 64, h) + w.l5b[2];
    }
    // compute_* warp the un-normalised round trip of xn (models/tensoRF.py:647-649)
    const float xw0 = norm_c(unnorm_c(xn0, a.box.lo[0], a.box.inv[0]) + d0, a.box.lo[0], a.box.inv[0]);
    const float xw1 = norm_c(unnorm_c(xn1, a.box.lo[1], a.box.inv[1]) + d1, a.box.lo[1], a.box.inv[1]);
    const float xw2 = norm_c(unnorm_c(xn2, a.box.lo[2], a.box.inv[2]) + d2, a.box.lo[2], a.box.inv[2]);
    if (act && h == 0) {
      if (!FEAT || a.xyz_prime != nullptr) {
        a.xyz_prime[(size_t)idx * 3 + 0] = (raw_in ? unnorm_c(xn0, a.box.lo[0], a.box.inv[0]) : px) + d0;
        a.xyz_prime[(size_t)idx * 3 + 1] = (raw_in ? unnorm_c(xn1, a.box.lo[1], a.box.inv[1]) : py) + d1;
        a.xyz_prime[(size_t)idx * 3 + 2] = (raw_in ? unnorm_c(xn2, a.box.lo[2], a.box.inv[2]) : pz) + d2;
      }
      a.xw[(size_t)idx * 3 + 0] = xw0;
      a.xw[(size_t)idx * 3 + 1] = xw1;
      a.xw[(size_t)idx * 3 + 2] = xw2;
    }
    // ---- density and blending heads: 3-stride VM features (72) + X0 + X1 -> 64 -> 1
    float fd, fb;
    {
      float Fv[36];
      gather_level_den<0>(w.density, point_taps(w.density, xw0, xw1, xw2, 0), h, Fv);
      gather_level_den<12>(w.density, point_taps(w.density, xw0, xw1, xw2, 1), h, Fv);
      gather_level_den<24>(w.density, point_taps(w.density, xw0, xw1, xw2, 2), h, Fv);
      if (!vld) {
#pragma unroll
        for (int i = 0; i < 36; ++i) Fv[i] = 0.f;
      }
      f32x16 acc[2];
      acc_bias<2>(acc, pkw + pk::K1_BD1, h);
#ifdef RDRF_HEADS_F32
      mfma_seg<2, 36>(acc, Fv, pkw + pk::K1_DEN1_F, lane);
      mfma_seg<2, 32>(acc, X0, pkw + pk::K1_DEN1_X0, lane);
      mfma_seg<2, 8>(acc, X1, pkw + pk::K1_DEN1_X1, lane);
#else
      head_layer1(acc, Fv, X0, X1, pkw + pk::K1_DEN1, pkw + pk::K1_DEN1_X1T, lane);
#endif
      float Hd[32];
      acc_relu<2>(Hd, acc);
      save_rows<36>(svb, sv::K1_FD, Fv, s, h);
      save_rows<32>(svb, sv::K1_HD, Hd, s, h);
      fd = dot_small<32>(Hd, pkw + pk::K1_DEN2, h) + w.db2[0];
    }
    {
      float Fv[36];
      gather_level_den<0>(w.blending, point_taps(w.blending, xw0, xw1, xw2, 0), h, Fv);
      gather_level_den<12>(w.blending, point_taps(w.blending, xw0, xw1, xw2, 1), h, Fv);
      gather_level_den<24>(w.blending, point_taps(w.blending, xw0, xw1, xw2, 2), h, Fv);
      if (!vld) {
#pragma unroll
        for (int i = 0; i < 36; ++i) Fv[i] = 0.f;
      }
      f32x16 acc[2];
      acc_bias<2>(acc, pkw + pk::K1_BB1, h);
#ifdef RDRF_HEADS_F32
      mfma_seg<2, 36>(acc, Fv, pkw + pk::K1_BLE1_F, lane);
      mfma_seg<2, 32>(acc, X0, pkw + pk::K1_BLE1_X0, lane);
      mfma_seg<2, 8>(acc, X1, pkw + pk::K1_BLE1_X1, lane);
#else
      head_layer1(acc, Fv, X0, X1, pkw + pk::K1_BLE1, pkw + pk::K1_BLE1_X1T, lane);
#endif
      float Hd[32];
      acc_relu<2>(Hd, acc);
      save_rows<36>(svb, sv::K1_FB, Fv, s, h);
      save_rows<32>(svb, sv::K1_HB, Hd, s, h);
      fb = dot_small<32>(Hd, pkw + pk::K1_BLE2, h) + w.bb2[0];
    }
    if constexpr (FEAT) {  // compute_densityfeature / compute_blendingfeature: the raw head outputs
      if (act && h == 0) {
        if (a.sigma != nullptr) a.sigma[idx] = fd;
        if (a.blending != nullptr) a.blending[idx] = fb;
        if (a.raw != nullptr) { a.raw[(size_t)idx * 2] = fd; a.raw[(size_t)idx * 2 + 1] = fb; }
      }
    } else if constexpr (FLAT) {   // the scan and the compaction follow in ray_scan_body
      if (act && h == 0) {
        a.sigma[idx] = vld ? density_act(fd, a.act, a.density_shift) : 0.0f;
        a.blending[idx] = vld ? sigmoidf_(fb) : 0.0f;
      }
    } else {
    const float sigma = vld ? density_act(fd, a.act, a.density_shift) : 0.0f;
    const float blend = vld ? sigmoidf_(fb) : 0.0f;
    const float zj = act ? a.z[idx] : 0.f;
    const float zn = (j + 1 < a.S) ? a.z[idx + 1] : zj;
    const float ds = ((j + 1 < a.S) ? (zn - zj) : 0.0f) * nrm * a.distance_scale;
    const float alpha = 1.0f - expf(-sigma * ds);
    const float p = act ? one_minus_alpha_eps(alpha) : 1.0f;
    const float incl = scan_mul32(p, s);
    float excl = __shfl_up(incl, 1, 32);
    if (s == 0) excl = 1.0f;
    const float Tr = carry * excl;
    const float wt = alpha * Tr;
    carry *= __shfl(incl, 31, 32);
    const bool m = act && h == 0 && wt > a.weight_thres;
    if (act && h == 0) {
      a.sigma[idx] = sigma;
      a.weight[idx] = wt;
      a.dists[idx] = ds;
      a.blending[idx] = blend;
      if (SAVE && a.raw != nullptr) { a.raw[(size_t)idx * 2] = fd; a.raw[(size_t)idx * 2 + 1] = fb; }
    }
    nmask += __popcll(__ballot(m));
    }
  }
  if constexpr (!FEAT && !FLAT) {
    // app-mask compaction (models/tensorBase.py:773-790), ONE append per ray: every append is an atomic with return on the
    // same address for the whole chip (~3.6 ns each, serialised at the memory side), so the ray's tiles count first and
    // the entries are written in a second sweep over the weights just stored (same lane, program order)
    if (nmask) {
      int base = 0;
      if (lane == 0) base = atomicAdd(a.counter, nmask);
      base = __shfl(base, 0, 64);
      for (int j0 = 0; j0 < a.S; j0 += 32) {
        const int j = j0 + s;
        const bool act = j < a.S && h == 0;
        const int idx = n * a.S + (act ? j : 0);
        const bool m = act && a.weight[idx] > a.weight_thres;
        const unsigned long long bal = __ballot(m);
        if (m) a.list[base + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = idx;   // rank among the set lanes below
        base += __popcll(bal);
      }
    }
  }
  }  // ray loop
}

// The per-ray half of the density phase for the flat-tile form: sigma -> alpha -> transmittance scan -> weight, dists, the
// app-mask compaction (models/tensorBase.py:773-790) and the zero fill of the colours the appearance phase will not write.
// A half-wave per ray (two rays per wave), 32 samples per round with the carry across rounds -- the expressions and the
// order of the multiplications are those of dyn_density_body, so weight / dists / the mask come out bit-identical.
// The compaction appends ONCE per workgroup (block_append_base).
RDRF_D void ray_scan_body(const FieldArgs a) {
  __shared__ int s_cnt[16];
  const int lane = threadIdx.x & 63, h = lane >> 5, s = lane & 31;
  const int n_ = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + h;
  const bool ray = n_ < a.N;
  const int n = ray ? n_ : 0;
  float vx, vy, vz;
  const float nrm = ray_norm(a.rays, n, a.ray_type, vx, vy, vz);
  float carry = 1.0f;
  int cnt = 0;
  for (int j0 = 0; j0 < a.S; j0 += 32) {
    const int j = j0 + s;
    const bool act = ray && j < a.S;
    const int idx = n * a.S + (act ? j : 0);
    const float sigma = act ? a.sigma[idx] : 0.0f;
    const float zj = act ? a.z[idx] : 0.f;
    const float zn = (j + 1 < a.S) ? a.z[idx + 1] : zj;
    const float ds = ((j + 1 < a.S) ? (zn - zj) : 0.0f) * nrm * a.distance_scale;
    const float alpha = 1.0f - expf(-sigma * ds);
    const float p = act ? one_minus_alpha_eps(alpha) : 1.0f;
    const float incl = scan_mul32(p, s);
    float excl = __shfl_up(incl, 1, 32);
    if (s == 0) excl = 1.0f;
    const float Tr = carry * excl;
    const float wt = alpha * Tr;
    carry *= __shfl(incl, 31, 32);
    if (act) {
      a.weight[idx] = wt;
      a.dists[idx] = ds;
    }
    if (ray && a.rgb != nullptr) {   // zero the round's colours (coalesced; the appearance phase writes the masked samples)
      float* r = a.rgb + ((size_t)n * a.S + j0) * 3 + s;
      const int lim = (a.S - j0 < 32 ? a.S - j0 : 32) * 3;
      if (s < lim) r[0] = 0.f;
      if (s + 32 < lim) r[32] = 0.f;
      if (s + 64 < lim) r[64] = 0.f;
    }
    cnt += __popcll(__ballot(act && wt > a.weight_thres));
  }
  int base = block_append_base(a.counter, cnt, s_cnt, threadIdx.x, blockDim.x);
  if (cnt) {
    for (int j0 = 0; j0 < a.S; j0 += 32) {
      const int j = j0 + s;
      const bool act = ray && j < a.S;
      const int idx = n * a.S + (act ? j : 0);
      const bool m = act && a.weight[idx] > a.weight_thres;   // the lane's own store above
      const unsigned long long bal = __ballot(m);
      if (m) a.list[base + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = idx;   // rank among the set lanes below
      base += __popcll(bal);
    }
  }
}

template <bool FEAT, bool SAVE = true, bool FUSED = false>
RDRF_D void dyn_app_body(const FieldArgs a, const DynW w, float* lds_fused, const GridCtx gc) {
  float* lds;
  if constexpr (FUSED) lds = lds_fused;
  else {   // the standalone kernel owns its image as a named LDS array (constant addresses in every ds_read)
    __shared__ __attribute__((aligned(16))) float lds_own[pk::K3_SIZE];
    lds = lds_own;
  }
  __shared__ int s_next;
  if (GC_TID == 0) s_next = GC_NTHR >> 6;
  lds_fill(lds, a.pk + pk::REG_K3, pk::K3_SIZE);
  const int lane = GC_TID & 63, h = lane >> 5, s = lane & 31;
  const int wave = GC_TID >> 6, nwaves = GC_NTHR >> 6;
  const int count = FEAT ? a.M : *a.counter;
  const int ntiles = (count + 31) >> 5;
  const float* pkw = lds;
  const bool dynq = !FUSED && (a.dynq & 1) != 0;   // tile queue of the workgroup (tile_queue_next)
  for (int k = wave, tile; (tile = GC_BID + k * GC_NBLK) < ntiles; k = tile_queue_next(&s_next, k, nwaves, dynq)) {
    const int li = tile * 32 + s;
    const bool act = li < count;
    const int idx = act ? (FEAT ? li : a.list[li]) : 0;
    const int n = idx / a.S;
    const float t = FEAT ? 0.f : a.ts[n];
    float* svb = (SAVE && a.act3) ? a.act3 + (size_t)tile * sv::K3_ROWS * 32 : nullptr;
    float vx = 0.f, vy = 0.f, vz = 0.f;
    float xn0 = 0.f, xn1 = 0.f, xn2 = 0.f;
    if constexpr (!FEAT) {
      ray_norm(a.rays, n, a.ray_type, vx, vy, vz);
      xn0 = norm_c(a.xyz[idx * 3 + 0], a.box.lo[0], a.box.inv[0]);
      xn1 = norm_c(a.xyz[idx * 3 + 1], a.box.lo[1], a.box.inv[1]);
      xn2 = norm_c(a.xyz[idx * 3 + 2], a.box.lo[2], a.box.inv[2]);
    }
    const float xw0 = a.xw[(size_t)idx * 3 + 0], xw1 = a.xw[(size_t)idx * 3 + 1], xw2 = a.xw[(size_t)idx * 3 + 2];
    float F[16];
    {
      // level by level: gather a level's 36 features (two load batches), feed them to the basis product, store their
      // rows -- the 108 gathered values never coexist, which leaves the registers for 36 loads in flight
      f32x16 accF[1];
      acc_bias<1>(accF, nullptr, h);
      {
        float A[36];
        gather_level_app<0>(w.app, point_taps(w.app, xw0, xw1, xw2, 0), h, A);
        if (!act) {
#pragma unroll
          for (int i = 0; i < 36; ++i) A[i] = 0.f;
        }
        mfma_seg<1, 36>(accF, A, pkw + pk::K3_BASIS, lane);
        save_rows<36>(svb, sv::K3_A, A, s, h);
      }
      {
        float A[36];
        gather_level_app<0>(w.app, point_taps(w.app, xw0, xw1, xw2, 1), h, A);
        if (!act) {
#pragma unroll
          for (int i = 0; i < 36; ++i) A[i] = 0.f;
        }
        mfma_seg<1, 36>(accF, A, pkw + pk::K3_BASIS + 9 * 256, lane);
        save_rows<36>(svb, sv::K3_A + 72, A, s, h);
      }
      {
        float A[36];
        gather_level_app<0>(w.app, point_taps(w.app, xw0, xw1, xw2, 2), h, A);
        if (!act) {
#pragma unroll
          for (int i = 0; i < 36; ++i) A[i] = 0.f;
        }
        mfma_seg<1, 36>(accF, A, pkw + pk::K3_BASIS + 18 * 256, lane);
        save_rows<36>(svb, sv::K3_A + 144, A, s, h);
      }
      acc_copy<1>(F, accF);
    }
    if constexpr (FEAT) {  // compute_appfeature: basis_mat output (models/tensoRF.py:734-811)
      if (act) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
          if (elem_of(kk, h) < 27) a.feat[(size_t)idx * 27 + elem_of(kk, h)] = F[kk];
      }
      continue;
    }
    float X0[32], X1[8];
#ifdef RDRF_ABL_APP_NOPE
#pragma unroll
    for (int i = 0; i < 32; ++i) X0[i] = xn0 * (float)(i + 1) + t;
#pragma unroll
    for (int i = 0; i < 8; ++i) X1[i] = t * (float)(i + 1);
#else
    fill_x0(X0, xn0, xn1, xn2, t, h);
    fill_x1(X1, t, h);
#endif
    if (svb != nullptr && h == 0) {
      svb[(size_t)(sv::K3_VD + 0) * 32 + s] = vx; svb[(size_t)(sv::K3_VD + 1) * 32 + s] = vy;
      svb[(size_t)(sv::K3_VD + 2) * 32 + s] = vz;
    }
    save_rows<16>(svb, sv::K3_F, F, s, h);
    save_rows<32>(svb, sv::K3_X0, X0, s, h);
    save_rows<8>(svb, sv::K3_X1, X1, s, h);
    f32x16 acc[4];
    acc_bias<4>(acc, pkw + pk::K3_B1, h);
    float H1[64];
#ifdef RDRF_APP_F32
    mfma_seg<4, 16>(acc, F, pkw + pk::K3_RGB1_F, lane);
    mfma_seg<4, 32>(acc, X0, pkw + pk::K3_RGB1_X0, lane);
    mfma_seg<4, 8>(acc, X1, pkw + pk::K3_RGB1_X1, lane);
    acc_relu<4>(H1, acc);
    save_rows<64>(svb, sv::K3_H1, H1, s, h);
    acc_bias<4>(acc, pkw + pk::K3_B2, h);
    mfma_seg<4, 64>(acc, H1, pkw + pk::K3_RGB2, lane);
#else
    {  // the two hidden layers on the bf16 matrix pipe (fp32-grade bf16 x 3, lo pieces streamed from the pack buffer)
      const B3sLo st = b3s_lo_stream(a.pk + pk::REG_K3_LO, lane);
      u32x4 lo[4];
      b3s_lo_load<4>(lo, st, pk::K3_LO_RGB1_F, 2, 0);
      mfma_seg_b3s<4, 16, 32>(acc, F, pkw + pk::K3_RGB1_F, st, pk::K3_LO_RGB1_F, pk::K3_LO_RGB1_X0, lo, lane);
      mfma_seg_b3s<4, 32, 8>(acc, X0, pkw + pk::K3_RGB1_X0, st, pk::K3_LO_RGB1_X0, pk::K3_LO_RGB1_X1, lo, lane);
      mfma_seg_b3s<4, 8, 64>(acc, X1, pkw + pk::K3_RGB1_X1, st, pk::K3_LO_RGB1_X1, pk::K3_LO_RGB2, lo, lane);
      acc_relu<4>(H1, acc);
      save_rows<64>(svb, sv::K3_H1, H1, s, h);
      acc_bias<4>(acc, pkw + pk::K3_B2, h);
      mfma_seg_b3s<4, 64, 0>(acc, H1, pkw + pk::K3_RGB2, st, pk::K3_LO_RGB2, 0, lo, lane);
    }
#endif
    acc_relu<4>(H1, acc);
    save_rows<64>(svb, sv::K3_H2, H1, s, h);
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float v = dot_small<64>(H1, pkw + pk::K3_RGBV + o * 128, h) + w.rbv[o];
      v += w.rwv[o * 131 + 128] * vx + w.rwv[o * 131 + 129] * vy + w.rwv[o * 131 + 130] * vz;
      if (act && h == 0) a.rgb[(size_t)idx * 3 + o] = sigmoidf_(v);
    }
  }
}

// time branch of the dynamic field for the rays [0, N): 32 lanes per ray (lane o: hidden neurons o and o + 32, then
// output o); s_h: [threads / 32][64] floats of LDS.  Every thread of the block must call (barriers inside).
template <bool FUSED = false>
RDRF_D void time_branch_body(const float* __restrict__ ts, const DynW w, int N, float* __restrict__ tout, float* s_h,
                             const GridCtx gc) {
  const int rpb = GC_NTHR >> 5;
  const int r = GC_TID >> 5, o = GC_TID & 31;
  for (int base = GC_BID * rpb; base < N; base += GC_NBLK * rpb) {
    const int n = base + r;
    const bool act = n < N;
    const float t = act ? ts[n] : 0.f;
    float tin[17];
    tin[0] = t;
#pragma unroll
    for (int f = 0; f < 8; ++f) sincosf(ldexpf(t, f), &tin[1 + f], &tin[9 + f]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = o + 32 * j;
      float hk = w.l1b[k];
#pragma unroll
      for (int i = 0; i < 17; ++i) hk = fmaf(w.l1w[k * 17 + i], tin[i], hk);
      s_h[r * 64 + k] = fmaxf(hk, 0.0f);
    }
    __syncthreads();
    float out = 0.f;
    if (o < 30) {
      out = w.l2b[o];
      for (int k = 0; k < 64; ++k) out = fmaf(w.l2w[o * 64 + k], s_h[r * 64 + k], out);   // k ascending, as before
    }
    if (act) tout[n * 32 + o] = out;
    __syncthreads();
  }
}

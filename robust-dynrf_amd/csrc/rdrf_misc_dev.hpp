// rdrf_misc_dev.hpp -- device bodies of the sampler (renderer.sampleXYZ, /root/reference/renderer.py:147-170, models/
// tensorBase.py:487-559) and of the compositor (raw2outputs, renderer.py:173-315), shared by their own kernels in
// rdrf_misc.hip and by the single-launch fused render of rdrf_render.hip.
#pragma once
#include "rdrf_fwd_dev.hpp"

// ATen's CPU linspace (aten/src/ATen/native/cpu/RangeFactoriesKernel.cpp), fp32, as the shipped torch
// binaries compute it: step = (end - start) / (steps - 1); element i is start + step * i in the first
// half and end - step * (steps - i - 1) in the second, and GCC contracts each multiply-add into ONE
// fused operation (checked bit-for-bit against torch.linspace over many (start, end, steps),
// tests/test_oracle_golden.py::test_linspace_formula).  The `valid` byte mask depends on these bits.
RDRF_D float linspace_at(float start, float end, int steps, int i) {
#pragma clang fp contract(off)
  if (steps <= 1) return start;
  const float step = (end - start) / (float)(steps - 1);
  if (i < steps / 2) return __builtin_fmaf(step, (float)i, start);
  return __builtin_fmaf(-step, (float)(steps - i - 1), end);
}

RDRF_D void sample_ndc_body(const float* __restrict__ rays, int N, int S, float near, float far,
                            const float* __restrict__ jitter, Box box, float* __restrict__ xyz,
                            float* __restrict__ z, uint8_t* __restrict__ valid, const GridCtx gc) {
#pragma clang fp contract(off)
  for (long i = (long)gc.bid * gc.nthr + gc.tid; i < (long)N * S; i += (long)gc.nblk * gc.nthr) {
  const int n = (int)(i / S), j = (int)(i - (long)n * S);
  float t = linspace_at(near, far, S, j);
  if (jitter) {
    const float c = (float)(((double)far - (double)near) / (double)S);
    const float jj = jitter[j] * c;
    t = t + jj;
  }
  const float* r = rays + (size_t)n * 6;
  bool out = false;
  for (int k = 0; k < 3; ++k) {
    const float m = r[3 + k] * t;
    const float p = r[k] + m;
    xyz[i * 3 + k] = p;
    out = out || (box.lo[k] > p) || (p > box.hi[k]);
  }
  z[i] = t;
  valid[i] = out ? 0 : 1;
  }
}

RDRF_D void sample_contract_body(const float* __restrict__ rays, int N, int S, float near,
                                 float far, const float* __restrict__ jin,
                                 const float* __restrict__ jout, float* __restrict__ xyz,
                                 float* __restrict__ z, uint8_t* __restrict__ valid, const GridCtx gc) {
#pragma clang fp contract(off)
  for (long i = (long)gc.bid * gc.nthr + gc.tid; i < (long)N * S; i += (long)gc.nblk * gc.nthr) {
  const int n = (int)(i / S), j = (int)(i - (long)n * S);
  const int inner = S - S / 2, outer = S / 2;
  float t;
  if (j < inner) {
    float a = linspace_at(near, 2.0f, inner + 1, j);
    float b = linspace_at(near, 2.0f, inner + 1, j + 1);
    if (jin) {
      const float c = (float)((2.0 - (double)near) / (double)inner);
      a = a + jin[j] * c;
      if (j + 1 < inner) b = b + jin[j + 1] * c;
    }
    t = (b + a) * 0.5f;
  } else {
    const int k = j - inner;  // flipped: rng'[k] = rng[outer-k]
    float r0 = (float)(outer - k), r1 = (float)(outer - k - 1);
    if (jout) {
      if (outer - k < outer) r0 = r0 + jout[outer - k];
      r1 = r1 + jout[outer - k - 1];
    }
    const float mid = (r1 + r0) * 0.5f;
    const float c2 = (float)(1.0 / 2.0 - 1.0 / (double)far);
    const float inv_far = (float)(1.0 / (double)far);
    const float den = inv_far + (c2 * mid) / (float)outer;
    t = 1.0f / den;
  }
  const float* r = rays + (size_t)n * 6;
  float p[3];
  float nrm = 0.f;
  for (int k = 0; k < 3; ++k) {
    const float m = r[3 + k] * t;
    p[k] = r[k] + m;
    nrm = fmaxf(nrm, fabsf(p[k]));
  }
  if (nrm > 1.0f) {
    const float sc = 2.0f - 1.0f / nrm;
    for (int k = 0; k < 3; ++k) p[k] = sc * (p[k] / nrm);
  }
  for (int k = 0; k < 3; ++k) xyz[i * 3 + k] = p[k];
  z[i] = t;
  valid[i] = 1;
  }
}


// ------------------------------------------------------------------------------------------------
// compositor (raw2outputs)
// ------------------------------------------------------------------------------------------------
struct CompArgs {
  const float *rgb_s, *sigma_s, *rgb_d, *sigma_d, *dists, *blending, *z, *rays;
  int N, S, ray_type, add_white_bg;
  const float* white_dev;   // nullable: the coin as a device float (overrides add_white_bg; HIP-graph replays)
  float* out[13];
};

RDRF_D float alpha_of(float sigma, float dist) { return 1.0f - expf(-sigma * dist); }
RDRF_D float tfull_factor(float ad, float as, float b) {
#pragma clang fp contract(off)
  const float u = 1.0f - ad * b;
  const float v = 1.0f - as * (1.0f - b);
  return u * v + 1e-10f;
}

RDRF_D void composite_body(const CompArgs a, const GridCtx gc) {
  const int lane = gc.tid & 63;
  const int wave_ = gc.tid >> 6, nwaves_ = gc.nthr >> 6;
  for (int n = gc.bid * nwaves_ + wave_; n < a.N; n += gc.nblk * nwaves_) {
  const int S = a.S;
  float cd = 1.f, cs = 1.f, cf = 1.f;  // scan carries
  float sum_wd = 0.f;
  float rs[3] = {0, 0, 0}, rf[3] = {0, 0, 0};
  float acc_s = 0.f, acc_f = 0.f, dep_s = 0.f, dep_f = 0.f, dyn = 0.f;
  float* w_full = a.out[3];
  float* w_s = a.out[7];
  float* w_d = a.out[11];
  for (int j0 = 0; j0 < S; j0 += 64) {
    const int j = j0 + lane;
    const bool act = j < S;
    const size_t idx = (size_t)n * S + (act ? j : 0);
    const float di = a.dists[idx], b = a.blending[idx], zz = a.z[idx];
    const float ad = act ? alpha_of(a.sigma_d[idx], di) : 0.f;
    const float as = act ? alpha_of(a.sigma_s[idx], di) : 0.f;
    const float pd = act ? one_minus_alpha_eps(ad) : 1.f;
    const float ps = act ? one_minus_alpha_eps(as) : 1.f;
    const float pf = act ? tfull_factor(ad, as, b) : 1.f;
    const float id = scan_mul64(pd, lane), is = scan_mul64(ps, lane), ifl = scan_mul64(pf, lane);
    float ed = __shfl_up(id, 1, 64), es = __shfl_up(is, 1, 64), ef = __shfl_up(ifl, 1, 64);
    if (lane == 0) { ed = 1.f; es = 1.f; ef = 1.f; }
    const float Td = cd * ed, Ts = cs * es, Tf = cf * ef;
    cd *= __shfl(id, 63, 64);
    cs *= __shfl(is, 63, 64);
    cf *= __shfl(ifl, 63, 64);
    if (act) {
      const float wd = ad * Td, ws = as * Ts;
      const float wf = (ad * b + as * (1.0f - b)) * Tf;
      const float fd = Tf * ad * b, fs = Tf * as * (1.0f - b);
      sum_wd += wd;
      w_d[idx] = wd;  // raw; normalised in pass 2
      w_s[idx] = ws;
      w_full[idx] = wf;
      for (int c = 0; c < 3; ++c) {
        const float cs_ = a.rgb_s[idx * 3 + c], cd_ = a.rgb_d[idx * 3 + c];
        rs[c] += ws * cs_;
        rf[c] += fd * cd_ + fs * cs_;
      }
      acc_s += ws; acc_f += wf;
      dep_s += ws * zz; dep_f += wf * zz;
      dyn += wf * b;
    }
  }
  sum_wd = wave_sum(sum_wd);
  const float denom = sum_wd + 1e-10f;
  float rd[3] = {0, 0, 0}, acc_d = 0.f, dep_d = 0.f;
  for (int j0 = 0; j0 < S; j0 += 64) {
    const int j = j0 + lane;
    if (j < S) {
      const size_t idx = (size_t)n * S + j;
      const float wd = w_d[idx] / denom;
      w_d[idx] = wd;
      for (int c = 0; c < 3; ++c) rd[c] += wd * a.rgb_d[idx * 3 + c];
      acc_d += wd;
      dep_d += wd * a.z[idx];
    }
  }
  for (int c = 0; c < 3; ++c) { rs[c] = wave_sum(rs[c]); rf[c] = wave_sum(rf[c]); rd[c] = wave_sum(rd[c]); }
  acc_s = wave_sum(acc_s); acc_f = wave_sum(acc_f); acc_d = wave_sum(acc_d);
  dep_s = wave_sum(dep_s); dep_f = wave_sum(dep_f); dep_d = wave_sum(dep_d);
  dyn = wave_sum(dyn);
  if (lane == 0) {
    const float rl = fmaxf(1.0f - acc_f, 0.0f);
    if (a.white_dev ? (a.white_dev[0] != 0.f) : (a.add_white_bg != 0)) {
      for (int c = 0; c < 3; ++c) { rd[c] += 1.0f - acc_d; rs[c] += 1.0f - acc_s; rf[c] += rl; }
    }
    if (a.ray_type == RDRF_RAY_NDC) {
      const float far = a.rays[(size_t)n * 6 + 2] + a.rays[(size_t)n * 6 + 5];
      dep_d += (1.0f - acc_d) * far; dep_s += (1.0f - acc_s) * far; dep_f += rl * far;
    } else if (a.ray_type == RDRF_RAY_CONTRACT) {
      dep_d += (1.0f - acc_d) * 256.0f; dep_s += (1.0f - acc_s) * 256.0f; dep_f += rl * 256.0f;
    }
    for (int c = 0; c < 3; ++c) {
      a.out[0][(size_t)n * 3 + c] = fminf(fmaxf(rf[c], 0.f), 1.f);
      a.out[4][(size_t)n * 3 + c] = fminf(fmaxf(rs[c], 0.f), 1.f);
      a.out[8][(size_t)n * 3 + c] = fminf(fmaxf(rd[c], 0.f), 1.f);
    }
    a.out[1][n] = dep_f; a.out[2][n] = acc_f;
    a.out[5][n] = dep_s; a.out[6][n] = acc_s;
    a.out[9][n] = dep_d; a.out[10][n] = acc_d;
    a.out[12][n] = dyn + rl * 0.0f;
  }
  }  // ray loop
}



// rdrf_misc.hip -- ray generation, ray samplers and the three-way alpha compositor.
//   ray generation : /root/reference/train.py:96-103,1062-1077; dataLoader/ray_utils.py:53-140;
//                    camera.py:8-15
//   samplers       : models/tensorBase.py:487-499 (ndc), 524-559 (contract); renderer.py:147-170
//   compositor     : renderer.py:173-315 (raw2outputs) -- one wave per ray, three exclusive
//                    transmittance scans done as wave-level multiplicative scans with a carry.
#include "rdrf_misc_dev.hpp"

// ------------------------------------------------------------------------------------------------
// ray generation
// ------------------------------------------------------------------------------------------------
// uv (nullable): [N][2] pixel coordinates (column + 0.5 + flow_x, row + 0.5 + flow_y) that replace the
// pixel centre of the ray id -- the flow-displaced rays of train.py:1433-1460, 1530-1557, 1968-1990;
// view_shift: the camera is frame view + view_shift clamped to [0, T-1] (allposes_refine_f / _b).
__global__ void k_generate_rays(const int64_t* __restrict__ ids, const float* __restrict__ uv, int view_shift,
                                const float* __restrict__ poses9,
                                const float* __restrict__ focal_p, int N, int T, int H, int W,
                                int ndc, float near, float* __restrict__ rays) {
#pragma clang fp contract(off)
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const long id = ids[n];
  const int col = (int)(id % W), row = (int)((id / W) % H);
  int view = (int)(id / ((long)W * H)) + view_shift;
  view = view < 0 ? 0 : (view >= T ? T - 1 : view);
  const float focal = focal_p[0];
  const float i = uv ? uv[2 * n] : (float)col + 0.5f, j = uv ? uv[2 * n + 1] : (float)row + 0.5f;
  const float cx = (float)((double)W / 2), cy = (float)((double)H / 2);
  const float dir0 = (i - cx) / focal, dir1 = -(j - cy) / focal, dir2 = -1.0f;
  const float* p = poses9 + view * 9;
  // pose_to_mtx: Gram-Schmidt on the 6-D rotation
  float b1[3] = {p[0], p[1], p[2]};
  const float n1 = sqrtf(b1[0] * b1[0] + b1[1] * b1[1] + b1[2] * b1[2]);
  b1[0] /= n1; b1[1] /= n1; b1[2] /= n1;
  const float dt = b1[0] * p[3] + b1[1] * p[4] + b1[2] * p[5];
  float b2[3] = {p[3] - dt * b1[0], p[4] - dt * b1[1], p[5] - dt * b1[2]};
  const float n2 = sqrtf(b2[0] * b2[0] + b2[1] * b2[1] + b2[2] * b2[2]);
  b2[0] /= n2; b2[1] /= n2; b2[2] /= n2;
  const float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2],
                       b1[0] * b2[1] - b1[1] * b2[0]};
  float d[3], o[3] = {p[6], p[7], p[8]};
  for (int r = 0; r < 3; ++r) d[r] = dir0 * b1[r] + dir1 * b2[r] + dir2 * b3[r];
  if (ndc) {  // ndc_rays_blender2
    const float t = -(near + o[2]) / d[2];
    o[0] = o[0] + t * d[0]; o[1] = o[1] + t * d[1]; o[2] = o[2] + t * d[2];
    const float kw = -1.0f / ((float)W / (2.0f * focal)), kh = -1.0f / ((float)H / (2.0f * focal));
    const float o0 = kw * o[0] / o[2];
    const float o1 = kh * o[1] / o[2];
    const float o2 = 1.0f + 2.0f * near / o[2];
    const float d0 = kw * (d[0] / d[2] - o[0] / o[2]);
    const float d1 = kh * (d[1] / d[2] - o[1] / o[2]);
    const float d2 = -2.0f * near / o[2];
    o[0] = o0; o[1] = o1; o[2] = o2; d[0] = d0; d[1] = d1; d[2] = d2;
  }
  float* r = rays + (size_t)n * 6;
  r[0] = o[0]; r[1] = o[1]; r[2] = o[2]; r[3] = d[0]; r[4] = d[1]; r[5] = d[2];
}

extern "C" int rdrf_generate_rays_uv(const int64_t* ids, const float* uv, int view_shift, const float* poses9,
                                     const float* focal, int N, int T, int H, int W, int ndc, float near,
                                     float* rays, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(ids && poses9 && focal && rays && N > 0 && T > 0 && H > 0 && W > 0, -1, "generate_rays: bad arguments");
  RDRF_LAUNCH("generate_rays", k_generate_rays, dim3((N + 255) / 256), dim3(256), stream, ids, uv, view_shift,
              poses9, focal, N, T, H, W, ndc, near, rays);
  return 0;
}
extern "C" int rdrf_generate_rays(const int64_t* ids, const float* poses9, const float* focal,
                                  int N, int T, int H, int W, int ndc, float near, float* rays,
                                  rdrf_stream_t stream_) {
  return rdrf_generate_rays_uv(ids, nullptr, 0, poses9, focal, N, T, H, W, ndc, near, rays, stream_);
}

// ------------------------------------------------------------------------------------------------
// samplers
// ------------------------------------------------------------------------------------------------
__global__ void k_sample_ndc(const float* __restrict__ rays, int N, int S, float near, float far,
                             const float* __restrict__ jitter, Box box, float* __restrict__ xyz,
                             float* __restrict__ z, uint8_t* __restrict__ valid) {
  sample_ndc_body(rays, N, S, near, far, jitter, box, xyz, z, valid, grid_ctx());
}
__global__ void k_sample_contract(const float* __restrict__ rays, int N, int S, float near,
                                  float far, const float* __restrict__ jin,
                                  const float* __restrict__ jout, float* __restrict__ xyz,
                                  float* __restrict__ z, uint8_t* __restrict__ valid) {
  sample_contract_body(rays, N, S, near, far, jin, jout, xyz, z, valid, grid_ctx());
}

extern "C" int rdrf_sample_ndc(const float* rays, int N, int S, float near, float far,
                               const float* jitter, const float aabb_host[6], float* xyz, float* z,
                               uint8_t* valid, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(N > 0 && S > 0 && aabb_host, -1, "sample_ndc: bad arguments");
  Box b;
  for (int i = 0; i < 3; ++i) {
    b.lo[i] = aabb_host[i];
    b.hi[i] = aabb_host[3 + i];
    b.inv[i] = 2.0f / (b.hi[i] - b.lo[i]);
  }
  const long total = (long)N * S;
  RDRF_LAUNCH("sample_ndc", k_sample_ndc, dim3((unsigned)((total + 255) / 256)), dim3(256), stream,
              rays, N, S, near, far, jitter, b, xyz, z, valid);
  return 0;
}

extern "C" int rdrf_sample_contract(const float* rays, int N, int S, float near, float far,
                                    const float* jitter_inner, const float* jitter_outer,
                                    float* xyz, float* z, uint8_t* valid, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(N > 0 && S > 1, -1, "sample_contract: bad arguments");
  const long total = (long)N * S;
  RDRF_LAUNCH("sample_contract", k_sample_contract, dim3((unsigned)((total + 255) / 256)),
              dim3(256), stream, rays, N, S, near, far, jitter_inner, jitter_outer, xyz, z, valid);
  return 0;
}

__global__ __launch_bounds__(64) void k_composite(CompArgs a) { composite_body(a, grid_ctx()); }

extern "C" int rdrf_composite_fwd(const float* rgb_s, const float* sigma_s, const float* rgb_d,
                                  const float* sigma_d, const float* dists, const float* blending,
                                  const float* z, const float* rays, int N, int S, int ray_type,
                                  int add_white_bg, const float* white_dev, float* const out13[13],
                                  rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(N > 0 && S > 0 && out13, -1, "composite_fwd: bad arguments");
  CompArgs a;
  a.rgb_s = rgb_s; a.sigma_s = sigma_s; a.rgb_d = rgb_d; a.sigma_d = sigma_d;
  a.dists = dists; a.blending = blending; a.z = z; a.rays = rays;
  a.N = N; a.S = S; a.ray_type = ray_type; a.add_white_bg = add_white_bg; a.white_dev = white_dev;
  for (int i = 0; i < 13; ++i) {
    RDRF_CHECK(out13[i] != nullptr, -1, "composite_fwd: output %d is NULL", i);
    a.out[i] = out13[i];
  }
  RDRF_LAUNCH("composite", k_composite, dim3(N), dim3(64), stream, a);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// compositor backward (autograd of renderer.py:173-315).  One wave per ray, three sweeps:
//   1. recompute the sums (pre-clamp maps, accumulations, normaliser U of weights_d)
//   2. per-sample upstream gradients on each weight -> the three suffix-sum totals
//   3. final per-sample gradients with running prefix sums (suffix = total - prefix)
// Transmittance: T_k = prod_{j<k} p_j  =>  dL/dp_j = (sum_{m>j} T_m A_m) / p_j.
// ------------------------------------------------------------------------------------------------
struct CompBArgs {
  const float *rgb_s, *sigma_s, *rgb_d, *sigma_d, *dists, *blending, *z, *rays;
  int N, S, ray_type, add_white_bg;
  const float* white_dev;
  const float* g[13];
  float* gi[8];
};

struct CompSample {
  float ad, as, b, di, zz, pd, ps, pf, Td, Ts, Tf, sd, ss;
  float cd[3], cs[3];
};

RDRF_D CompSample comp_load(const CompBArgs& a, int n, int j, int lane, float& cd, float& cs, float& cf) {
  CompSample q;
  const bool act = j < a.S;
  const size_t idx = (size_t)n * a.S + (act ? j : 0);
  q.di = a.dists[idx]; q.b = a.blending[idx]; q.zz = a.z[idx];
  q.sd = a.sigma_d[idx]; q.ss = a.sigma_s[idx];
  q.ad = act ? alpha_of(q.sd, q.di) : 0.f;
  q.as = act ? alpha_of(q.ss, q.di) : 0.f;
  q.pd = act ? one_minus_alpha_eps(q.ad) : 1.f;
  q.ps = act ? one_minus_alpha_eps(q.as) : 1.f;
  q.pf = act ? tfull_factor(q.ad, q.as, q.b) : 1.f;
  const float id = scan_mul64(q.pd, lane), is = scan_mul64(q.ps, lane), ifl = scan_mul64(q.pf, lane);
  float ed = __shfl_up(id, 1, 64), es = __shfl_up(is, 1, 64), ef = __shfl_up(ifl, 1, 64);
  if (lane == 0) { ed = 1.f; es = 1.f; ef = 1.f; }
  q.Td = cd * ed; q.Ts = cs * es; q.Tf = cf * ef;
  cd *= __shfl(id, 63, 64); cs *= __shfl(is, 63, 64); cf *= __shfl(ifl, 63, 64);
  for (int c = 0; c < 3; ++c) { q.cd[c] = a.rgb_d[idx * 3 + c]; q.cs[c] = a.rgb_s[idx * 3 + c]; }
  if (!act) { q.ad = 0.f; q.as = 0.f; }
  return q;
}

// exclusive reverse (suffix) sum over the wave: sum of v over lanes > lane
RDRF_D float wave_suffix_excl(float v, int lane, float& total) {
  float inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_down(inc, d, 64);
    if (lane + d < 64) inc += o;
  }
  total = __shfl(inc, 0, 64);
  float ex = __shfl_down(inc, 1, 64);
  if (lane == 63) ex = 0.f;
  return ex;
}

__global__ __launch_bounds__(64) void k_composite_bwd(CompBArgs a) {
  __shared__ float carr[3][64];  // transmittance carries at each 64-sample tile start (S <= 4096)
  const int lane = threadIdx.x;
  const int n = blockIdx.x;
  if (n >= a.N) return;
  const int S = a.S;
  // ---- sweep 1
  float U = 0.f, acc_s = 0.f, acc_f = 0.f;
  float ru[3] = {0, 0, 0}, rs[3] = {0, 0, 0}, rf[3] = {0, 0, 0};
  {
    float cd = 1.f, cs = 1.f, cf = 1.f;
    for (int j0 = 0; j0 < S; j0 += 64) {
      if (lane == 0) { carr[0][j0 >> 6] = cd; carr[1][j0 >> 6] = cs; carr[2][j0 >> 6] = cf; }
      const int j = j0 + lane;
      CompSample q = comp_load(a, n, j, lane, cd, cs, cf);
      if (j < S) {
        const float u = q.ad * q.Td, ws = q.as * q.Ts;
        const float fd = q.Tf * q.ad * q.b, fs = q.Tf * q.as * (1.0f - q.b);
        U += u; acc_s += ws; acc_f += (q.ad * q.b + q.as * (1.0f - q.b)) * q.Tf;
        for (int c = 0; c < 3; ++c) { ru[c] += u * q.cd[c]; rs[c] += ws * q.cs[c]; rf[c] += fd * q.cd[c] + fs * q.cs[c]; }
      }
    }
    U = wave_sum(U); acc_s = wave_sum(acc_s); acc_f = wave_sum(acc_f);
    for (int c = 0; c < 3; ++c) { ru[c] = wave_sum(ru[c]); rs[c] = wave_sum(rs[c]); rf[c] = wave_sum(rf[c]); }
  }
  __syncthreads();
  const float Ue = U + 1e-10f;
  const float acc_d = U / Ue;
  const float white = (a.white_dev ? (a.white_dev[0] != 0.f) : (a.add_white_bg != 0)) ? 1.f : 0.f;
  const bool rl_on = (1.0f - acc_f) > 0.f;
  const float rl = rl_on ? 1.0f - acc_f : 0.f;
  float far = 0.f;
  if (a.ray_type == RDRF_RAY_NDC) far = a.rays[(size_t)n * 6 + 2] + a.rays[(size_t)n * 6 + 5];
  else if (a.ray_type == RDRF_RAY_CONTRACT) far = 256.0f;
  float grf[3], grs[3], grd[3], sgf = 0.f, sgs = 0.f, sgd = 0.f;
  for (int c = 0; c < 3; ++c) {
    const float pf = rf[c] + white * rl, ps = rs[c] + white * (1.0f - acc_s),
                pd = ru[c] / Ue + white * (1.0f - acc_d);
    grf[c] = (a.g[0] && pf >= 0.f && pf <= 1.f) ? a.g[0][(size_t)n * 3 + c] : 0.f;
    grs[c] = (a.g[4] && ps >= 0.f && ps <= 1.f) ? a.g[4][(size_t)n * 3 + c] : 0.f;
    grd[c] = (a.g[8] && pd >= 0.f && pd <= 1.f) ? a.g[8][(size_t)n * 3 + c] : 0.f;
    sgf += grf[c]; sgs += grs[c]; sgd += grd[c];
  }
  const float Gdep_f = a.g[1] ? a.g[1][n] : 0.f, Gacc_f = a.g[2] ? a.g[2][n] : 0.f;
  const float Gdep_s = a.g[5] ? a.g[5][n] : 0.f, Gacc_s = a.g[6] ? a.g[6][n] : 0.f;
  const float Gdep_d = a.g[9] ? a.g[9][n] : 0.f, Gacc_d = a.g[10] ? a.g[10][n] : 0.f;
  const float Gdyn = a.g[12] ? a.g[12][n] : 0.f;
  const float const_d = Gacc_d - white * sgd - Gdep_d * far;
  const float const_s = Gacc_s - white * sgs - Gdep_s * far;
  const float const_f = Gacc_f + (rl_on ? (-white * sgf - Gdep_f * far) : 0.f);
  // ---- sweep 2: D1 = sum_m gwd_m * wd_m  (normalisation of weights_d)
  float D1 = 0.f;
  {
    float cd = 1.f, cs = 1.f, cf = 1.f;
    for (int j0 = 0; j0 < S; j0 += 64) {
      const int j = j0 + lane;
      CompSample q = comp_load(a, n, j, lane, cd, cs, cf);
      if (j < S) {
        const size_t idx = (size_t)n * S + j;
        float dd = 0.f;
        for (int c = 0; c < 3; ++c) dd += grd[c] * q.cd[c];
        const float gwd = (a.g[11] ? a.g[11][idx] : 0.f) + dd + Gdep_d * q.zz + const_d;
        D1 += gwd * ((q.ad * q.Td) / Ue);
      }
    }
    D1 = wave_sum(D1);
  }
  // ---- sweep 3, LAST tile first: direct suffix sums (no total - prefix cancellation; the
  // factor 1/p can be 1e10 when alpha -> 1)
  {
    float suf_d = 0.f, suf_s = 0.f, suf_f = 0.f;  // sums over all later tiles
    const int ntile = (S + 63) >> 6;
    for (int tl = ntile - 1; tl >= 0; --tl) {
      const int j0 = tl << 6;
      float cd = carr[0][tl], cs = carr[1][tl], cf = carr[2][tl];
      const int j = j0 + lane;
      CompSample q = comp_load(a, n, j, lane, cd, cs, cf);
      const bool act = j < S;
      const size_t idx = (size_t)n * S + (act ? j : 0);
      const float u = q.ad * q.Td, ws = q.as * q.Ts;
      const float cfd = q.ad * q.b, cfs = q.as * (1.0f - q.b);
      const float wf = (cfd + cfs) * q.Tf;
      float dd = 0.f, dsv = 0.f, dfd = 0.f, dfs = 0.f;
      for (int c = 0; c < 3; ++c) { dd += grd[c] * q.cd[c]; dsv += grs[c] * q.cs[c]; dfd += grf[c] * q.cd[c]; dfs += grf[c] * q.cs[c]; }
      const float gwd = (a.g[11] ? a.g[11][idx] : 0.f) + dd + Gdep_d * q.zz + const_d;
      const float gws = (a.g[7] ? a.g[7][idx] : 0.f) + dsv + Gdep_s * q.zz + const_s;
      const float gwf = (a.g[3] ? a.g[3][idx] : 0.f) + Gdep_f * q.zz + Gdyn * q.b + const_f;
      const float gu = (gwd - D1) / Ue;
      const float td = act ? gu * u : 0.f, ts_ = act ? gws * ws : 0.f;
      const float tf = act ? q.Tf * (cfd * dfd + cfs * dfs + gwf * (cfd + cfs)) : 0.f;
      float tot_d, tot_s, tot_f;
      const float sd = suf_d + wave_suffix_excl(td, lane, tot_d);
      const float ss = suf_s + wave_suffix_excl(ts_, lane, tot_s);
      const float sfv = suf_f + wave_suffix_excl(tf, lane, tot_f);
      suf_d += tot_d; suf_s += tot_s; suf_f += tot_f;
      if (act) {
        float g_ad = gu * q.Td - sd / q.pd;
        float g_as = gws * q.Ts - ss / q.ps;
        const float g_pf = sfv / q.pf;
        const float g_cfd = q.Tf * (dfd + gwf), g_cfs = q.Tf * (dfs + gwf);
        const float uu = 1.0f - q.ad * q.b, vv = 1.0f - q.as * (1.0f - q.b);
        g_ad += g_cfd * q.b - g_pf * q.b * vv;
        g_as += g_cfs * (1.0f - q.b) - g_pf * uu * (1.0f - q.b);
        const float g_b = Gdyn * wf + g_cfd * q.ad - g_cfs * q.as + g_pf * (-q.ad * vv + uu * q.as);
        const float wdn = u / Ue;
        if (a.gi[1]) a.gi[1][idx] += g_as * q.di * (1.0f - q.as);
        if (a.gi[3]) a.gi[3][idx] += g_ad * q.di * (1.0f - q.ad);
        if (a.gi[4]) a.gi[4][idx] += g_ad * q.sd * (1.0f - q.ad) + g_as * q.ss * (1.0f - q.as);
        if (a.gi[5]) a.gi[5][idx] += g_b;
        if (a.gi[6]) a.gi[6][idx] += Gdep_d * wdn + Gdep_s * ws + Gdep_f * wf;
        for (int c = 0; c < 3; ++c) {
          if (a.gi[0]) a.gi[0][idx * 3 + c] += grs[c] * ws + grf[c] * q.Tf * cfs;
          if (a.gi[2]) a.gi[2][idx * 3 + c] += grd[c] * wdn + grf[c] * q.Tf * cfd;
        }
      }
    }
  }
  if (lane == 0 && a.gi[7] && a.ray_type == RDRF_RAY_NDC) {
    const float gfar = Gdep_d * (1.0f - acc_d) + Gdep_s * (1.0f - acc_s) + Gdep_f * rl;
    a.gi[7][(size_t)n * 6 + 2] += gfar;
    a.gi[7][(size_t)n * 6 + 5] += gfar;
  }
}

extern "C" int rdrf_composite_bwd(const float* rgb_s, const float* sigma_s, const float* rgb_d,
                                  const float* sigma_d, const float* dists, const float* blending,
                                  const float* z, const float* rays, int N, int S, int ray_type,
                                  int add_white_bg, const float* white_dev, const float* const g_out13[13],
                                  float* const g_in8[8], rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(N > 0 && S > 0 && S <= 4096 && g_out13 && g_in8, -1, "composite_bwd: bad arguments (S <= 4096)");
  CompBArgs a;
  a.rgb_s = rgb_s; a.sigma_s = sigma_s; a.rgb_d = rgb_d; a.sigma_d = sigma_d;
  a.dists = dists; a.blending = blending; a.z = z; a.rays = rays;
  a.N = N; a.S = S; a.ray_type = ray_type; a.add_white_bg = add_white_bg; a.white_dev = white_dev;
  for (int i = 0; i < 13; ++i) a.g[i] = g_out13[i];
  for (int i = 0; i < 8; ++i) a.gi[i] = g_in8[i];
  RDRF_LAUNCH("composite_bwd", k_composite_bwd, dim3(N), dim3(64), stream, a);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// sampler backward: xyz = o + d * z (then the L-inf contraction for ray_type contract)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_sample_bwd(const float* __restrict__ rays,
                                                   const float* __restrict__ zrow, int N, int S,
                                                   int ray_type, const float* __restrict__ g_xyz,
                                                   float* __restrict__ g_rays) {
  const int lane = threadIdx.x, n = blockIdx.x;
  if (n >= N) return;
  const float* r = rays + (size_t)n * 6;
  float go[3] = {0, 0, 0}, gd[3] = {0, 0, 0};
  for (int j = lane; j < S; j += 64) {
    const float t = zrow[(size_t)n * S + j];
    float g[3] = {g_xyz[((size_t)n * S + j) * 3 + 0], g_xyz[((size_t)n * S + j) * 3 + 1],
                  g_xyz[((size_t)n * S + j) * 3 + 2]};
    if (ray_type == RDRF_RAY_CONTRACT) {
      float p[3], nrm = 0.f;
      int m = 0;
      for (int k = 0; k < 3; ++k) {
        p[k] = r[k] + r[3 + k] * t;
        if (fabsf(p[k]) > nrm) { nrm = fabsf(p[k]); m = k; }
      }
      if (nrm > 1.0f) {  // pc_i = (2/n - 1/n^2) p_i, n = |p_m|
        const float sc = 2.0f / nrm - 1.0f / (nrm * nrm);
        const float dsc = (-2.0f / (nrm * nrm) + 2.0f / (nrm * nrm * nrm)) * (p[m] > 0.f ? 1.f : -1.f);
        const float gp = g[0] * p[0] + g[1] * p[1] + g[2] * p[2];
        for (int k = 0; k < 3; ++k) g[k] = g[k] * sc;
        g[m] += gp * dsc;
      }
    }
    for (int k = 0; k < 3; ++k) { go[k] += g[k]; gd[k] += g[k] * t; }
  }
  for (int k = 0; k < 3; ++k) { go[k] = wave_sum(go[k]); gd[k] = wave_sum(gd[k]); }
  if (lane == 0)
    for (int k = 0; k < 3; ++k) { g_rays[(size_t)n * 6 + k] += go[k]; g_rays[(size_t)n * 6 + 3 + k] += gd[k]; }
}

extern "C" int rdrf_sample_bwd(const float* rays, const float* z, int N, int S, int ray_type,
                               const float* grad_xyz, float* grad_rays, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(N > 0 && S > 0 && rays && z && grad_xyz && grad_rays, -1, "sample_bwd: bad arguments");
  RDRF_LAUNCH("sample_bwd", k_sample_bwd, dim3(N), dim3(64), stream, rays, z, N, S, ray_type, grad_xyz,
              grad_rays);
  return 0;
}


// ------------------------------------------------------------------------------------------------
// induced flow / disparity (renderer.py:1334-1392): wave per ray.  Forward: strided reduction of
// sum w and sum w*p over the S samples, then ~40 scalar operations per ray.  Backward recomputes the
// reduction (16 B/sample read, 16 B/sample written: HBM-trivial) and differentiates the per-ray
// chain by hand; torch's max() subgradient (the arg-max component) is used for the L-inf norms.
// ------------------------------------------------------------------------------------------------
struct FlowRay {
  float far[3], P[3], world[3], q[3], cam[3];
  float acc, c, pz;            // NDC: clamped P.z and 2/(c-1)
  float fn, fs; int fk;        // contract far point: norm, scale g(n), arg-max component (fk<0: inside)
  float wn, wsc; int wk;       // contract2world: norm, scale s(m), arg-max (wk<0: identity)
};
RDRF_D int argmax_abs3(const float (&v)[3], float& n) {
  int k = 0;
  n = fabsf(v[0]);
  if (fabsf(v[1]) > n) { n = fabsf(v[1]); k = 1; }
  if (fabsf(v[2]) > n) { n = fabsf(v[2]); k = 2; }
  return k;
}
RDRF_D void flow_ray_fwd(FlowRay& r, const float* ray, const float* c2w, float acc, const float (&ps)[3],
                         int H, int W, float f, int ray_type, float& u, float& v, float& disp) {
#pragma clang fp contract(off)
  r.acc = acc;
  if (ray_type == RDRF_RAY_NDC) {
    for (int i = 0; i < 3; ++i) r.far[i] = ray[i] + ray[3 + i];
    r.fk = -1;
  } else {
    float f0[3];
    for (int i = 0; i < 3; ++i) f0[i] = ray[i] + ray[3 + i] * 256.0f;
    const int k = argmax_abs3(f0, r.fn);
    if (r.fn > 1.0f) {
      r.fk = k;
      r.fs = (2.0f - 1.0f / r.fn);
      for (int i = 0; i < 3; ++i) r.far[i] = r.fs * (f0[i] / r.fn);
    } else {
      r.fk = -1;
      for (int i = 0; i < 3; ++i) r.far[i] = f0[i];
    }
  }
  for (int i = 0; i < 3; ++i) r.P[i] = ps[i] + (1.0f - acc) * r.far[i];
  if (ray_type == RDRF_RAY_NDC) {
    r.c = fminf(fmaxf(r.P[2], -1.0f), 1.0f - 1e-6f);
    r.pz = 2.0f / (r.c - 1.0f);
    r.world[0] = -r.P[0] * r.pz * (float)W / 2.0f / f;
    r.world[1] = -r.P[1] * r.pz * (float)H / 2.0f / f;
    r.world[2] = r.pz;
    r.wk = -1;
  } else {
    const int k = argmax_abs3(r.P, r.wn);
    if (r.wn > 1.0f) {
      r.wk = k;
      r.wsc = -1.0f / (r.wn - 2.0f);
      for (int i = 0; i < 3; ++i) r.world[i] = r.P[i] / r.wn * r.wsc;
    } else {
      r.wk = -1;
      for (int i = 0; i < 3; ++i) r.world[i] = r.P[i];
    }
  }
  for (int j = 0; j < 3; ++j) r.q[j] = r.world[j] - c2w[j * 4 + 3];
  for (int i = 0; i < 3; ++i)   // cam_i = sum_j q_j R[j][i]  (w2c = R^T)
    r.cam[i] = r.q[0] * c2w[0 * 4 + i] + r.q[1] * c2w[1 * 4 + i] + r.q[2] * c2w[2 * 4 + i];
  u = r.cam[0] / (-r.cam[2]) * f + (float)W * 0.5f;
  v = -r.cam[1] / (-r.cam[2]) * f + (float)H * 0.5f;
  disp = 1.0f + 2.0f / r.cam[2];
}

__global__ __launch_bounds__(64) void k_induce_flow(int H, int W, const float* __restrict__ focal,
                                                    const float* __restrict__ c2w,
                                                    const float* __restrict__ weights,
                                                    const float* __restrict__ pts,
                                                    const float* __restrict__ pts_2d,
                                                    const float* __restrict__ rays, int N, int S,
                                                    int ray_type, float* __restrict__ flow,
                                                    float* __restrict__ disp) {
  const int n = blockIdx.x, lane = threadIdx.x;
  float acc = 0.f, ps[3] = {0.f, 0.f, 0.f};
  for (int s = lane; s < S; s += 64) {
    const float w = weights[(size_t)n * S + s];
    const float* p = pts + ((size_t)n * S + s) * 3;
    acc += w; ps[0] += w * p[0]; ps[1] += w * p[1]; ps[2] += w * p[2];
  }
  acc = wave_sum(acc); ps[0] = wave_sum(ps[0]); ps[1] = wave_sum(ps[1]); ps[2] = wave_sum(ps[2]);
  if (lane == 0) {
    FlowRay r;
    float u, v, d;
    flow_ray_fwd(r, rays + (size_t)n * 6, c2w + (size_t)n * 12, acc, ps, H, W, focal[0], ray_type, u, v, d);
    flow[(size_t)n * 2 + 0] = u - pts_2d[(size_t)n * 2 + 0];
    flow[(size_t)n * 2 + 1] = v - pts_2d[(size_t)n * 2 + 1];
    disp[n] = d;
  }
}

// one wave per ray; returns d loss / d focal of the ray (wave-uniform)
RDRF_D float induce_flow_bwd_ray(const int n, const int lane, int H, int W, const float* __restrict__ focal,
                                 const float* __restrict__ c2w, const float* __restrict__ weights,
                                 const float* __restrict__ pts, const float* __restrict__ rays, int S, int ray_type,
                                 const float* __restrict__ g_flow, const float* __restrict__ g_disp,
                                 float* __restrict__ g_weights, float* __restrict__ g_pts, float* __restrict__ g_rays,
                                 float* __restrict__ g_c2w) {
  float acc = 0.f, ps[3] = {0.f, 0.f, 0.f};
  for (int s = lane; s < S; s += 64) {
    const float w = weights[(size_t)n * S + s];
    const float* p = pts + ((size_t)n * S + s) * 3;
    acc += w; ps[0] += w * p[0]; ps[1] += w * p[1]; ps[2] += w * p[2];
  }
  acc = wave_sum(acc); ps[0] = wave_sum(ps[0]); ps[1] = wave_sum(ps[1]); ps[2] = wave_sum(ps[2]);
  // every lane runs the (cheap) per-ray chain so that dP / dacc are wave-uniform without shuffles
  const float* ray = rays + (size_t)n * 6;
  const float* M = c2w + (size_t)n * 12;
  const float f = focal[0];
  FlowRay r;
  float u, v, d;
  flow_ray_fwd(r, ray, M, acc, ps, H, W, f, ray_type, u, v, d);
  const float gu = g_flow ? g_flow[(size_t)n * 2 + 0] : 0.f, gv = g_flow ? g_flow[(size_t)n * 2 + 1] : 0.f;
  const float gd = g_disp ? g_disp[n] : 0.f;
  const float c2 = r.cam[2], ic2 = 1.0f / c2;
  float gcam[3];
  gcam[0] = -gu * f * ic2;                      // u = -cam0/cam2 f + W/2
  gcam[1] = gv * f * ic2;                       // v =  cam1/cam2 f + H/2
  gcam[2] = (gu * r.cam[0] - gv * r.cam[1]) * f * ic2 * ic2 - 2.0f * gd * ic2 * ic2;
  float gf = -gu * r.cam[0] * ic2 + gv * r.cam[1] * ic2;
  float gq[3], gM[12];
  for (int j = 0; j < 3; ++j) {
    gq[j] = gcam[0] * M[j * 4 + 0] + gcam[1] * M[j * 4 + 1] + gcam[2] * M[j * 4 + 2];
    for (int i = 0; i < 3; ++i) gM[j * 4 + i] = r.q[j] * gcam[i];
    gM[j * 4 + 3] = 0.f;
  }
  for (int j = 0; j < 3; ++j) gM[j * 4 + 3] = -gq[j];
  float gP[3];
  if (ray_type == RDRF_RAY_NDC) {
    const float sx = (float)W / 2.0f / f, sy = (float)H / 2.0f / f;
    gP[0] = -gq[0] * r.pz * sx;
    gP[1] = -gq[1] * r.pz * sy;
    const float gpz = -gq[0] * r.P[0] * sx - gq[1] * r.P[1] * sy + gq[2];
    gf += -(r.world[0] * gq[0] + r.world[1] * gq[1]) / f;
    const float gc = gpz * (-2.0f / ((r.c - 1.0f) * (r.c - 1.0f)));
    gP[2] = (r.P[2] >= -1.0f && r.P[2] <= 1.0f - 1e-6f) ? gc : 0.f;
  } else if (r.wk >= 0) {
    // world = P * s(m), s = 1/(m (2-m)), m = |P_k|
    const float m = r.wn, sc = r.wsc / m;
    const float ds = (2.0f * m - 2.0f) / ((m * (2.0f - m)) * (m * (2.0f - m)));
    const float dot = gq[0] * r.P[0] + gq[1] * r.P[1] + gq[2] * r.P[2];
    for (int i = 0; i < 3; ++i) gP[i] = gq[i] * sc;
    gP[r.wk] += dot * ds * (r.P[r.wk] >= 0.f ? 1.0f : -1.0f);
  } else {
    for (int i = 0; i < 3; ++i) gP[i] = gq[i];
  }
  const float gacc = -(gP[0] * r.far[0] + gP[1] * r.far[1] + gP[2] * r.far[2]);
  for (int s = lane; s < S; s += 64) {
    const size_t o = (size_t)n * S + s;
    const float w = weights[o];
    const float* p = pts + o * 3;
    if (g_weights) g_weights[o] += gP[0] * p[0] + gP[1] * p[1] + gP[2] * p[2] + gacc;
    if (g_pts) { g_pts[o * 3 + 0] += w * gP[0]; g_pts[o * 3 + 1] += w * gP[1]; g_pts[o * 3 + 2] += w * gP[2]; }
  }
  if (lane == 0) {
    if (g_rays) {
      float gfar[3], go[3], gdr[3];
      for (int i = 0; i < 3; ++i) gfar[i] = (1.0f - acc) * gP[i];
      if (ray_type == RDRF_RAY_NDC) {
        for (int i = 0; i < 3; ++i) { go[i] = gfar[i]; gdr[i] = gfar[i]; }
      } else {
        float g0[3];
        if (r.fk >= 0) {
          // far = g(n) f0, g = (2n-1)/n^2, dg/dn = (2-2n)/n^3, n = |f0_k|
          const float nn = r.fn, gsc = r.fs / nn;
          float f0[3];
          for (int i = 0; i < 3; ++i) f0[i] = ray[i] + ray[3 + i] * 256.0f;
          const float dot = gfar[0] * f0[0] + gfar[1] * f0[1] + gfar[2] * f0[2];
          for (int i = 0; i < 3; ++i) g0[i] = gfar[i] * gsc;
          g0[r.fk] += dot * (2.0f - 2.0f * nn) / (nn * nn * nn) * (f0[r.fk] >= 0.f ? 1.0f : -1.0f);
        } else {
          for (int i = 0; i < 3; ++i) g0[i] = gfar[i];
        }
        for (int i = 0; i < 3; ++i) { go[i] = g0[i]; gdr[i] = 256.0f * g0[i]; }
      }
      for (int i = 0; i < 3; ++i) { g_rays[(size_t)n * 6 + i] += go[i]; g_rays[(size_t)n * 6 + 3 + i] += gdr[i]; }
    }
    if (g_c2w)
      for (int i = 0; i < 12; ++i) g_c2w[(size_t)n * 12 + i] += gM[i];
  }
  return gf;
}

// IFB_WAVES rays per workgroup: the focal length is ONE float that every ray's gradient lands on -- one atomic per ray
// serialised 4096 same-address atomics per launch (33 us of a 4096-ray launch at S = 13, eight launches per iteration of
// the pose-optimising configs); the workgroup's rays are summed in LDS first
#define IFB_WAVES 8
__global__ __launch_bounds__(64 * IFB_WAVES) void k_induce_flow_bwd(int H, int W, const float* __restrict__ focal,
                                                                   const float* __restrict__ c2w,
                                                                   const float* __restrict__ weights,
                                                                   const float* __restrict__ pts,
                                                                   const float* __restrict__ rays, int N, int S,
                                                                   int ray_type, const float* __restrict__ g_flow,
                                                                   const float* __restrict__ g_disp,
                                                                   float* __restrict__ g_weights,
                                                                   float* __restrict__ g_pts,
                                                                   float* __restrict__ g_rays,
                                                                   float* __restrict__ g_c2w,
                                                                   float* __restrict__ g_focal) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.x * IFB_WAVES + wave;
  float gf = 0.f;
  if (n < N) gf = induce_flow_bwd_ray(n, lane, H, W, focal, c2w, weights, pts, rays, S, ray_type, g_flow, g_disp, g_weights,
                                      g_pts, g_rays, g_c2w);
  if (g_focal) {   // (uniform)
    __shared__ float s_gf[IFB_WAVES];
    if (lane == 0) s_gf[wave] = gf;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int i = 0; i < IFB_WAVES; ++i) t += s_gf[i];
      if (t != 0.f) atomicAdd(g_focal, t);
    }
  }
}

extern "C" int rdrf_induce_flow_fwd(int H, int W, const float* focal, const float* c2w,
                                    const float* weights, const float* pts, const float* pts_2d,
                                    const float* rays, int N, int S, int ray_type, float* flow,
                                    float* disp, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(N > 0 && S > 0 && H > 0 && W > 0 && focal && c2w && weights && pts && pts_2d && rays && flow && disp,
             -1, "induce_flow_fwd: bad arguments");
  RDRF_CHECK(ray_type == RDRF_RAY_NDC || ray_type == RDRF_RAY_CONTRACT, -1,
             "induce_flow_fwd: ray_type must be ndc or contract (renderer.py:1343-1361)");
  RDRF_LAUNCH("induce_flow", k_induce_flow, dim3(N), dim3(64), stream, H, W, focal, c2w, weights, pts,
              pts_2d, rays, N, S, ray_type, flow, disp);
  return 0;
}

extern "C" int rdrf_induce_flow_bwd(int H, int W, const float* focal, const float* c2w,
                                    const float* weights, const float* pts, const float* rays, int N,
                                    int S, int ray_type, const float* g_flow, const float* g_disp,
                                    float* g_weights, float* g_pts, float* g_rays, float* g_c2w,
                                    float* g_focal, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(N > 0 && S > 0 && H > 0 && W > 0 && focal && c2w && weights && pts && rays, -1,
             "induce_flow_bwd: bad arguments");
  RDRF_CHECK(ray_type == RDRF_RAY_NDC || ray_type == RDRF_RAY_CONTRACT, -1,
             "induce_flow_bwd: ray_type must be ndc or contract");
  RDRF_CHECK(g_flow || g_disp, -1, "induce_flow_bwd: no output gradient given");
  RDRF_LAUNCH("induce_flow_bwd", k_induce_flow_bwd, dim3((N + IFB_WAVES - 1) / IFB_WAVES), dim3(64 * IFB_WAVES), stream, H, W, focal, c2w, weights,
              pts, rays, N, S, ray_type, g_flow, g_disp, g_weights, g_pts, g_rays, g_c2w, g_focal);
  return 0;
}


// ------------------------------------------------------------------------------------------------
// distortion loss: wave per ray, 64-sample tiles, additive wave scans with carries
// ------------------------------------------------------------------------------------------------
RDRF_D float wave_excl_sum(float v, int lane, float& total) {
  float inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  total = __shfl(inc, 63, 64);
  return inc - v;
}

__global__ __launch_bounds__(64) void k_distloss(const float* __restrict__ w, const float* __restrict__ m,
                                                 float interval, const float* __restrict__ ipt, int N,
                                                 int S, float* __restrict__ loss_ray) {
  const int n = blockIdx.x, lane = threadIdx.x;
  float cw = 0.f, cwm = 0.f, acc = 0.f;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const bool act = s < S;
    const size_t o = (size_t)n * S + (act ? s : 0);
    const float wi = act ? w[o] : 0.f, mi = act ? m[o] : 0.f;
    const float iv = ipt ? (act ? ipt[o] : 0.f) : interval;
    float tw, twm;
    const float pw = cw + wave_excl_sum(wi, lane, tw);
    const float pwm = cwm + wave_excl_sum(wi * mi, lane, twm);
    acc += 2.0f * wi * (mi * pw - pwm) + (1.0f / 3.0f) * iv * wi * wi;
    cw += tw; cwm += twm;
  }
  acc = wave_sum(acc);
  if (lane == 0) loss_ray[n] = acc;
}

__global__ __launch_bounds__(64) void k_distloss_bwd(const float* __restrict__ w, const float* __restrict__ m,
                                                     float interval, const float* __restrict__ ipt,
                                                     int N, int S, const float* __restrict__ g_ray,
                                                     float* __restrict__ g_w) {
  const int n = blockIdx.x, lane = threadIdx.x;
  float tw_all = 0.f, twm_all = 0.f;
  for (int s = lane; s < S; s += 64) {
    const float wi = w[(size_t)n * S + s];
    tw_all += wi; twm_all += wi * m[(size_t)n * S + s];
  }
  tw_all = wave_sum(tw_all); twm_all = wave_sum(twm_all);
  const float g = g_ray[n];
  float cw = 0.f, cwm = 0.f;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const bool act = s < S;
    const size_t o = (size_t)n * S + (act ? s : 0);
    const float wi = act ? w[o] : 0.f, mi = act ? m[o] : 0.f;
    const float iv = ipt ? (act ? ipt[o] : 0.f) : interval;
    float tw, twm;
    const float pw = cw + wave_excl_sum(wi, lane, tw);
    const float pwm = cwm + wave_excl_sum(wi * mi, lane, twm);
    const float qw = tw_all - (pw + wi), qwm = twm_all - (pwm + wi * mi);
    if (act) g_w[o] += g * (2.0f * (mi * (pw - qw) + (qwm - pwm)) + (2.0f / 3.0f) * iv * wi);
    cw += tw; cwm += twm;
  }
}

extern "C" int rdrf_distloss_fwd(const float* w, const float* m, float interval, const float* interval_pt,
                                 int N, int S, float* loss_ray, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(w && m && loss_ray && N > 0 && S > 0, -1, "distloss_fwd: bad arguments");
  RDRF_LAUNCH("distloss", k_distloss, dim3(N), dim3(64), stream, w, m, interval, interval_pt, N, S, loss_ray);
  return 0;
}
extern "C" int rdrf_distloss_bwd(const float* w, const float* m, float interval, const float* interval_pt,
                                 int N, int S, const float* g_ray, float* g_w, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(w && m && g_ray && g_w && N > 0 && S > 0, -1, "distloss_bwd: bad arguments");
  RDRF_LAUNCH("distloss_bwd", k_distloss_bwd, dim3(N), dim3(64), stream, w, m, interval, interval_pt, N, S,
              g_ray, g_w);
  return 0;
}


// ------------------------------------------------------------------------------------------------
// TV regulariser of the VM factors: one launch over all tensors of a factor family.  Elements are
// walked in (slow, fast, c) order with c fastest, which is memory order for the channel-last
// storage: 17 MB of factors are read once per pass (HBM-trivial; the torch formulation is ~16 small
// kernels per tensor per direction).
// ------------------------------------------------------------------------------------------------
struct TvJobs {
  RdrfTensor4 t[RDRF_TV_MAX];
  int n;
};
RDRF_D void tv_decode(const RdrfTensor4& T, long long i, int& c, int& h, int& w, long long& off) {
  c = (int)(i % T.C);
  const long long r = i / T.C;
  if (T.sW <= T.sH) { w = (int)(r % T.W); h = (int)(r / T.W); }
  else { h = (int)(r % T.H); w = (int)(r / T.H); }
  off = c * T.sC + h * T.sH + w * T.sW;
}
__global__ __launch_bounds__(256) void k_tv_fwd(TvJobs J, float* __restrict__ sums) {
  const RdrfTensor4& T = J.t[blockIdx.y];
  const long long total = (long long)T.C * T.H * T.W;
  float sh = 0.f, sw = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c, h, w;
    long long off;
    tv_decode(T, i, c, h, w, off);
    const float v = T.x[off];
    if (h + 1 < T.H) { const float d = T.x[off + T.sH] - v; sh += d * d; }
    if (w + 1 < T.W) { const float d = T.x[off + T.sW] - v; sw += d * d; }
  }
  __shared__ float red[2][4];
  sh = wave_sum(sh); sw = wave_sum(sw);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = sh; red[1][wave] = sw; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(sums + blockIdx.y * 2 + 0, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(sums + blockIdx.y * 2 + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}
__global__ __launch_bounds__(256) void k_tv_bwd(TvJobs J, const float* __restrict__ g_sums) {
  const RdrfTensor4& T = J.t[blockIdx.y];
  const long long total = (long long)T.C * T.H * T.W;
  const float gh = T.H > 1 ? 2.0f * g_sums[blockIdx.y * 2 + 0] : 0.f;
  const float gw = T.W > 1 ? 2.0f * g_sums[blockIdx.y * 2 + 1] : 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c, h, w;
    long long off;
    tv_decode(T, i, c, h, w, off);
    const float v = T.x[off];
    float g = 0.f;
    if (T.H > 1) {
      const float a = h > 0 ? v - T.x[off - T.sH] : 0.f, b = h + 1 < T.H ? T.x[off + T.sH] - v : 0.f;
      g += gh * (a - b);
    }
    if (T.W > 1) {
      const float a = w > 0 ? v - T.x[off - T.sW] : 0.f, b = w + 1 < T.W ? T.x[off + T.sW] - v : 0.f;
      g += gw * (a - b);
    }
    T.g[off] += g;
  }
}
static int tv_jobs(TvJobs& J, const RdrfTensor4* t, int n, bool need_g, long long& maxel) {
  RDRF_CHECK(t && n > 0 && n <= RDRF_TV_MAX, -1, "tv: 1..RDRF_TV_MAX tensors per call");
  J.n = n;
  maxel = 0;
  for (int i = 0; i < n; ++i) {
    RDRF_CHECK(t[i].x && t[i].C > 0 && t[i].H > 0 && t[i].W > 0 && (!need_g || t[i].g), -1, "tv: bad tensor");
    J.t[i] = t[i];
    const long long e = (long long)t[i].C * t[i].H * t[i].W;
    maxel = e > maxel ? e : maxel;
  }
  return 0;
}
extern "C" int rdrf_tv_fwd(const RdrfTensor4* t, int n, float* sums, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TvJobs J;
  long long maxel;
  int rc = tv_jobs(J, t, n, false, maxel);
  if (rc) return rc;
  RDRF_CHECK(sums, -1, "tv_fwd: sums is null");
  RDRF_FILL(sums, 0, sizeof(float) * 2 * n, stream);
  long long gx = (maxel + 256 * 8 - 1) / (256 * 8);
  gx = gx < 1 ? 1 : (gx > 512 ? 512 : gx);
  RDRF_LAUNCH("tv_fwd", k_tv_fwd, dim3((unsigned)gx, n), dim3(256), stream, J, sums);
  return 0;
}
extern "C" int rdrf_tv_bwd(const RdrfTensor4* t, int n, const float* g_sums, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TvJobs J;
  long long maxel;
  int rc = tv_jobs(J, t, n, true, maxel);
  if (rc) return rc;
  RDRF_CHECK(g_sums, -1, "tv_bwd: g_sums is null");
  long long gx = (maxel + 256 * 8 - 1) / (256 * 8);
  gx = gx < 1 ? 1 : (gx > 512 ? 512 : gx);
  RDRF_LAUNCH("tv_bwd", k_tv_bwd, dim3((unsigned)gx, n), dim3(256), stream, J, g_sums);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// TV gradient in ONE pass (no forward sums): the trainer only ever needs d(TV loss)/d(factors) -- the value
// is NaN in the reference (0/0 for the line tensors) and is never used for anything but logging.
// g[c,h,w] += ch * 2 ((x[h]-x[h-1]) - (x[h+1]-x[h])) + cw * 2 ((x[w]-x[w-1]) - (x[w+1]-x[w])) with host-side
// coefficients ch = weight * 2 / (count_h * batch) * family coefficient (cw likewise; 0 for W == 1).
// Channel-last fast path: a thread owns a quad of 4 components of one texel (five 16-byte loads).
// ------------------------------------------------------------------------------------------------
struct TvGradJobs {
  RdrfTensor4 t[RDRF_TV_MAX];
  float ch[RDRF_TV_MAX], cw[RDRF_TV_MAX];
  int n;
};
__global__ __launch_bounds__(256) void k_tv_grad(TvGradJobs J) {
  const RdrfTensor4& T = J.t[blockIdx.y];
  const float gh = T.H > 1 ? 2.0f * J.ch[blockIdx.y] : 0.f, gw = T.W > 1 ? 2.0f * J.cw[blockIdx.y] : 0.f;
  if (T.sC == 1 && (T.C & 3) == 0 && (T.sH & 3) == 0 && (T.sW & 3) == 0) {
    const int cq = T.C >> 2;
    const long long total = (long long)cq * T.H * T.W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
      const int q = (int)(i % cq);
      const long long r = i / cq;
      int h, w;
      if (T.sW <= T.sH) { w = (int)(r % T.W); h = (int)(r / T.W); }
      else { h = (int)(r % T.H); w = (int)(r / T.H); }
      const long long off = 4LL * q + h * T.sH + w * T.sW;
      const f32x4 v = ld4(T.x + off);
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
      if (T.H > 1) {
        if (h > 0) g += (v - ld4(T.x + off - T.sH)) * gh;
        if (h + 1 < T.H) g -= (ld4(T.x + off + T.sH) - v) * gh;
      }
      if (T.W > 1) {
        if (w > 0) g += (v - ld4(T.x + off - T.sW)) * gw;
        if (w + 1 < T.W) g -= (ld4(T.x + off + T.sW) - v) * gw;
      }
      f32x4* gp = (f32x4*)(T.g + off);
      *gp = *gp + g;
    }
    return;
  }
  const long long total = (long long)T.C * T.H * T.W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c, h, w;
    long long off;
    tv_decode(T, i, c, h, w, off);
    const float v = T.x[off];
    float g = 0.f;
    if (T.H > 1) {
      const float a = h > 0 ? v - T.x[off - T.sH] : 0.f, b = h + 1 < T.H ? T.x[off + T.sH] - v : 0.f;
      g += gh * (a - b);
    }
    if (T.W > 1) {
      const float a = w > 0 ? v - T.x[off - T.sW] : 0.f, b = w + 1 < T.W ? T.x[off + T.sW] - v : 0.f;
      g += gw * (a - b);
    }
    T.g[off] += g;
  }
}
extern "C" int rdrf_tv_grad(const RdrfTensor4* t, int n, const float* coef_host, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RDRF_CHECK(t && coef_host && n > 0, -1, "tv_grad: bad arguments");
  for (int i0 = 0; i0 < n; i0 += RDRF_TV_MAX) {
    const int m = n - i0 < RDRF_TV_MAX ? n - i0 : RDRF_TV_MAX;
    TvGradJobs J;
    J.n = m;
    long long maxel = 0;
    for (int i = 0; i < m; ++i) {
      RDRF_CHECK(t[i0 + i].x && t[i0 + i].g && t[i0 + i].C > 0 && t[i0 + i].H > 0 && t[i0 + i].W > 0, -1, "tv_grad: bad tensor");
      J.t[i] = t[i0 + i];
      J.ch[i] = coef_host[2 * (i0 + i)];
      J.cw[i] = coef_host[2 * (i0 + i) + 1];
      const long long e = (long long)t[i0 + i].C * t[i0 + i].H * t[i0 + i].W / 4;
      maxel = e > maxel ? e : maxel;
    }
    long long gx = (maxel + 255) / 256;
    gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
    RDRF_LAUNCH("tv_grad", k_tv_grad, dim3((unsigned)gx, m), dim3(256), stream, J);
  }
  return 0;
}


// ------------------------------------------------------------------------------------------------
// adjoint of a row gather  rows[n] = table[idx[n]]  (the trainer's `allposes_refine[view +- 1]`, train.py:1895-1948: the
// camera matrices of the neighbour frames, live when the poses are optimised): g_table[idx[n]] += g_rows[n].
// torch's index backward sorts the indices and runs a segmented scan (270 us for 4096 rows of 12 floats into a 12-row
// table: two launches per iteration); here each workgroup accumulates in LDS and issues one global atomic per touched
// entry (tables beyond RSA_LDS_FLOATS fall back to global atomics).
// ------------------------------------------------------------------------------------------------
#define RSA_LDS_FLOATS 8192
__global__ __launch_bounds__(256) void k_rows_scatter_add(const int64_t* __restrict__ idx, const float* __restrict__ g_rows, long N,
                                                          int R, int C, float* __restrict__ g_table) {
  __shared__ float s_acc[RSA_LDS_FLOATS];
  const bool in_lds = (long)R * C <= RSA_LDS_FLOATS;   // (uniform)
  if (in_lds) {
    for (int i = threadIdx.x; i < R * C; i += blockDim.x) s_acc[i] = 0.f;
    __syncthreads();
  }
  const long total = N * C;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long n = e / C;
    const int c = (int)(e - n * C);
    const long r = idx[n];
    if (r < 0 || r >= R) continue;   // (torch raises on such an index in the forward gather)
    const float g = g_rows[e];
    if (in_lds) atomicAdd(&s_acc[r * C + c], g);
    else atomicAdd(g_table + r * C + c, g);
  }
  if (in_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < R * C; i += blockDim.x)
      if (s_acc[i] != 0.f) atomicAdd(g_table + i, s_acc[i]);
  }
}

extern "C" int rdrf_rows_scatter_add(const int64_t* idx, const float* g_rows, int N, int R, int C, float* g_table,
                                     rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return 0;
  RDRF_CHECK(N > 0 && R > 0 && C > 0 && idx && g_rows && g_table, -1, "rows_scatter_add: bad arguments");
  long blocks = ((long)N * C + 256 * 8 - 1) / (256 * 8);
  blocks = blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks);
  RDRF_LAUNCH("rows_scatter_add", k_rows_scatter_add, dim3((unsigned)blocks), dim3(256), stream, idx, g_rows, (long)N, R, C, g_table);
  return 0;
}

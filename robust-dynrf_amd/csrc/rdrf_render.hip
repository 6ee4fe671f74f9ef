// rdrf_render.hip -- no-grad render of a ray chunk in one launch sequence (the loop body of
// /root/reference/renderer.py:740-812: sampleXYZ -> static -> dynamic -> raw2outputs), and a
// self-test of the MFMA layer primitive used by tests/.
#include "rdrf_kernels.hpp"

extern "C" size_t rdrf_render_workspace_bytes(int N, int S) {
  const size_t ns = (size_t)N * S;
  // xyz, xyz_prime, rgb_s, rgb_d (3 floats) + 12 scalar planes + valid + 13 outputs + field workspace
  return ns * 4 * (3 * 4 + 12) + ns + (size_t)N * 4 * 16 + rdrf_workspace_bytes(N, S) + (1 << 14);
}

extern "C" int rdrf_render_fwd(const RdrfStaticParams* PS, const RdrfFieldCfg* cfg_s,
                               const RdrfDynamicParams* PD, const RdrfFieldCfg* cfg_d,
                               const float* rays, const float* ts, int N, int S, float near,
                               float far, float* rgb_map, float* depth_map, void* ws, size_t ws_bytes,
                               rdrf_stream_t stream) {
  if (N == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(PS && PD && cfg_s && cfg_d && rays && ts && rgb_map && depth_map && N > 0 && S > 0, -1,
             "render_fwd: bad arguments");
  RDRF_CHECK((size_t)N * S * 3 < (size_t)INT32_MAX, -1, "render_fwd: N * S * 3 must stay below 2^31: render in chunks");
  RDRF_CHECK(ws_bytes >= rdrf_render_workspace_bytes(N, S), -3, "render_fwd: workspace too small");
  WsCarver c(ws, ws_bytes);
  const size_t ns = (size_t)N * S;
  float* xyz = c.take<float>(ns * 3);
  float* z = c.take<float>(ns);
  uint8_t* valid = c.take<uint8_t>(ns);
  float* rgb_s = c.take<float>(ns * 3);
  float* sigma_s = c.take<float>(ns);
  float* weight_s = c.take<float>(ns);
  float* dists_s = c.take<float>(ns);
  float* rgb_d = c.take<float>(ns * 3);
  float* sigma_d = c.take<float>(ns);
  float* weight_d = c.take<float>(ns);
  float* dists_d = c.take<float>(ns);
  float* blending = c.take<float>(ns);
  float* xyz_prime = c.take<float>(ns * 3);
  float* out[13];
  const size_t osz[13] = {0, 0, (size_t)N, ns, (size_t)N * 3, (size_t)N, (size_t)N, ns, (size_t)N * 3,
                          (size_t)N, (size_t)N, ns, (size_t)N};
  for (int i = 0; i < 13; ++i) out[i] = (i < 2) ? nullptr : c.take<float>(osz[i]);
  out[0] = rgb_map;
  out[1] = depth_map;
  void* fws = c.take<char>(rdrf_workspace_bytes(N, S));
  RDRF_CHECK(c.ok(), -3, "render_fwd: workspace too small: need %zu have %zu", c.off, ws_bytes);
  int rc;
  if (cfg_d->ray_type == RDRF_RAY_NDC)
    rc = rdrf_sample_ndc(rays, N, S, near, far, nullptr, cfg_d->aabb, xyz, z, valid, stream);
  else
    rc = rdrf_sample_contract(rays, N, S, near, far, nullptr, nullptr, xyz, z, valid, stream);
  if (rc) return rc;
  rc = rdrf_static_fwd(PS, cfg_s, rays, ts, xyz, z, valid, N, S, rgb_s, sigma_s, weight_s, dists_s,
                       nullptr, 0, fws, rdrf_workspace_bytes(N, S), stream);
  if (rc) return rc;
  rc = rdrf_dynamic_fwd(PD, cfg_d, rays, ts, xyz, z, valid, N, S, blending, weight_d, xyz_prime, rgb_d,
                        sigma_d, dists_d, nullptr, 0, fws, rdrf_workspace_bytes(N, S), stream);
  if (rc) return rc;
  return rdrf_composite_fwd(rgb_s, sigma_s, rgb_d, sigma_d, dists_d, blending, z, rays, N, S,
                            cfg_d->ray_type, 0, out, stream);
}

// ------------------------------------------------------------------------------------------------
// self-test: y[M][64] = relu(x[M][64] W^T + b) through pack + LDS + mfma_seg, one tile per wave
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_selftest(const float* __restrict__ x,
                                                 const float* __restrict__ pkw, int M,
                                                 float* __restrict__ y) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 32 * 64 + 64];
  lds_fill(lds, pkw, 2 * 32 * 64 + 64);
  const int lane = threadIdx.x & 63, h = lane >> 5, s = lane & 31;
  const int row = blockIdx.x * 32 + s;
  float in[32];
#pragma unroll
  for (int kk = 0; kk < 32; ++kk) in[kk] = row < M ? x[(size_t)row * 64 + elem_of(kk, h)] : 0.f;
  f32x16 acc[2];
  acc_bias<2>(acc, lds + 2 * 32 * 64, h);
  mfma_seg<2, 32>(acc, in, lds, lane);
  float out[32];
  acc_relu<2>(out, acc);
  if (row < M)
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) y[(size_t)row * 64 + elem_of(kk, h)] = out[kk];
}

extern "C" int rdrf_selftest_mlp(const float* x, const float* w, const float* b, int M, int K,
                                 int OUT, float* y, void* ws, size_t ws_bytes, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(x && w && b && y && M > 0, -1, "selftest_mlp: bad arguments");
  RDRF_CHECK(K == 64 && OUT == 64, -1, "selftest_mlp: the self-test layer is 64 -> 64");
  RDRF_CHECK(ws_bytes >= (2 * 32 * 64 + 64) * sizeof(float), -3, "selftest_mlp: workspace too small");
  PackJobs J;
  J.n = 0;
  pack_add(J, w, 64, 64, 64, SEG_IDENT, 0, 2, 32, 0);
  pack_add(J, b, 0, 64, 0, 0, 3, 0, 32, 2 * 32 * 64);
  int rc = pack_launch(J, (float*)ws, stream);
  if (rc) return rc;
  RDRF_LAUNCH("selftest", k_selftest, dim3((M + 31) / 32), dim3(64), stream, x, (const float*)ws, M, y);
  return 0;
}

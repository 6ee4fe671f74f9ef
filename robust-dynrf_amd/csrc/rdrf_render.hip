// rdrf_render.hip -- no-grad render of a ray chunk in one launch sequence (the loop body of
// /root/reference/renderer.py:740-812: sampleXYZ -> static -> dynamic -> raw2outputs), and a
// self-test of the MFMA layer primitive used by tests/.
#include "rdrf_misc_dev.hpp"

// host helpers of rdrf_fwd.hip
void fill_common(FieldArgs& a, const RdrfFieldCfg* cfg, const float* rays, const float* ts, const float* xyz, const float* z,
                 const uint8_t* valid, int N, int S);
void fill_static_w(StaticW& w, const RdrfStaticParams* P);
void fill_dyn_w(DynW& w, const RdrfDynamicParams* P);
int ws_carve_fwd(FieldArgs& a, void* ws, size_t ws_bytes, int N, int S, void* saved, size_t saved_bytes, int dynamic);
void dyn_pack_jobs_fwd(PackJobs& J, const RdrfDynamicParams* P);
void static_pack_jobs_fwd(PackJobs& J, const RdrfStaticParams* P, int head);

// ------------------------------------------------------------------------------------------------
// FUSED render: sample -> static + dynamic density phases -> static + dynamic appearance phases -> compositor in ONE
// cooperative launch (/root/reference/renderer.py:740-812 loop body).  The persistent workgroups (one per CU, 8
// waves) walk the phases of the per-phase kernels -- the same device bodies (rdrf_fwd_dev.hpp, rdrf_misc_dev.hpp),
// so the results are those of the launch sequence bit for bit -- separated by grid-wide barriers; the LDS is
// re-filled with the weight image of each MLP phase (121 / 159 / 153 KB: they cannot be resident together).  What
// it buys is the launch sequence itself: 11 stream operations -> 2, which is what a 512-ray chunk (the reference's
// eval chunk, renderer.py:732) spends most of its time on; the per-sample intermediates stay in the L2 / MALL.
// ------------------------------------------------------------------------------------------------
struct RenderFusedArgs {
  FieldArgs as, ad;      // static / dynamic field (outputs + workspaces in the caller's scratch)
  StaticW ws;
  DynW wd;
  CompArgs comp;
  float near, far;
  unsigned* barrier;     // zero at launch
};

// grid-wide barrier of the cooperative launch (all workgroups are resident): a monotonically increasing arrival
// counter; agent-scope fences publish this workgroup's writes (other XCDs have their own L2) and invalidate stale
// lines before the next phase reads what other workgroups wrote.  A bounded spin: a lost workgroup cannot hang the GPU;
// a barrier that times out raises the error word bar[1], every workgroup then leaves at its next barrier and the
// compositor phase is replaced by NaN outputs -- a stalled launch gives a loudly wrong image, never a plausible one.
// Returns false when the launch has failed.
RDRF_D bool grid_barrier(unsigned* bar, unsigned nblk, unsigned& epoch) {
  __syncthreads();
  epoch += 1;
  __shared__ unsigned failed_s;
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(bar, 1u);
    const unsigned target = epoch * nblk;
    unsigned spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > (1u << 26) || __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        atomicExch(bar + 1, 1u);
        break;
      }
    }
    __threadfence();
    failed_s = __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return failed_s == 0u;
}

template <int HEAD>
__global__ __launch_bounds__(64 * RDRF_MAXW) void k_render_fused(RenderFusedArgs r) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const GridCtx gc = grid_ctx();
  unsigned epoch = 0;
  const int N = r.as.N, S = r.as.S;
  // ---- phase 0: sample points, per-ray time branch, clear the per-sample rgb (0 off the appearance masks) and the
  // compaction counters
  if (r.ad.ray_type == RDRF_RAY_NDC)
    sample_ndc_body(r.as.rays, N, S, r.near, r.far, nullptr, r.ad.box, (float*)r.as.xyz, (float*)r.as.z, (uint8_t*)r.as.valid, gc);
  else
    sample_contract_body(r.as.rays, N, S, r.near, r.far, nullptr, nullptr, (float*)r.as.xyz, (float*)r.as.z,
                         (uint8_t*)r.as.valid, gc);
  time_branch_body<true>(r.ad.ts, r.wd, N, r.ad.tout, lds, gc);
  {
    const size_t n3 = (size_t)N * S * 3;
    for (size_t i = (size_t)gc.bid * gc.nthr + gc.tid; i < n3; i += (size_t)gc.nblk * gc.nthr) {
      r.as.rgb[i] = 0.f;
      r.ad.rgb[i] = 0.f;
    }
    if (gc.bid == 0 && gc.tid == 0) { *r.as.counter = 0; *r.ad.counter = 0; }
  }
  bool ok = grid_barrier(r.barrier, gc.nblk, epoch);
  // ---- phase 1: density of both fields (weights, compaction of the appearance masks)
  static_density_body<false, true>(r.as, r.ws, gc);
  dyn_density_body<false, false, true>(r.ad, r.wd, lds, gc);
  ok = grid_barrier(r.barrier, gc.nblk, epoch) && ok;
  // ---- phase 2: appearance of both fields over the compacted lists
  static_app_body<HEAD, false, false, true>(r.as, r.ws, lds, gc);
  __syncthreads();
  dyn_app_body<false, false, true>(r.ad, r.wd, lds, gc);
  ok = grid_barrier(r.barrier, gc.nblk, epoch) && ok;
  // ---- phase 3: raw2outputs per ray
  if (ok) {
    composite_body(r.comp, gc);
  } else {   // a barrier timed out: the phases ran on partial data -- poison this workgroup's share of the image
    const float nan = __int_as_float(0x7fc00000);
    for (int i = gc.bid * gc.nthr + gc.tid; i < N; i += gc.nblk * gc.nthr) {
      r.comp.out[0][i * 3 + 0] = nan; r.comp.out[0][i * 3 + 1] = nan; r.comp.out[0][i * 3 + 2] = nan;
      r.comp.out[1][i] = nan;
    }
  }
}

extern "C" size_t rdrf_render_workspace_bytes(int N, int S) {
  const size_t ns = (size_t)N * S;
  // xyz, xyz_prime, rgb_s, rgb_d (3 floats) + 12 scalar planes + valid + 13 outputs + both fields' workspaces
  return ns * 4 * (3 * 4 + 12) + ns + (size_t)N * 4 * 16 + 2 * rdrf_forward_workspace_bytes(N, S) + (1 << 14);
}

struct RenderBufs {
  float *xyz, *z, *rgb_s, *sigma_s, *weight_s, *dists_s, *rgb_d, *sigma_d, *weight_d, *dists_d, *blending, *xyz_prime;
  uint8_t* valid;
  float* out[13];
  void *fws_s, *fws_d;
  unsigned* barrier;
};
static int carve_render(RenderBufs& b, void* ws, size_t ws_bytes, int N, int S, float* rgb_map, float* depth_map) {
  WsCarver c(ws, ws_bytes);
  const size_t ns = (size_t)N * S;
  b.xyz = c.take<float>(ns * 3);
  b.z = c.take<float>(ns);
  b.valid = c.take<uint8_t>(ns);
  b.rgb_s = c.take<float>(ns * 3);
  b.sigma_s = c.take<float>(ns);
  b.weight_s = c.take<float>(ns);
  b.dists_s = c.take<float>(ns);
  b.rgb_d = c.take<float>(ns * 3);
  b.sigma_d = c.take<float>(ns);
  b.weight_d = c.take<float>(ns);
  b.dists_d = c.take<float>(ns);
  b.blending = c.take<float>(ns);
  b.xyz_prime = c.take<float>(ns * 3);
  const size_t osz[13] = {0, 0, (size_t)N, ns, (size_t)N * 3, (size_t)N, (size_t)N, ns, (size_t)N * 3,
                          (size_t)N, (size_t)N, ns, (size_t)N};
  for (int i = 0; i < 13; ++i) b.out[i] = (i < 2) ? nullptr : c.take<float>(osz[i]);
  b.out[0] = rgb_map;
  b.out[1] = depth_map;
  b.barrier = c.take<unsigned>(64);
  b.fws_s = c.take<char>(rdrf_forward_workspace_bytes(N, S));
  b.fws_d = c.take<char>(rdrf_forward_workspace_bytes(N, S));
  RDRF_CHECK(c.ok(), -3, "render: workspace too small: need %zu have %zu", c.off, ws_bytes);
  return 0;
}

static int render_check(const RdrfStaticParams* PS, const RdrfFieldCfg* cfg_s, const RdrfDynamicParams* PD,
                        const RdrfFieldCfg* cfg_d, const float* rays, const float* ts, int N, int S, float* rgb_map,
                        float* depth_map, size_t ws_bytes) {
  RDRF_CHECK(PS && PD && cfg_s && cfg_d && rays && ts && rgb_map && depth_map && N > 0 && S > 0, -1, "render: bad arguments");
  RDRF_CHECK((size_t)N * S * 3 < (size_t)INT32_MAX, -1, "render: N * S * 3 must stay below 2^31: render in chunks");
  RDRF_CHECK(ws_bytes >= rdrf_render_workspace_bytes(N, S), -3, "render: workspace too small");
  RDRF_CHECK(vm_ok(PS->density, 16, 4) && vm_ok(PS->app, 48, 12) && vm_ok(PD->density, 16, 4) && vm_ok(PD->blending, 16, 4) &&
             vm_ok(PD->app, 48, 12) && vm_same_grid(PD->density, PD->blending), -1,
             "render: only density comps {16,4,4} / app comps {48,12,12} of one grid are built");
  return 0;
}

/* one cooperative launch (see k_render_fused) */
extern "C" int rdrf_render_fused_fwd(const RdrfStaticParams* PS, const RdrfFieldCfg* cfg_s, const RdrfDynamicParams* PD,
                                     const RdrfFieldCfg* cfg_d, const float* rays, const float* ts, int N, int S, float near,
                                     float far, float* rgb_map, float* depth_map, void* ws, size_t ws_bytes,
                                     rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return 0;
  int rc = render_check(PS, cfg_s, PD, cfg_d, rays, ts, N, S, rgb_map, depth_map, ws_bytes);
  if (rc) return rc;
  RenderBufs b;
  rc = carve_render(b, ws, ws_bytes, N, S, rgb_map, depth_map);
  if (rc) return rc;
  RenderFusedArgs r;
  memset(&r, 0, sizeof(r));
  fill_common(r.as, cfg_s, rays, ts, b.xyz, b.z, b.valid, N, S);
  r.as.rgb = b.rgb_s; r.as.sigma = b.sigma_s; r.as.weight = b.weight_s; r.as.dists = b.dists_s;
  rc = ws_carve_fwd(r.as, b.fws_s, rdrf_forward_workspace_bytes(N, S), N, S, nullptr, 0, 0);
  if (rc) return rc;
  fill_common(r.ad, cfg_d, rays, ts, b.xyz, b.z, b.valid, N, S);
  r.ad.rgb = b.rgb_d; r.ad.sigma = b.sigma_d; r.ad.weight = b.weight_d; r.ad.dists = b.dists_d;
  r.ad.blending = b.blending; r.ad.xyz_prime = b.xyz_prime;
  rc = ws_carve_fwd(r.ad, b.fws_d, rdrf_forward_workspace_bytes(N, S), N, S, nullptr, 0, 1);
  if (rc) return rc;
  fill_static_w(r.ws, PS);
  fill_dyn_w(r.wd, PD);
  if (PS->packed_fwd != nullptr) r.as.pk = PS->packed_fwd;
  else {
    PackJobs J;
    static_pack_jobs_fwd(J, PS, cfg_s->static_head);
    rc = pack_launch(J, (float*)r.as.pk, stream);
    if (rc) return rc;
  }
  if (PD->packed_fwd != nullptr) r.ad.pk = PD->packed_fwd;
  else {
    PackJobs J;
    dyn_pack_jobs_fwd(J, PD);
    rc = pack_launch(J, (float*)r.ad.pk, stream);
    if (rc) return rc;
  }
  r.comp.rgb_s = b.rgb_s; r.comp.sigma_s = b.sigma_s; r.comp.rgb_d = b.rgb_d; r.comp.sigma_d = b.sigma_d;
  r.comp.dists = b.dists_d; r.comp.blending = b.blending; r.comp.z = b.z; r.comp.rays = rays;
  r.comp.N = N; r.comp.S = S; r.comp.ray_type = cfg_d->ray_type; r.comp.add_white_bg = 0; r.comp.white_dev = nullptr;
  for (int i = 0; i < 13; ++i) r.comp.out[i] = b.out[i];
  r.near = near; r.far = far; r.barrier = b.barrier;
  RDRF_FILL(b.barrier, 0, 256, stream);
  // one workgroup per CU at most (its LDS holds a whole weight image); fewer when the chunk has fewer units of work
  const long tiles = ((long)N * S + 31) / 32;
  const long units = tiles > N ? tiles : N;
  int ncu = 0, dev = 0;   // of the CURRENT device (a process may drive several)
  RDRF_HIP(hipGetDevice(&dev));
  RDRF_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  if (ncu <= 0) ncu = 256;
  long g = (units + RDRF_MAXW - 1) / RDRF_MAXW;
  g = g < 1 ? 1 : (g > ncu ? ncu : g);
  const size_t lds_bytes = (size_t)(pk::S3_SIZE > pk::K3_SIZE ? (pk::S3_SIZE > pk::K1_SIZE ? pk::S3_SIZE : pk::K1_SIZE)
                                                            : (pk::K3_SIZE > pk::K1_SIZE ? pk::K3_SIZE : pk::K1_SIZE)) * 4;
  const void* kern = cfg_s->static_head == RDRF_HEAD_MLP_FEA ? (const void*)k_render_fused<RDRF_HEAD_MLP_FEA>
                                                            : (const void*)k_render_fused<RDRF_HEAD_MLP_FEA_TIMEEMBEDDING>;
  RDRF_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  void* kargs[] = {(void*)&r};
  rdrf_prof_begin("render_fused", stream);
  hipError_t e = hipLaunchCooperativeKernel(kern, dim3((unsigned)g), dim3(64 * RDRF_MAXW), kargs, (unsigned)lds_bytes, stream);
  rdrf_prof_end("render_fused", stream);
  RDRF_CHECK(e == hipSuccess, -5, "render_fused: cooperative launch failed: %s", hipGetErrorString(e));
  return 0;
}

/* launch sequence of the per-phase kernels (whole frames: every kernel fills the chip, nothing to gain from fusion) */
extern "C" int rdrf_render_sequence_fwd(const RdrfStaticParams* PS, const RdrfFieldCfg* cfg_s, const RdrfDynamicParams* PD,
                           const RdrfFieldCfg* cfg_d, const float* rays, const float* ts, int N, int S, float near,
                           float far, float* rgb_map, float* depth_map, void* ws, size_t ws_bytes, rdrf_stream_t stream) {
  if (N == 0) return 0;
  int rc = render_check(PS, cfg_s, PD, cfg_d, rays, ts, N, S, rgb_map, depth_map, ws_bytes);
  if (rc) return rc;
  RenderBufs b;
  rc = carve_render(b, ws, ws_bytes, N, S, rgb_map, depth_map);
  if (rc) return rc;
  if (cfg_d->ray_type == RDRF_RAY_NDC)
    rc = rdrf_sample_ndc(rays, N, S, near, far, nullptr, cfg_d->aabb, b.xyz, b.z, b.valid, stream);
  else
    rc = rdrf_sample_contract(rays, N, S, near, far, nullptr, nullptr, b.xyz, b.z, b.valid, stream);
  if (rc) return rc;
  // (a side stream for the static density phase under the dynamic field's density kernel does not overlap: the MLP kernels
  // hold all 512 VGPRs of every SIMD, so no other wave becomes resident; measured)
  const size_t fwb = rdrf_forward_workspace_bytes(N, S);
  rc = rdrf_static_fwd(PS, cfg_s, rays, ts, b.xyz, b.z, b.valid, N, S, b.rgb_s, b.sigma_s, b.weight_s, b.dists_s,
                       nullptr, 0, b.fws_s, fwb, stream);
  if (rc) return rc;
  rc = rdrf_dynamic_fwd(PD, cfg_d, rays, ts, b.xyz, b.z, b.valid, N, S, b.blending, b.weight_d, b.xyz_prime, b.rgb_d,
                        b.sigma_d, b.dists_d, nullptr, 0, b.fws_d, fwb, stream);
  if (rc) return rc;
  return rdrf_composite_fwd(b.rgb_s, b.sigma_s, b.rgb_d, b.sigma_d, b.dists_d, b.blending, b.z, rays, N, S,
                            cfg_d->ray_type, 0, nullptr, b.out, stream);
}

// Measured on MI355X (tools/render_bench.py, Balloon1 stage-0 shape, 240 x 135 frame): whole frame 7.7 ms either way (the
// kernels ARE the frame time: their HIP-event sum is 7.9 ms); 512-ray chunks 274 us per chunk as a launch sequence -- the
// seven kernels run back to back, their event times add up to 282 us -- against 378 us for the single cooperative launch
// (static density at 2 waves per SIMD instead of 6, three serial LDS fills, barrier round trips).  What makes small chunks
// slow is not the launch sequence but the wave-per-ray density phase: 512 rays = 512 busy waves of 2048.  The launch
// sequence is what rdrf_render_fwd runs at every size; rdrf_render_fused_fwd is the explicit entry point of the single launch.
extern "C" int rdrf_render_fwd(const RdrfStaticParams* PS, const RdrfFieldCfg* cfg_s,
                               const RdrfDynamicParams* PD, const RdrfFieldCfg* cfg_d, const float* rays,
                               const float* ts, int N, int S, float near, float far, float* rgb_map,
                               float* depth_map, void* ws, size_t ws_bytes, rdrf_stream_t stream) {
  if (N == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  int rc = render_check(PS, cfg_s, PD, cfg_d, rays, ts, N, S, rgb_map, depth_map, ws_bytes);
  if (rc) return rc;
  return rdrf_render_sequence_fwd(PS, cfg_s, PD, cfg_d, rays, ts, N, S, near, far, rgb_map, depth_map, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// The reference's eval loop (renderer.py:740-812: `for chunk_idx in range(N_rays_all // chunk + ...)`, chunk = 512,
// renderer.py:732) as ONE native call: the chunks' launch sequences are issued round-robin on `nstreams` caller streams.
// Why native: issued from Python, one chunk costs ~250 us of host time (struct marshalling + 10 launches) against
// ~190 us of GPU time -- the loop is host-bound; from here a chunk costs its launches.  Why streams: one 512-ray chunk
// fills 64-110 of the 256 CUs (the MLP kernels hold one workgroup per CU: 121-159 KB of LDS), the chunks are
// independent, so several run side by side.  Both packed weight images must be supplied (packed_fwd: they are shared,
// read-only, by every stream); `ws` is cut into one slice of rdrf_render_workspace_bytes(chunk, S) per stream.
// Ordering: every stream first waits for `main_stream` (inputs, weights), `main_stream` finally waits for every stream.
// ------------------------------------------------------------------------------------------------
extern "C" size_t rdrf_render_chunks_workspace_bytes(int chunk, int S, int nstreams) {
  return (size_t)(nstreams < 1 ? 1 : nstreams) * ((rdrf_render_workspace_bytes(chunk, S) + 255) & ~(size_t)255);
}
extern "C" int rdrf_render_chunks_fwd(const RdrfStaticParams* PS, const RdrfFieldCfg* cfg_s, const RdrfDynamicParams* PD,
                                      const RdrfFieldCfg* cfg_d, const float* rays, const float* ts, int N, int S, int chunk,
                                      float near, float far, float* rgb_map, float* depth_map, void* ws, size_t ws_bytes,
                                      rdrf_stream_t main_stream_, const rdrf_stream_t* streams, int nstreams) {
  if (N == 0) return 0;
  RDRF_CHECK(PS && PD && cfg_s && cfg_d && rays && ts && rgb_map && depth_map && ws && chunk > 0 && S > 0, -1,
             "render_chunks: bad arguments");
  RDRF_CHECK(PS->packed_fwd != nullptr && PD->packed_fwd != nullptr, -1,
             "render_chunks: both packed weight images are required (rdrf_static_pack / rdrf_dynamic_pack)");
  RDRF_CHECK(nstreams >= 0 && nstreams <= 16 && (nstreams == 0 || streams != nullptr), -1, "render_chunks: 0..16 streams");
  hipStream_t main_stream = (hipStream_t)main_stream_;
#ifdef RDRF_DETERMINISTIC
  // the deterministic build sorts every compaction list through ONE process-wide scratch (rdrf_sort_ints_inplace): chunks
  // on different streams would race on it (and it may be re-allocated under them), so this build runs the loop in order
  // on the caller's stream -- same bits, no concurrency
  nstreams = 0;
#endif
  const int ns = nstreams < 1 ? 1 : nstreams;
  const size_t slice = (rdrf_render_workspace_bytes(chunk < N ? chunk : N, S) + 255) & ~(size_t)255;
  RDRF_CHECK(ws_bytes >= slice * ns, -3, "render_chunks: workspace too small: need %zu have %zu", slice * ns, ws_bytes);
  {
    // Chunk coalescing (round 6).  The workspace the caller sized for `nstreams` concurrent chunks holds ONE launch sequence
    // over that many chunks' rays just as well, and a ray's result does not depend on which other rays share its launch
    // (per-ray scans, per-sample MFMA columns: tests/test_gpu_forward.py compares chunked and whole-batch renders bit for
    // bit).  A group of g chunks pays the fixed costs of a launch sequence once -- ten launches, two 121-159 KB LDS weight
    // images per CU and MLP kernel -- instead of g times: the frame time of the 512-ray loop was the SUM of its chunks'
    // isolated kernel times (profiles/r05_render_chunk512_kernel_stats.csv), streams or not.
    static const int coalesce = RDRF_ENV("RDRF_CHUNK_COALESCE") ? atoi(RDRF_ENV("RDRF_CHUNK_COALESCE")) : 1;   // 0: one sequence per chunk (tools build)
    int g = 1;
    while (coalesce && g < 256 && (long)g * chunk < N && rdrf_render_workspace_bytes((g + 1) * chunk, S) <= ws_bytes &&
           (size_t)(g + 1) * chunk * S * 3 < (size_t)INT32_MAX)
      ++g;
    if (g > 1) {
      const int super = g * chunk;
      for (int c0 = 0; c0 < N; c0 += super) {
        const int n = N - c0 < super ? N - c0 : super;
        const int rc = rdrf_render_sequence_fwd(PS, cfg_s, PD, cfg_d, rays + (size_t)c0 * 6, ts + c0, n, S, near, far,
                                                rgb_map + (size_t)c0 * 3, depth_map + c0, ws, ws_bytes, main_stream_);
        if (rc) return rc;
      }
      return 0;
    }
  }
  hipEvent_t ev_start = nullptr, ev_done[16];
  if (nstreams >= 1) {
    RDRF_HIP(hipEventCreateWithFlags(&ev_start, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev_start, main_stream);
    for (int k = 0; k < nstreams && e == hipSuccess; ++k) e = hipStreamWaitEvent((hipStream_t)streams[k], ev_start, 0);
    if (e != hipSuccess) {   // nothing was issued on the side streams yet: release the event and report
      (void)hipEventDestroy(ev_start);
      RDRF_CHECK(false, -5, "render_chunks: fork onto the side streams failed: %s", hipGetErrorString(e));
    }
  }
  int rc = 0, k = 0;
  for (int c0 = 0; c0 < N && rc == 0; c0 += chunk, ++k) {
    const int n = N - c0 < chunk ? N - c0 : chunk;
    const int si = k % ns;
    rdrf_stream_t st = nstreams >= 1 ? streams[si] : main_stream_;
    rc = rdrf_render_sequence_fwd(PS, cfg_s, PD, cfg_d, rays + (size_t)c0 * 6, ts + c0, n, S, near, far, rgb_map + (size_t)c0 * 3,
                                  depth_map + c0, (char*)ws + slice * si, slice, st);
  }
  if (nstreams >= 1) {   // join (also on error: the streams must not run past the caller's buffers unobserved)
    for (int q = 0; q < nstreams; ++q) {
      if (hipEventCreateWithFlags(&ev_done[q], hipEventDisableTiming) != hipSuccess) { rc = rc ? rc : -5; continue; }
      (void)hipEventRecord(ev_done[q], (hipStream_t)streams[q]);
      (void)hipStreamWaitEvent(main_stream, ev_done[q], 0);
      (void)hipEventDestroy(ev_done[q]);   // released when the recorded work completes
    }
    (void)hipEventDestroy(ev_start);
  }
  return rc;
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

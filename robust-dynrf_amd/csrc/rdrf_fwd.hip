// rdrf_fwd.hip -- forward kernels of the two fields (TensorBase.forward,
// /root/reference/models/tensorBase.py:704-850) for gfx950.
//
//   static  field: k_static_density (wave per ray: VM gather -> sigma -> wave scan -> weight ->
//                  app-mask compaction) ; k_static_app (32-sample MFMA tiles over the compacted
//                  list: VM gather -> basis -> PE -> 138|135->128->128->3 MLP -> sigmoid)
//   dynamic field: k_time_branch (per ray 17->64->30), k_dyn_density (wave per ray, 32-sample
//                  tiles: warp MLP -> 3-stride VM gathers at the warped point -> density and
//                  blending MLPs -> scan -> weight/compaction), k_dyn_app (MFMA tiles over the
//                  compacted list: 216-feature gather -> basis -> 107->128->128->(131)->3)
//   scene flow   : k_scene_flow (36->64->64->64->6, models/tensoRF.py:446-462)
#include "rdrf_fwd_dev.hpp"

// ------------------------------------------------------------------------------------------------
// per-phase kernels: thin wrappers over the bodies of rdrf_fwd_dev.hpp
// ------------------------------------------------------------------------------------------------
template <bool FEAT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_static_density(FieldArgs a, StaticW w) { static_density_body<FEAT>(a, w, grid_ctx()); }

template <int HEAD, bool FEAT, bool SAVE = true>
__global__ __launch_bounds__(64 * RDRF_MAXW) void k_static_app(FieldArgs a, StaticW w) {
  static_app_body<HEAD, FEAT, SAVE>(a, w, nullptr, grid_ctx());
}

#ifdef RDRF_TOOLS
// 16-sample tiles, sixteen waves per workgroup = four per SIMD (rdrf_fwd_dev.hpp, static_app16_body): tools build only
template <int HEAD, bool SAVE>
__global__ __launch_bounds__(1024) void k_static_app16(FieldArgs a, StaticW w) { static_app16_body<HEAD, SAVE>(a, w); }
#endif

template <bool FEAT, bool SAVE = true>
__global__ __launch_bounds__(64 * RDRF_MAXW) void k_dyn_density(FieldArgs a, DynW w) {
  dyn_density_body<FEAT, SAVE>(a, w, nullptr, grid_ctx());
}

template <bool FEAT, bool SAVE = true>
__global__ __launch_bounds__(64 * RDRF_MAXW) void k_dyn_app(FieldArgs a, DynW w) {
  dyn_app_body<FEAT, SAVE>(a, w, nullptr, grid_ctx());
}

// `zero64`: 64 ints cleared on the way (the append counter of the density phase that follows on the same stream)
__global__ __launch_bounds__(256) void k_time_branch(const float* __restrict__ ts, DynW w, int N,
                                                     float* __restrict__ tout, int* __restrict__ zero64) {
  __shared__ float s_h[8 * 64];
  if (zero64 != nullptr && blockIdx.x == 0 && threadIdx.x < 64) zero64[threadIdx.x] = 0;
  time_branch_body(ts, w, N, tout, s_h, grid_ctx());
}

// inference: the density phase of the dynamic field at tile granularity + the per-ray scan (rdrf_fwd_dev.hpp)
__global__ __launch_bounds__(64 * RDRF_MAXW) void k_dyn_density_flat(FieldArgs a, DynW w) {
  dyn_density_body<false, false, false, true>(a, w, nullptr, grid_ctx());
}
__global__ __launch_bounds__(512) void k_ray_scan(FieldArgs a) { ray_scan_body(a); }   // 16 rays per workgroup

// ------------------------------------------------------------------------------------------------
// scene flow MLP over all N*S samples (models/tensoRF.py:446-462)
// ------------------------------------------------------------------------------------------------
RDRF_D void fill_sf_x(float (&X)[20], float xn0, float xn1, float xn2, float t, int h) {
#pragma unroll
  for (int o = 0; o < 5; ++o) {
    if (o == 0 && h == 0) {
      X[0] = xn0; X[1] = xn1; X[2] = xn2; X[3] = t;
    } else {
      const int k = 2 * o + h - 1;
      // (exact sin / cos for every octave here: the scene-flow losses are L1 terms, and with the doubled-angle form of
      // rdrf_common.hpp sincos_double one sign flip of a ~0 flow component moved the nvidia_late reference-trainer
      // comparison by 8e-4 of a gradient's max -- the kernel is 0.15 ms of the step, the 60 VALU instructions stay)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int pr = 2 * k + p;
        float sv = 0.f, cv = 0.f;
        if (pr < 16) {
          const int d = pr >> 2, f = pr & 3;
          const float x = pr < 12 ? (d == 0 ? xn0 : (d == 1 ? xn1 : xn2)) : t;
          const float a_ = ldexpf(x, f);
          if (__builtin_expect(!(fabsf(a_) <= RDRF_PE_FAST_MAX), 0)) sincosf(a_, &sv, &cv);
          else sincos_sel<true>(a_, sv, cv);
        }
        X[o * 4 + 2 * p] = sv;
        X[o * 4 + 2 * p + 1] = cv;
      }
    }
  }
}

__global__ __launch_bounds__(64 * RDRF_MAXW) void k_scene_flow(const float* __restrict__ pts,
                                                   const float* __restrict__ ts, int N, int S,
                                                   Box box, const float* __restrict__ pkg, DynW w,
                                                   float* __restrict__ sf_f,
                                                   float* __restrict__ sf_b,
                                                   float* __restrict__ act_rows, int dynq) {
  __shared__ __attribute__((aligned(16))) float lds[pk::SF_SIZE];
  __shared__ int s_next;
  if (threadIdx.x == 0) s_next = blockDim.x >> 6;
  lds_fill(lds, pkg + pk::REG_SF, pk::SF_SIZE);
  const int lane = threadIdx.x & 63, h = lane >> 5, s = lane & 31;
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const float* pkw = lds;
  const int total = N * S;
  const int ntiles = (total + 31) >> 5;
  for (int k = wave, tile; (tile = blockIdx.x + k * gridDim.x) < ntiles; k = tile_queue_next(&s_next, k, nwaves, dynq != 0)) {
    const int li = tile * 32 + s;
    const bool act = li < total;
    const int idx = act ? li : 0;
    const float t = ts[idx / S];
    const float xn0 = norm_c(pts[(size_t)idx * 3 + 0], box.lo[0], box.inv[0]);
    const float xn1 = norm_c(pts[(size_t)idx * 3 + 1], box.lo[1], box.inv[1]);
    const float xn2 = norm_c(pts[(size_t)idx * 3 + 2], box.lo[2], box.inv[2]);
    float X[20];
    fill_sf_x(X, xn0, xn1, xn2, t, h);
    float* svb = act_rows ? act_rows + (size_t)tile * sv::SF_ROWS * 32 : nullptr;
    save_rows<20>(svb, sv::SF_X, X, s, h);
    f32x16 acc[2];
    acc_bias<2>(acc, pkw + pk::SF_B0, h);
    mfma_seg<2, 20>(acc, X, pkw + pk::SF_W0, lane);
    float H[32];
    acc_relu<2>(H, acc);
    save_rows<32>(svb, sv::SF_H0, H, s, h);
    acc_bias<2>(acc, pkw + pk::SF_B2, h);
    mfma_seg<2, 32>(acc, H, pkw + pk::SF_W2, lane);
    acc_relu<2>(H, acc);
    save_rows<32>(svb, sv::SF_H2, H, s, h);
    acc_bias<2>(acc, pkw + pk::SF_B4, h);
    mfma_seg<2, 32>(acc, H, pkw + pk::SF_W4, lane);
    acc_relu<2>(H, acc);
    save_rows<32>(svb, sv::SF_H4, H, s, h);
#pragma unroll
    for (int o = 0; o < 6; ++o) {
      const float v = dot_small<32>(H, pkw + pk::SF_W6 + o * 64, h) + w.sfb6[o];
      if (act && h == 0) {
        if (o < 3) sf_f[(size_t)idx * 3 + o] = v;
        else sf_b[(size_t)idx * 3 + (o - 3)] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
void fill_common(FieldArgs& a, const RdrfFieldCfg* cfg, const float* rays, const float* ts,
                        const float* xyz, const float* z, const uint8_t* valid, int N, int S) {
  memset(&a, 0, sizeof(a));
  a.rays = rays; a.ts = ts; a.xyz = xyz; a.z = z; a.valid = valid;
  a.N = N; a.S = S;
  a.box = make_box(cfg);
  a.distance_scale = cfg->distance_scale;
  a.weight_thres = cfg->weight_thres;
  a.density_shift = cfg->density_shift;
  a.act = cfg->act;
  a.ray_type = cfg->ray_type;
  a.static_head = cfg->static_head;
}

void dyn_pack_jobs_fwd(PackJobs& J, const RdrfDynamicParams* P) {
  using namespace pk;
  J.n = 0;
  const int k1 = REG_K1, k3 = REG_K3, sf = REG_SF;
  pack_add(J, P->l3w, 93, 64, 93, SEG_WARP3_X0, 0, 2, 32, k1 + K1_W3_X0);
  pack_add(J, P->l3w, 93, 64, 93, SEG_WARP3_T, 0, 2, 16, k1 + K1_W3_T);
  pack_add(J, P->l4w, 64, 64, 64, SEG_IDENT, 0, 2, 32, k1 + K1_W4);
  pack_add(J, P->l5w, 64, 3, 64, SEG_IDENT, 1, 3, 32, k1 + K1_W5);
#ifdef RDRF_HEADS_F32
  pack_add(J, P->dw1, 152, 64, 72, SEG_IDENT, 0, 2, 36, k1 + K1_DEN1_F);
  pack_add(J, P->dw1, 152, 64, 152, SEG_DEN1_X0, 0, 2, 32, k1 + K1_DEN1_X0);
  pack_add(J, P->dw1, 152, 64, 152, SEG_DEN1_X1, 0, 2, 8, k1 + K1_DEN1_X1);
  pack_add(J, P->bw1, 152, 64, 72, SEG_IDENT, 0, 2, 36, k1 + K1_BLE1_F);
  pack_add(J, P->bw1, 152, 64, 152, SEG_DEN1_X0, 0, 2, 32, k1 + K1_BLE1_X0);
  pack_add(J, P->bw1, 152, 64, 152, SEG_DEN1_X1, 0, 2, 8, k1 + K1_BLE1_X1);
#else
  for (int head = 0; head < 2; ++head) {   // bf16 x 3 image [F 36 | X0 32 | X1 0..3] + the fp32 tail X1 4..7 (pk::K1_DEN1)
    const float* w1 = head == 0 ? P->dw1 : P->bw1;
    const int img = k1 + (head == 0 ? K1_DEN1 : K1_BLE1), tail = k1 + (head == 0 ? K1_DEN1_X1T : K1_BLE1_X1T);
    pack_add_b3(J, w1, 152, 64, 72, SEG_IDENT, 2, 36, 0, 0, K1_HEAD_KK, img);
    pack_add_b3(J, w1, 152, 64, 152, SEG_DEN1_X0, 2, 32, 0, 36, K1_HEAD_KK, img);
    pack_add_b3(J, w1, 152, 64, 152, SEG_DEN1_X1, 2, 4, 0, 68, K1_HEAD_KK, img);
    pack_add_from(J, w1, 152, 64, 152, SEG_DEN1_X1, 2, 4, 4, tail);
  }
#endif
  pack_add(J, P->dw2, 64, 1, 64, SEG_IDENT, 1, 1, 32, k1 + K1_DEN2);
  pack_add(J, P->bw2, 64, 1, 64, SEG_IDENT, 1, 1, 32, k1 + K1_BLE2);
  pack_add(J, P->l3b, 0, 64, 0, 0, 3, 0, 32, k1 + K1_B3);
  pack_add(J, P->l4b, 0, 64, 0, 0, 3, 0, 32, k1 + K1_B4);
  pack_add(J, P->db1, 0, 64, 0, 0, 3, 0, 32, k1 + K1_BD1);
  pack_add(J, P->bb1, 0, 64, 0, 0, 3, 0, 32, k1 + K1_BB1);
  pack_add(J, P->basis, 216, 27, 216, SEG_IDENT, 0, 1, 108, k3 + K3_BASIS);
#ifdef RDRF_APP_F32   // A/B builds: the hidden layers on the fp32 matrix pipe, as up to round 5
  pack_add(J, P->rw1, 107, 128, 107, SEG_RGB1_F, 0, 4, 16, k3 + K3_RGB1_F);
  pack_add(J, P->rw1, 107, 128, 107, SEG_RGB1_X0, 0, 4, 32, k3 + K3_RGB1_X0);
  pack_add(J, P->rw1, 107, 128, 107, SEG_RGB1_X1, 0, 4, 8, k3 + K3_RGB1_X1);
  pack_add(J, P->rw2, 128, 128, 128, SEG_IDENT, 0, 4, 64, k3 + K3_RGB2);
#else                 // bf16 x 3 with split storage (same image offsets and sizes; lo pieces behind the images)
  pack_add_b3s(J, P->rw1, 107, 128, 107, SEG_RGB1_F, 4, 16, 0, 0, 16, k3 + K3_RGB1_F, REG_K3_LO + K3_LO_RGB1_F);
  pack_add_b3s(J, P->rw1, 107, 128, 107, SEG_RGB1_X0, 4, 32, 0, 0, 32, k3 + K3_RGB1_X0, REG_K3_LO + K3_LO_RGB1_X0);
  pack_add_b3s(J, P->rw1, 107, 128, 107, SEG_RGB1_X1, 4, 8, 0, 0, 8, k3 + K3_RGB1_X1, REG_K3_LO + K3_LO_RGB1_X1);
  pack_add_b3s(J, P->rw2, 128, 128, 128, SEG_IDENT, 4, 64, 0, 0, 64, k3 + K3_RGB2, REG_K3_LO + K3_LO_RGB2);
#endif
  pack_add(J, P->rwv, 131, 3, 128, SEG_IDENT, 1, 3, 64, k3 + K3_RGBV);
  pack_add(J, P->rb1, 0, 128, 0, 0, 3, 0, 64, k3 + K3_B1);
  pack_add(J, P->rb2, 0, 128, 0, 0, 3, 0, 64, k3 + K3_B2);
  pack_add(J, P->sfw[0], 36, 64, 36, SEG_SF_X, 0, 2, 20, sf + SF_W0);
  pack_add(J, P->sfw[1], 64, 64, 64, SEG_IDENT, 0, 2, 32, sf + SF_W2);
  pack_add(J, P->sfw[2], 64, 64, 64, SEG_IDENT, 0, 2, 32, sf + SF_W4);
  pack_add(J, P->sfw[3], 64, 6, 64, SEG_IDENT, 1, 6, 32, sf + SF_W6);
  pack_add(J, P->sfb[0], 0, 64, 0, 0, 3, 0, 32, sf + SF_B0);
  pack_add(J, P->sfb[1], 0, 64, 0, 0, 3, 0, 32, sf + SF_B2);
  pack_add(J, P->sfb[2], 0, 64, 0, 0, 3, 0, 32, sf + SF_B4);
}

void static_pack_jobs_fwd(PackJobs& J, const RdrfStaticParams* P, int head) {
  using namespace pk;
  J.n = 0;
  const bool fea = head == RDRF_HEAD_MLP_FEA;
  const int in1 = fea ? 138 : 135;
  pack_add(J, P->basis, 72, 27, 72, SEG_IDENT, 0, 1, 36, REG_S3 + S3_BASIS);
#ifdef RDRF_APP_F32   // A/B builds: the hidden layers on the fp32 matrix pipe, as up to round 5
  pack_add(J, P->w1, in1, 128, in1, fea ? SEG_STAT1_F_FEA : SEG_STAT1_F_TE, 0, 4, 16, REG_S3 + S3_W1_F);
  pack_add(J, P->w1, in1, 128, in1, fea ? SEG_STAT1_P_FEA : SEG_STAT1_P_TE, 0, 4, 64, REG_S3 + S3_W1_P);
  pack_add(J, P->w2, 128, 128, 128, SEG_IDENT, 0, 4, 64, REG_S3 + S3_W2);
#else                 // bf16 x 3 with split storage: hi + mid pieces in the image (same offsets and sizes), lo pieces streamed
  pack_add_b3s(J, P->w1, in1, 128, in1, fea ? SEG_STAT1_F_FEA : SEG_STAT1_F_TE, 4, 16, 0, 0, 16, REG_S3 + S3_W1_F, REG_S3_LO + S3_LO_W1_F);
  pack_add_b3s(J, P->w1, in1, 128, in1, fea ? SEG_STAT1_P_FEA : SEG_STAT1_P_TE, 4, 64, 0, 0, 64, REG_S3 + S3_W1_P, REG_S3_LO + S3_LO_W1_P);
  pack_add_b3s(J, P->w2, 128, 128, 128, SEG_IDENT, 4, 64, 0, 0, 64, REG_S3 + S3_W2, REG_S3_LO + S3_LO_W2);
#endif
  pack_add(J, P->w3, fea ? 128 : 131, 3, 128, SEG_IDENT, 1, 3, 64, REG_S3 + S3_W3);
  pack_add(J, P->b1, 0, 128, 0, 0, 3, 0, 64, REG_S3 + S3_B1);
  pack_add(J, P->b2, 0, 128, 0, 0, 3, 0, 64, REG_S3 + S3_B2);
#ifdef RDRF_TOOLS
  // the same weights as 16x16x4 fragments for the 16-sample-tile kernel (k_static_app16, tools build)
  pack_add(J, P->basis, 72, 27, 72, SEG16_APP_G, 4, 2, 20, REG_S16 + S16_BASIS);
  pack_add(J, P->w1, in1, 128, in1, fea ? SEG_STAT1_F_FEA : SEG_STAT1_F_TE, 4, 8, 8, REG_S16 + S16_W1_F);
  pack_add(J, P->w1, in1, 128, in1, fea ? SEG_STAT1_P_FEA : SEG_STAT1_P_TE, 4, 8, 32, REG_S16 + S16_W1_P);
  pack_add(J, P->w2, 128, 128, 128, SEG_IDENT, 4, 8, 32, REG_S16 + S16_W2);
  pack_add(J, P->w3, fea ? 128 : 131, 3, 128, SEG_IDENT, 5, 3, 32, REG_S16 + S16_W3);
  pack_add(J, P->b1, 0, 128, 0, 0, 6, 0, 32, REG_S16 + S16_B1);
  pack_add(J, P->b2, 0, 128, 0, 0, 6, 0, 32, REG_S16 + S16_B2);
#endif
}

void fill_static_w(StaticW& w, const RdrfStaticParams* P) {
  w.density = P->density; w.app = P->app;
  w.b1 = P->b1; w.b2 = P->b2; w.b3 = P->b3; w.w3 = P->w3;
}
void fill_dyn_w(DynW& w, const RdrfDynamicParams* P) {
  w.density = P->density; w.blending = P->blending; w.app = P->app;
  w.l1w = P->l1w; w.l1b = P->l1b; w.l2w = P->l2w; w.l2b = P->l2b;
  w.l3b = P->l3b; w.l4b = P->l4b; w.l5b = P->l5b;
  w.db1 = P->db1; w.db2 = P->db2; w.bb1 = P->bb1; w.bb2 = P->bb2;
  w.rb1 = P->rb1; w.rb2 = P->rb2; w.rbv = P->rbv; w.rwv = P->rwv;
  w.sfb0 = P->sfb[0]; w.sfb2 = P->sfb[1]; w.sfb4 = P->sfb[2]; w.sfb6 = P->sfb[3];
}

#define PACK_AREA_FLOATS (1 << 20) /* 4 MiB: forward + transposed packs of either field */


int ws_carve_fwd(FieldArgs& a, void* ws, size_t ws_bytes, int N, int S, void* saved,
                 size_t saved_bytes, int dynamic) {
  WsCarver c(ws, ws_bytes);
  size_t ns = (size_t)N * S;
  a.pk = c.take<float>(PACK_AREA_FLOATS);
  a.counter = c.take<int>(64);
  a.tout = c.take<float>((size_t)N * 32);
  a.xw = c.take<float>(ns * 3);
  a.list = c.take<int>(ns);
  RDRF_CHECK(c.ok(), -3, "workspace too small: need %zu have %zu", c.off, ws_bytes);
  if (saved != nullptr) {  // training mode: the backward re-uses these instead of recomputing
    SavedPtrs sp;
    RDRF_CHECK(carve_saved(sp, saved, saved_bytes, dynamic, N, S), -3,
               "saved buffer too small: need %zu have %zu", saved_bytes_field(dynamic, N, S),
               saved_bytes);
    a.counter = &sp.hdr->count;
    a.list = sp.list;
    a.xw = sp.xw;
    a.tout = sp.tout;
    a.raw = sp.raw;
    a.act1 = sp.act1;
    a.act3 = sp.act3;
  }
  return 0;
}

extern "C" size_t rdrf_saved_bytes(int kind, int N, int S) {
  if (kind == 2) return ((size_t)N * S + 31) / 32 * sv::SF_ROWS * 32 * 4 + 256;
  return saved_bytes_field(kind == 1, N, S);
}

extern "C" size_t rdrf_saved_row_bytes(int phase) {
  return (size_t)4 * (phase == 0 ? sv::K1_ROWS : phase == 1 ? sv::K3_ROWS : phase == 2 ? sv::S3_ROWS : sv::SF_ROWS);
}

// persistent launch geometry: one workgroup per CU (its LDS holds the kernel's weight image),
// up to 8 waves per workgroup, each wave walking its own rays / tiles.
struct Geo {
  int grid, block;
};
#ifndef RDRF_GEO_TPW_DEFAULT
#define RDRF_GEO_TPW_DEFAULT 1
#endif
static Geo geo_for_units(long units) {
  Geo g;
  const int ncu = 256;
  int waves = (int)((units + ncu - 1) / ncu);
  waves = waves < 1 ? 1 : (waves > RDRF_MAXW ? RDRF_MAXW : waves);
  g.block = waves * 64;
  // small launches (a 512-ray eval chunk = 1840 tiles): RDRF_GEO_TPW tiles per wave instead of one, on proportionally
  // fewer CUs -- every workgroup pays the 121-159 KB LDS fill of its weight image once per launch, and independent
  // launches on other streams find free CUs (rdrf_render_chunks_fwd)
  static const int tpw_env = RDRF_ENV("RDRF_GEO_TPW") ? atoi(RDRF_ENV("RDRF_GEO_TPW")) : RDRF_GEO_TPW_DEFAULT;
  const long per_block = (long)waves * (tpw_env < 1 ? 1 : tpw_env);
  long blocks = (units + per_block - 1) / per_block;
  g.grid = (int)(blocks < 1 ? 1 : (blocks > ncu ? ncu : blocks));
  return g;
}
static Geo geo_for_tiles(int N, int S) { return geo_for_units(((long)N * S + 31) / 32); }

#ifndef RDRF_DYNQ_DEFAULT
#define RDRF_DYNQ_DEFAULT 1
#endif
#ifndef RDRF_WPRIO_DEFAULT
#define RDRF_WPRIO_DEFAULT 0
#endif
#ifndef RDRF_STAGGER_DEFAULT
#define RDRF_STAGGER_DEFAULT 0
#endif
// FieldArgs::dynq of the forward launches: bit 0 = per-workgroup tile queue (default on: +1-2 % on the compacted-tile MLP
// kernels), bit 1 = distinct wave priorities, bits 8.. = start-up stagger -- the last two are recorded negative experiments
// (profiles/r06_ab_static_app16.txt), reachable in the tools build only
int fwd_dynq() {
  static const int dynq = RDRF_ENV("RDRF_DYNQ") ? atoi(RDRF_ENV("RDRF_DYNQ")) : RDRF_DYNQ_DEFAULT;
  static const int wprio = RDRF_ENV("RDRF_WPRIO") ? atoi(RDRF_ENV("RDRF_WPRIO")) : RDRF_WPRIO_DEFAULT;
  static const int stagger = RDRF_ENV("RDRF_STAGGER") ? atoi(RDRF_ENV("RDRF_STAGGER")) : RDRF_STAGGER_DEFAULT;
  return (dynq ? 1 : 0) | (wprio ? 2 : 0) | (stagger << 8);
}

extern "C" int rdrf_static_fwd(const RdrfStaticParams* P, const RdrfFieldCfg* cfg, const float* rays,
                               const float* ts, const float* xyz, const float* z,
                               const uint8_t* valid, int N, int S, float* rgb, float* sigma,
                               float* weight, float* dists, void* saved, size_t saved_bytes,
                               void* ws, size_t ws_bytes, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(P && cfg && N > 0 && S > 0, -1, "static_fwd: bad arguments");
  RDRF_CHECK((size_t)N * S * 3 < (size_t)INT32_MAX, -1, "static_fwd: N * S * 3 must stay below 2^31 (32-bit sample indices): render / train in smaller chunks");
  RDRF_CHECK(vm_ok(P->density, 16, 4) && vm_ok(P->app, 48, 12), -1,
             "static_fwd: only density comps {16,4,4} / app comps {48,12,12}, the planes and lines of a set spanning one grid, are built");
  FieldArgs a;
  fill_common(a, cfg, rays, ts, xyz, z, valid, N, S);
  a.rgb = rgb; a.sigma = sigma; a.weight = weight; a.dists = dists;
  int rc = ws_carve_fwd(a, ws, ws_bytes, N, S, saved, saved_bytes, 0);
  if (rc) return rc;
  StaticW w;
  fill_static_w(w, P);
  if (P->packed_fwd != nullptr) a.pk = P->packed_fwd;   // caller-packed image (rdrf_static_pack)
  else {
    PackJobs J;
    static_pack_jobs_fwd(J, P, cfg->static_head);
    rc = pack_launch(J, (float*)a.pk, stream);
    if (rc) return rc;
  }
  RDRF_FILL(a.counter, 0, 256, stream);   // (the colours are zero-filled by k_static_density)
#ifdef RDRF_DETERMINISTIC
  RDRF_FILL(a.list, 0x7f, (size_t)N * S * sizeof(int), stream);
#endif
  RDRF_LAUNCH("static_density", k_static_density<false>, dim3((N + 3) / 4), dim3(256), stream, a, w);   // 4 rays per workgroup: one list append
  if (rgb == nullptr) return 0;   // the caller does not consume the colours: the appearance phase is not run
#ifdef RDRF_DETERMINISTIC
  { int rc_ = rdrf_sort_ints_inplace(a.list, (unsigned)((size_t)N * S), stream); if (rc_) return rc_; }   // append order depends on wave timing
#endif
  const Geo g = geo_for_tiles(N, S);
  const bool fea = cfg->static_head == RDRF_HEAD_MLP_FEA;
  a.dynq = fwd_dynq();
#ifdef RDRF_TOOLS
  // k_static_app16 (16-sample tiles, four waves per SIMD) is an EXPERIMENT of the tools build: 4-11 % slower than the 32-sample
  // kernel (profiles/r06_ab_static_app16.txt, r06_pmc_static_app16.csv): RDRF_SA16=1 selects it
  static const int sa16 = RDRF_ENV("RDRF_SA16") ? atoi(RDRF_ENV("RDRF_SA16")) : 0;
  if (sa16) {
    // 16-sample tiles: up to sixteen waves per workgroup, one workgroup per CU (its LDS holds the 159 KB image)
    const long t16 = (((long)N * S + 31) / 32) * 2;
    int waves = (int)((t16 + 255) / 256);
    waves = waves < 1 ? 1 : (waves > 16 ? 16 : waves);
    const long blocks = (t16 + waves - 1) / waves;
    const dim3 gr((unsigned)(blocks > 256 ? 256 : blocks)), bl(waves * 64);
    if (saved != nullptr) {
      if (fea) RDRF_LAUNCH("static_app", (k_static_app16<RDRF_HEAD_MLP_FEA, true>), gr, bl, stream, a, w);
      else RDRF_LAUNCH("static_app", (k_static_app16<RDRF_HEAD_MLP_FEA_TIMEEMBEDDING, true>), gr, bl, stream, a, w);
    } else {
      if (fea) RDRF_LAUNCH("static_app", (k_static_app16<RDRF_HEAD_MLP_FEA, false>), gr, bl, stream, a, w);
      else RDRF_LAUNCH("static_app", (k_static_app16<RDRF_HEAD_MLP_FEA_TIMEEMBEDDING, false>), gr, bl, stream, a, w);
    }
    return 0;
  }
#endif
  if (saved != nullptr) {
    if (fea) RDRF_LAUNCH("static_app", (k_static_app<RDRF_HEAD_MLP_FEA, false, true>), dim3(g.grid), dim3(g.block), stream, a, w);
    else RDRF_LAUNCH("static_app", (k_static_app<RDRF_HEAD_MLP_FEA_TIMEEMBEDDING, false, true>), dim3(g.grid), dim3(g.block), stream, a, w);
  } else {
    if (fea) RDRF_LAUNCH("static_app", (k_static_app<RDRF_HEAD_MLP_FEA, false, false>), dim3(g.grid), dim3(g.block), stream, a, w);
    else RDRF_LAUNCH("static_app", (k_static_app<RDRF_HEAD_MLP_FEA_TIMEEMBEDDING, false, false>), dim3(g.grid), dim3(g.block), stream, a, w);
  }
  return 0;
}

extern "C" int rdrf_dynamic_fwd(const RdrfDynamicParams* P, const RdrfFieldCfg* cfg,
                                const float* rays, const float* ts, const float* xyz,
                                const float* z, const uint8_t* valid, int N, int S,
                                float* blending, float* weight, float* xyz_prime, float* rgb,
                                float* sigma, float* dists, void* saved, size_t saved_bytes,
                                void* ws, size_t ws_bytes, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(P && cfg && N > 0 && S > 0, -1, "dynamic_fwd: bad arguments");
  RDRF_CHECK((size_t)N * S * 3 < (size_t)INT32_MAX, -1, "dynamic_fwd: N * S * 3 must stay below 2^31 (32-bit sample indices): render / train in smaller chunks");
  RDRF_CHECK(vm_ok(P->density, 16, 4) && vm_ok(P->blending, 16, 4) && vm_ok(P->app, 48, 12) && vm_same_grid(P->density, P->blending), -1,
             "dynamic_fwd: only density comps {16,4,4} / app comps {48,12,12}, the planes and lines of a set spanning one grid, are built");
  FieldArgs a;
  fill_common(a, cfg, rays, ts, xyz, z, valid, N, S);
  a.rgb = rgb; a.sigma = sigma; a.weight = weight; a.dists = dists;
  a.blending = blending; a.xyz_prime = xyz_prime;
  a.dynq = fwd_dynq();
  int rc = ws_carve_fwd(a, ws, ws_bytes, N, S, saved, saved_bytes, 1);
  if (rc) return rc;
  DynW w;
  fill_dyn_w(w, P);
  if (P->packed_fwd != nullptr) a.pk = P->packed_fwd;   // caller-packed image (rdrf_dynamic_pack)
  else {
    PackJobs J;
    dyn_pack_jobs_fwd(J, P);
    rc = pack_launch(J, (float*)a.pk, stream);
    if (rc) return rc;
  }
  // inference calls run the density phase at tile granularity (k_dyn_density_flat + k_ray_scan): a 512-ray eval chunk
  // fills the chip and no tile is padded to the end of its ray; training keeps the wave-per-ray kernel, whose saved rows
  // the backward kernels address by (ray, tile)
  static const int flat_env = RDRF_ENV("RDRF_FLAT") ? atoi(RDRF_ENV("RDRF_FLAT")) : 1;   // 0: wave per ray (tools build)
  const bool flat = saved == nullptr && flat_env != 0;
  if (!flat && rgb != nullptr) RDRF_FILL(rgb, 0, (size_t)N * S * 3 * sizeof(float), stream);   // flat: k_ray_scan
  RDRF_LAUNCH("time_branch", k_time_branch, dim3((N + 7) / 8), dim3(256), stream, ts, w, N, a.tout, a.counter);
  const Geo g1 = geo_for_units(N), g3 = geo_for_tiles(N, S);
#ifdef RDRF_DETERMINISTIC
  RDRF_FILL(a.list, 0x7f, (size_t)N * S * sizeof(int), stream);
#endif
  if (saved != nullptr) RDRF_LAUNCH("dyn_density", (k_dyn_density<false, true>), dim3(g1.grid), dim3(g1.block), stream, a, w);
  else if (flat) {
    RDRF_LAUNCH("dyn_density", k_dyn_density_flat, dim3(g3.grid), dim3(g3.block), stream, a, w);
    RDRF_LAUNCH("ray_scan", k_ray_scan, dim3((N + 15) / 16), dim3(512), stream, a);
  } else RDRF_LAUNCH("dyn_density", (k_dyn_density<false, false>), dim3(g1.grid), dim3(g1.block), stream, a, w);
#ifdef RDRF_DETERMINISTIC
  { int rc_ = rdrf_sort_ints_inplace(a.list, (unsigned)((size_t)N * S), stream); if (rc_) return rc_; }
#endif
  if (rgb == nullptr) return 0;   // the caller does not consume the colours: the appearance phase is not run
  if (saved != nullptr) RDRF_LAUNCH("dyn_app", (k_dyn_app<false, true>), dim3(g3.grid), dim3(g3.block), stream, a, w);
  else RDRF_LAUNCH("dyn_app", (k_dyn_app<false, false>), dim3(g3.grid), dim3(g3.block), stream, a, w);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// feature mode: compute_densityfeature / compute_appfeature / compute_blendingfeature /
// warp_coordinate on a batch of M independent points (models/tensoRF.py:118-196, 521-811).  The same
// kernels in pseudo-ray geometry: N = ceil(M/32) rays of 32 samples, time per point.
// ------------------------------------------------------------------------------------------------
extern "C" size_t rdrf_features_saved_bytes(int dynamic, int M) { return saved_bytes_feat(dynamic, M); }
extern "C" size_t rdrf_features_workspace_bytes(int M) {
  const int Np = (M + 31) / 32;
  return rdrf_workspace_bytes(Np, 32) + (size_t)Np * 32 * (32 * 4 + 27 * 4 + 8) + (1 << 14);
}

static int feat_carve_fwd(FieldArgs& a, void* ws, size_t ws_bytes, int M, void* saved, size_t saved_bytes,
                          int dynamic) {
  WsCarver c(ws, ws_bytes);
  const size_t mp = ((size_t)M + 31) / 32 * 32;
  a.pk = c.take<float>(PACK_AREA_FLOATS);
  a.tout = c.take<float>(mp * 32);
  a.xw = c.take<float>(mp * 3);
  RDRF_CHECK(c.ok(), -3, "features: workspace too small: need %zu have %zu", c.off, ws_bytes);
  if (saved != nullptr) {
    SavedPtrs sp;
    RDRF_CHECK(carve_saved_feat(sp, saved, saved_bytes, dynamic, M), -3,
               "features: saved buffer too small: need %zu have %zu", saved_bytes_feat(dynamic, M), saved_bytes);
    a.xw = sp.xw; a.tout = sp.tout; a.raw = sp.raw; a.act1 = sp.act1; a.act3 = sp.act3;
  }
  return 0;
}

extern "C" int rdrf_static_features_fwd(const RdrfStaticParams* P, const RdrfFieldCfg* cfg, const float* xn,
                                        int M, float* density, float* app, void* saved, size_t saved_bytes,
                                        void* ws, size_t ws_bytes, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(P && cfg && xn && M > 0 && (density || app), -1, "static_features_fwd: bad arguments");
  RDRF_CHECK(vm_ok(P->density, 16, 4) && vm_ok(P->app, 48, 12), -1,
             "static_features_fwd: only density comps {16,4,4} / app comps {48,12,12}, the planes and lines of a set spanning one grid, are built");
  const int Np = (M + 31) / 32;
  FieldArgs a;
  fill_common(a, cfg, nullptr, nullptr, xn, nullptr, nullptr, Np, 32);
  a.M = M; a.in_norm = 1; a.sigma = density; a.feat = app;
  int rc = feat_carve_fwd(a, ws, ws_bytes, M, saved, saved_bytes, 0);
  if (rc) return rc;
  StaticW w;
  fill_static_w(w, P);
  if (density != nullptr)
    RDRF_LAUNCH("feat_static_density", k_static_density<true>, dim3(Np), dim3(64), stream, a, w);
  if (app != nullptr) {
    if (P->packed_fwd != nullptr) a.pk = P->packed_fwd;
    else {
      PackJobs J;
      static_pack_jobs_fwd(J, P, cfg->static_head);
      rc = pack_launch(J, (float*)a.pk, stream);
      if (rc) return rc;
    }
    const Geo g = geo_for_units(Np);
    RDRF_LAUNCH("feat_static_app", (k_static_app<RDRF_HEAD_MLP_FEA, true>), dim3(g.grid), dim3(g.block), stream,
                a, w);
  }
  return 0;
}

extern "C" int rdrf_dynamic_features_fwd(const RdrfDynamicParams* P, const RdrfFieldCfg* cfg, const float* x,
                                         const float* t, int M, int x_is_normalized, float* density,
                                         float* blending, float* app, float* xyz_prime, void* saved,
                                         size_t saved_bytes, void* ws, size_t ws_bytes,
                                         rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(P && cfg && x && t && M > 0 && (density || blending || app || xyz_prime), -1,
             "dynamic_features_fwd: bad arguments");
  RDRF_CHECK(vm_ok(P->density, 16, 4) && vm_ok(P->blending, 16, 4) && vm_ok(P->app, 48, 12) && vm_same_grid(P->density, P->blending), -1,
             "dynamic_features_fwd: only density comps {16,4,4} / app comps {48,12,12}, the planes and lines of a set spanning one grid, are built");
  const int Np = (M + 31) / 32;
  FieldArgs a;
  fill_common(a, cfg, nullptr, t, x, nullptr, nullptr, Np, 32);
  a.M = M; a.in_norm = x_is_normalized ? 1 : 0;
  a.sigma = density; a.blending = blending; a.feat = app; a.xyz_prime = xyz_prime;
  int rc = feat_carve_fwd(a, ws, ws_bytes, M, saved, saved_bytes, 1);
  if (rc) return rc;
  DynW w;
  fill_dyn_w(w, P);
  if (P->packed_fwd != nullptr) a.pk = P->packed_fwd;   // caller-packed image (rdrf_dynamic_pack)
  else {
    PackJobs J;
    dyn_pack_jobs_fwd(J, P);
    rc = pack_launch(J, (float*)a.pk, stream);
    if (rc) return rc;
  }
  RDRF_LAUNCH("time_branch", k_time_branch, dim3((M + 7) / 8), dim3(256), stream, t, w, M, a.tout, (int*)nullptr);
  const Geo g = geo_for_units(Np);
  RDRF_LAUNCH("feat_dyn_density", k_dyn_density<true>, dim3(g.grid), dim3(g.block), stream, a, w);
  if (app != nullptr) RDRF_LAUNCH("feat_dyn_app", k_dyn_app<true>, dim3(g.grid), dim3(g.block), stream, a, w);
  return 0;
}

extern "C" int rdrf_scene_flow_fwd(const RdrfDynamicParams* P, const RdrfFieldCfg* cfg,
                                   const float* pts, const float* ts, int N, int S, float* sf_f,
                                   float* sf_b, void* saved, size_t saved_bytes, void* ws,
                                   size_t ws_bytes, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(P && cfg && N > 0 && S > 0, -1, "scene_flow_fwd: bad arguments");
  RDRF_CHECK((size_t)N * S * 3 < (size_t)INT32_MAX, -1, "scene_flow_fwd: N * S * 3 must stay below 2^31 (32-bit sample indices): render / train in smaller chunks");
  FieldArgs a;
  memset(&a, 0, sizeof(a));
  RDRF_CHECK(saved == nullptr || saved_bytes >= rdrf_saved_bytes(2, N, S), -3,
             "scene_flow_fwd: saved buffer too small");
  int rc = ws_carve_fwd(a, ws, ws_bytes, N, S, nullptr, 0, 1);
  if (rc) return rc;
  DynW w;
  fill_dyn_w(w, P);
  if (P->packed_fwd != nullptr) a.pk = P->packed_fwd;   // caller-packed image (rdrf_dynamic_pack)
  else {
    PackJobs J;
    dyn_pack_jobs_fwd(J, P);
    rc = pack_launch(J, (float*)a.pk, stream);
    if (rc) return rc;
  }
  const Geo g = geo_for_tiles(N, S);
  RDRF_LAUNCH("scene_flow", k_scene_flow, dim3(g.grid), dim3(g.block), stream, pts, ts, N, S,
              make_box(cfg), a.pk, w, sf_f, sf_b, (float*)saved, fwd_dynq());
  return 0;
}

// rdrf_kernels.hpp -- argument structs, saved-activation layout and helpers shared by the forward
// (rdrf_fwd.hip) and backward (rdrf_bwd.hip) kernels.
#pragma once
#include "rdrf_host.hpp"

// ------------------------------------------------------------------------------------------------
struct FieldArgs {
  // inputs
  const float* rays;
  const float* ts;
  const float* xyz;
  const float* z;
  const uint8_t* valid;
  int N, S;
  Box box;
  float distance_scale, weight_thres, density_shift;
  int act, ray_type, static_head;
  // outputs
  float *rgb, *sigma, *weight, *dists, *blending, *xyz_prime;
  // workspace
  const float* pk;   // packed weights
  float* tout;       // [N][32]
  float* xw;         // [N][S][3] warped normalised coordinate
  int* list;         // compacted sample ids
  int* counter;
  // training mode: saved activations (nullptr in inference)
  float* act1;
  float* act3;
  float* raw;
  // feature mode (the compute_* / warp_coordinate entry points, template parameter FEAT of the field
  // kernels): the batch is M independent points in pseudo-ray geometry -- N = ceil(M/32) "rays" of
  // S = 32 "samples", idx = point id -- `ts` is indexed PER POINT and every point is live.
  int M;         // number of points
  int in_norm;   // 1: `xyz` holds NORMALISED coordinates (compute_*), 0: un-normalised (warp_coordinate)
  float* feat;   // [M][27] appearance features (basis_mat output) or nullptr
  int dynq;      // 1: the waves of a persistent workgroup draw their tiles from a queue in LDS (tile_queue_next)
};

struct StaticW {
  RdrfVM density, app;
  const float *b1, *b2, *b3, *w3;
};
struct DynW {
  RdrfVM density, blending, app;
  const float *l1w, *l1b, *l2w, *l2b;
  const float *l3b, *l4b, *l5b, *db1, *db2, *bb1, *bb2, *rb1, *rb2, *rbv, *rwv;
  const float *sfb0, *sfb2, *sfb4, *sfb6;
};

RDRF_D float density_act(float f, int act, float shift) {
  return act == RDRF_ACT_RELU ? fmaxf(f, 0.0f) : softplusf_(f + shift);
}

RDRF_D float ray_norm(const float* rays, int n, int ray_type, float& vx, float& vy, float& vz) {
  vx = rays[n * 6 + 3];
  vy = rays[n * 6 + 4];
  vz = rays[n * 6 + 5];
  float nrm = 1.0f;
  if (ray_type != RDRF_RAY_OTHER) {
    nrm = sqrtf(vx * vx + vy * vy + vz * vz);
    vx /= nrm;
    vy /= nrm;
    vz /= nrm;
  }
  return nrm;
}


// Tile queue of a persistent workgroup (round 6).  Workgroup b owns the tiles b, b + G, b + 2 G, ... (G workgroups); its
// waves used to walk them with a static stride.  The SIMD's issue arbitration favours the oldest wave, so the waves of a
// workgroup progress at different speeds: the favoured ones finish their share early and the pipe then runs the
// stragglers alone (measured: matrix pipe 0.82 busy inside the steady state of k_static_app but 0.745 over the CU's busy
// time).  With the queue a wave takes the workgroup's next tile whenever it is done: position k = wave for the first
// tile, then a returning LDS atomic.  s_next must be initialised to the wave count before the first barrier.
RDRF_D int tile_queue_next(int* s_next, int k, int nwaves, bool dyn) {
  if (!dyn) return k + nwaves;
  int v = 0;
  if ((threadIdx.x & 63) == 0) v = atomicAdd(s_next, 1);
  return __builtin_amdgcn_readfirstlane(v);
}

// ------------------------------------------------------------------------------------------------
// Saved activations (training mode).  Every per-sample vector a backward kernel needs is stored
// as rows of 32 samples: [tile][row][32].  A lane of the producing wave (sample s, half h) writes
// its slot kk to row elem_of(kk,h), column s -> two full 128-byte segments per store instruction;
// the dW kernel reads 64 contiguous bytes per lane (row i, samples 16h..16h+15) straight into the
// MFMA operands, so no transpose is ever needed.
// ------------------------------------------------------------------------------------------------
namespace sv {
// dynamic density phase (ray-aligned tiles): rows per tile
constexpr int K1_X0 = 0, K1_X1 = 64, K1_T = 96, K1_H3 = 128, K1_H4 = 192, K1_FD = 256, K1_HD = 352,
              K1_FB = 416, K1_HB = 512, K1_ROWS = 576;
// dynamic appearance phase (compacted tiles)
constexpr int K3_A = 0, K3_F = 224, K3_X0 = 256, K3_X1 = 320, K3_H1 = 352, K3_H2 = 480, K3_VD = 608,
              K3_ROWS = 640;
// static appearance phase
constexpr int S3_G = 0, S3_F = 96, S3_P = 128, S3_H1 = 256, S3_H2 = 384, S3_VD = 512, S3_ROWS = 544;
// scene flow
constexpr int SF_X = 0, SF_H0 = 64, SF_H2 = 128, SF_H4 = 192, SF_ROWS = 256;
// gradient rows written by the backward-data kernels (workspace, same layout)
constexpr int K1G_DZ3 = 0, K1G_DZ4 = 64, K1G_SM = 128, K1G_DZD = 160, K1G_DZB = 224, K1G_DFD = 288,
              K1G_DFB = 384, K1G_DX0 = 480, K1G_ROWS = 544;  // DFD/DFB: d(features) rows for the scatter kernel
constexpr int K3G_DZV = 0, K3G_DZ2 = 32, K3G_DZ1 = 160, K3G_DF = 288, K3G_DA = 320, K3G_ROWS = 544;
constexpr int SFG_DZ6 = 0, SFG_DZ4 = 32, SFG_DZ2 = 96, SFG_DZ0 = 160, SFG_ROWS = 224;
}  // namespace sv

template <int KK>
RDRF_D void save_rows(float* __restrict__ tile_base, int row0, const float (&v)[KK], int s, int h) {
  if (tile_base == nullptr) return;
#ifdef RDRF_ABL_NOSAVE
  return;
#endif
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    // streaming (non-temporal) store: the rows are written once and read back a whole pass later,
    // long after L2 eviction; keeping them out of L2 leaves it to the factor gathers (-15 % on the
    // forward kernels)
#ifdef RDRF_SAVE_TEMPORAL
    tile_base[(size_t)(row0 + elem_of(kk, h)) * 32 + s] = v[kk];
#else
    __builtin_nontemporal_store(v[kk], tile_base + (size_t)(row0 + elem_of(kk, h)) * 32 + s);
#endif
  }
}
template <int KK>
RDRF_D void load_rows(const float* __restrict__ tile_base, int row0, float (&v)[KK], int s, int h) {
#pragma unroll
  // (non-temporal LOADS were measured too: the dW kernel, whose waves share rows through L2, got
  // 10 % slower, the backward-data kernels 2 % faster -- not adopted; re-measured in round 6 against k_dw3's
  // row DMA with -DRDRF_NT_ROWS / -DRDRF_SAVE_TEMPORAL: same outcome, profiles/r06_ab_cache_policies.txt)
#ifdef RDRF_NT_ROWS
  for (int kk = 0; kk < KK; ++kk) v[kk] = __builtin_nontemporal_load(tile_base + (size_t)(row0 + elem_of(kk, h)) * 32 + s);
#else
  for (int kk = 0; kk < KK; ++kk) v[kk] = tile_base[(size_t)(row0 + elem_of(kk, h)) * 32 + s];
#endif
}

// header of the per-call saved buffer (device memory owned by the caller)
struct SavedHdr {
  int count;       // compacted (app-mask) sample count
  int pad[63];
};

struct SavedPtrs {
  SavedHdr* hdr;
  int* list;       // [N*S]
  float* xw;       // [N*S*3]   (dynamic)
  float* tout;     // [N*32]    (dynamic)
  float* raw;      // [N*S*2]   raw density / blending features (dynamic), [N*S] static
  float* act1;     // density-phase rows (dynamic): [N*ceil(S/32)][K1_ROWS][32]
  float* act3;     // appearance-phase rows: [ceil(N*S/32)][K3_ROWS|S3_ROWS][32]
};

// feature mode (rdrf_*_features_*): M points, Mp = 32 * ceil(M/32); per-point time branch
static inline size_t saved_bytes_feat(int dynamic, int M) {
  size_t t = ((size_t)M + 31) / 32, mp = t * 32;
  size_t b = sizeof(SavedHdr) + 256;
  b += mp * 3 * 4 + 256 + mp * 32 * 4 + 256 + mp * 2 * 4 + 256;   // xw, tout, raw
  if (dynamic) b += t * sv::K1_ROWS * 32 * 4 + 256;
  b += t * (dynamic ? sv::K3_ROWS : sv::S3_ROWS) * 32 * 4 + 256;
  return b;
}
static inline bool carve_saved_feat(SavedPtrs& p, void* saved, size_t bytes, int dynamic, int M) {
  WsCarver c(saved, bytes);
  size_t t = ((size_t)M + 31) / 32, mp = t * 32;
  p.hdr = c.take<SavedHdr>(1);
  p.list = nullptr;
  p.xw = c.take<float>(mp * 3);
  p.tout = c.take<float>(mp * 32);
  p.raw = c.take<float>(mp * 2);
  p.act1 = dynamic ? c.take<float>(t * sv::K1_ROWS * 32) : nullptr;
  p.act3 = c.take<float>(t * (dynamic ? sv::K3_ROWS : sv::S3_ROWS) * 32);
  return c.ok();
}
static inline size_t saved_bytes_field(int dynamic, int N, int S) {
  size_t ns = (size_t)N * S, t1 = (size_t)N * ((S + 31) / 32), t3 = (ns + 31) / 32;
  size_t b = sizeof(SavedHdr) + 256;
  b += ns * 4 + 256;                                  // list
  b += ns * 3 * 4 + 256 + (size_t)N * 32 * 4 + 256;   // xw, tout
  b += ns * 2 * 4 + 256;                              // raw
  if (dynamic) b += t1 * sv::K1_ROWS * 32 * 4 + 256;
  b += t3 * (dynamic ? sv::K3_ROWS : sv::S3_ROWS) * 32 * 4 + 256;
  return b;
}
static inline bool carve_saved(SavedPtrs& p, void* saved, size_t bytes, int dynamic, int N, int S) {
  WsCarver c(saved, bytes);
  size_t ns = (size_t)N * S, t1 = (size_t)N * ((S + 31) / 32), t3 = (ns + 31) / 32;
  p.hdr = c.take<SavedHdr>(1);
  p.list = c.take<int>(ns);
  p.xw = c.take<float>(ns * 3);
  p.tout = c.take<float>((size_t)N * 32);
  p.raw = c.take<float>(ns * 2);
  p.act1 = dynamic ? c.take<float>(t1 * sv::K1_ROWS * 32) : nullptr;
  p.act3 = c.take<float>(t3 * (dynamic ? sv::K3_ROWS : sv::S3_ROWS) * 32);
  return c.ok();
}

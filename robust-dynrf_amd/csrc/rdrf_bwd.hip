// rdrf_bwd.hip -- backward of the two fields and of the scene-flow MLP for gfx950.
//
// Structure per phase (appearance / density / scene flow):
//   1. the training-mode forward saved the per-tile activations as [tile][row][32 samples]
//      (rdrf_kernels.hpp, namespace sv);
//   2. a backward-DATA kernel (k_*_bwd) walks the same tiles with the TRANSPOSED weight packs
//      resident in LDS: d_in = W^T dz runs on the fp32 MFMA in the same canonical register layout
//      as the forward (dz of one layer is the B operand of the next), applies the relu masks from
//      the saved rows, back-propagates the positional encodings and the VM gathers (atomic scatter
//      into the channel-last planes/lines + coordinate gradients), and writes every dz as rows;
//   3. k_dw (generic) forms dW = sum_samples dz (x) in on the MFMA: a lane reads 64 contiguous
//      bytes of a row (16 samples) straight into its operand registers -- rows of 32 samples are
//      already the layout the k-contraction over samples needs, so nothing is transposed.
// References: autograd of /root/reference/models/tensorBase.py:704-850, models/tensoRF.py:118-196,
// 446-462, 521-811 (grid_sample backward per SURVEY.md Appendix A).
#ifdef RDRF_GROWS_TEMPORAL   // A/B: the gradient rows this file's kernels write (read back by k_dw3 within the pass) with plain stores
#define RDRF_SAVE_TEMPORAL
#endif
#include "rdrf_kernels.hpp"
#ifdef RDRF_NO_BIAS_ATOMICS
#define BIAS_ATOMIC(p, v) ((void)0)
#else
#define BIAS_ATOMIC(p, v) grad_add(p, v)
#endif

// ------------------------------------------------------------------------------------------------
// backward LDS images (transposed packs + small layers), float offsets inside each region
// ------------------------------------------------------------------------------------------------
namespace pkb {
// dynamic density phase, heads kernel image
constexpr int K1H_DEN2 = 0;                           // small 1 x [2][32]
constexpr int K1H_BLE2 = K1H_DEN2 + 64;
#ifdef RDRF_HEADS_BWD_F32   // A/B builds: the transposed first layers of the heads on the fp32 matrix pipe
constexpr int K1H_DEN1T_F = K1H_BLE2 + 64;            // NBI 3 x KK 32
constexpr int K1H_DEN1T_X0 = K1H_DEN1T_F + 3 * 32 * 64;
constexpr int K1H_BLE1T_F = K1H_DEN1T_X0 + 2 * 32 * 64;
constexpr int K1H_BLE1T_X0 = K1H_BLE1T_F + 3 * 32 * 64;
constexpr int K1H_SIZE = K1H_BLE1T_X0 + 2 * 32 * 64;
#else                       // bf16 x 3 fragments (mfma_seg_b3_pair): 96 dwords per block and slot
constexpr int K1H_DEN1T_F = K1H_BLE2 + 64;            // NBI 3 x KK 32
constexpr int K1H_DEN1T_X0 = K1H_DEN1T_F + 3 * 32 * 96;
constexpr int K1H_BLE1T_F = K1H_DEN1T_X0 + 2 * 32 * 96;
constexpr int K1H_BLE1T_X0 = K1H_BLE1T_F + 3 * 32 * 96;
constexpr int K1H_SIZE = K1H_BLE1T_X0 + 2 * 32 * 96;
#endif
// dynamic density phase, warp kernel image
constexpr int K1W_W5 = 0;                             // small 3 x [2][32]
#ifdef RDRF_HEADS_BWD_F32
constexpr int K1W_W4T = K1W_W5 + 3 * 64;              // NBI 2 x KK 32
constexpr int K1W_W3T_X0 = K1W_W4T + 2 * 32 * 64;     // NBI 2
constexpr int K1W_W3T_T = K1W_W3T_X0 + 2 * 32 * 64;   // NBI 1
constexpr int K1W_SIZE = K1W_W3T_T + 1 * 32 * 64;
#else                       // bf16 x 3 fragments
constexpr int K1W_W4T = K1W_W5 + 3 * 64;              // NBI 2 x KK 32
constexpr int K1W_W3T_X0 = K1W_W4T + 2 * 32 * 96;     // NBI 2
constexpr int K1W_W3T_T = K1W_W3T_X0 + 2 * 32 * 96;   // NBI 1
constexpr int K1W_SIZE = K1W_W3T_T + 1 * 32 * 96;
#endif
// dynamic appearance phase
constexpr int K3_RGBV = 0;                        // small 3 x [2][64]
constexpr int K3_RGB2T = K3_RGBV + 3 * 128;       // NBI 4 x KK 64
constexpr int K3_RGB1T_F = K3_RGB2T + 4 * 64 * 64;   // NBI 1
constexpr int K3_RGB1T_X0 = K3_RGB1T_F + 1 * 64 * 64;  // NBI 2
constexpr int K3_BASIST = K3_RGB1T_X0 + 2 * 64 * 64;  // NBI 7 x KK 16
constexpr int K3_SIZE = K3_BASIST + 7 * 16 * 64;
// static appearance phase
constexpr int S3_W3 = 0;                          // small 3 x [2][64]
constexpr int S3_W2T = S3_W3 + 3 * 128;           // NBI 4 x 64
constexpr int S3_W1T_F = S3_W2T + 4 * 64 * 64;    // NBI 1
constexpr int S3_W1T_P = S3_W1T_F + 1 * 64 * 64;  // NBI 4
constexpr int S3_BASIST = S3_W1T_P + 4 * 64 * 64; // NBI 3 x KK 16
constexpr int S3_SIZE = S3_BASIST + 3 * 16 * 64;
// scene flow
constexpr int SF_W6 = 0;                          // small 6 x [2][32]
constexpr int SF_W4T = SF_W6 + 6 * 64;
constexpr int SF_W2T = SF_W4T + 2 * 32 * 64;
constexpr int SF_W0T = SF_W2T + 2 * 32 * 64;      // NBI 2 (40 -> 64)
constexpr int SF_SIZE = SF_W0T + 2 * 32 * 64;
constexpr int REG_K1H = 0, REG_K1W = REG_K1H + K1H_SIZE, REG_K3 = REG_K1W + K1W_SIZE,
              REG_SF = REG_K3 + K3_SIZE, REG_DYN_END = REG_SF + SF_SIZE;
constexpr int REG_S3 = 0, REG_STAT_END = S3_SIZE;
// lo pieces of the appearance backward kernels' bf16 x 3 layers with split storage (mfma_seg_b3s): streamed, never in LDS
constexpr int K3_LO_RGB2T = 0;                                // 4 x 64 slots
constexpr int K3_LO_RGB1T = K3_LO_RGB2T + 4 * 64 * 32;        // (1 + 2) x 64: the F block, then the two X0 blocks
constexpr int K3_LO_BASIST = K3_LO_RGB1T + 3 * 64 * 32;       // 7 x 16
constexpr int K3_LO_SIZE = K3_LO_BASIST + 7 * 16 * 32;
constexpr int S3_LO_W2T = 0;                                  // 4 x 64
constexpr int S3_LO_W1T = S3_LO_W2T + 4 * 64 * 32;            // (1 + 4) x 64: the F block, then the four PE blocks
constexpr int S3_LO_BASIST = S3_LO_W1T + 5 * 64 * 32;         // 3 x 16
constexpr int S3_LO_SIZE = S3_LO_BASIST + 3 * 16 * 32;
constexpr int REG_K3_LO = REG_DYN_END, REG_S3_LO = REG_STAT_END;
static_assert(K3_RGB1T_X0 == K3_RGB1T_F + 64 * 64 && S3_W1T_P == S3_W1T_F + 64 * 64, "layer-1 blocks form one image");
static_assert(K1H_SIZE * 4 <= 160 * 1024 && K3_SIZE * 4 <= 160 * 1024 && S3_SIZE * 4 <= 160 * 1024,
              "backward weight images must fit the LDS");
}  // namespace pkb

// ------------------------------------------------------------------------------------------------
// argument block of the backward kernels
// ------------------------------------------------------------------------------------------------

// sample-major d(feature) record of the sorted scatter: per factor set 72 floats ordered
// [XY: level 0 (16) | level 1 (16) | level 2 (16)] [XZ: 3 x 4] [YZ: 3 x 4]  (each XY level block is one 64-byte line)
#define DFS_FLOATS 144
RDRF_HD constexpr int dfs_off(int Q) {   // feature quad Q = 6 level + w  (w < 4: XY quad w, 4: XZ, 5: YZ)
  return (Q % 6) < 4 ? (Q / 6) * 16 + 4 * (Q % 6) : ((Q % 6) == 4 ? 48 + 4 * (Q / 6) : 60 + 4 * (Q / 6));
}

// sample-major d(feature) record of the SORTED APPEARANCE scatter: 216 floats per compacted sample, ordered
// [XY: level 0 (48) | level 1 (48) | level 2 (48)] [XZ: 3 x 12] [YZ: 3 x 12]  (an XY level block = three 64-byte lines)
#define DFA_FLOATS 216
RDRF_HD constexpr int dfa_off(int Q) {   // feature quad Q = 18 level + w  (w < 12: XY quad w, 12..14: XZ, 15..17: YZ)
  return (Q % 18) < 12 ? (Q / 18) * 48 + 4 * (Q % 18)
                       : ((Q % 18) < 15 ? 144 + 12 * (Q / 18) + 4 * ((Q % 18) - 12) : 180 + 12 * (Q / 18) + 4 * ((Q % 18) - 15));
}

struct BwdArgs {
  const float* rays;
  const float* ts;
  const float* xyz;
  const float* z;
  const uint8_t* valid;
  int N, S;
  Box box;
  float distance_scale, weight_thres, density_shift;
  int act, ray_type, static_head;
  // upstream gradients (nullable)
  const float *g_rgb, *g_sigma, *g_weight, *g_dists, *g_blending, *g_xyz_prime;
  // saved by the forward
  SavedPtrs sp;
  // packed weights (global) and gradient rows (workspace)
  const float* pk;
  float* grows1;   // density-phase dz rows
  float* grows3;   // appearance-phase dz rows
  float* dxw_app;  // [N*S*3] coordinate grads arriving from the appearance phase
  float* dxn_app;  // [N*S*3]
  float* dtout;    // [N*32]
  float* dfs;      // sorted scatter: d(features) of the density / blending heads SAMPLE-major, [N*S][2][72] in the
                   // order [XY quads of level 0, 1, 2 | XZ quads | YZ quads] (nullptr: row layout for the ray-tile scatter)
  float* dfa;      // sorted appearance scatter: d(app features) per COMPACTED sample, [count][216] (dfa_off); nullptr: rows
  // outputs
  float* g_xyz;
  float* g_rays;   // [N][6] (+=): through dists (ray norm) and the static head's view directions
  float* g_z;      // [N][S] (+=): through dists = (z[j+1] - z[j]) |d| scale (nullable; no reference loss
                   // reaches it, kept for autograd completeness: models/tensorBase.py:726-731)
  // feature mode (template parameter FEAT; see FieldArgs): M points, g_sigma / g_blending carry the
  // gradients of the RAW density / blending features, g_feat [M][27] of the appearance features
  int M, in_norm;
  const float* g_feat;
  int small_dw;   // ray path: k_dyn_density_bwd forms the weight gradients of layer5 / density_layer2 / blending_layer2
  int dynq;       // 1: the compacted-tile kernels draw their tiles from the workgroup's queue (tile_queue_next)
};

struct StaticG {
  RdrfVM density, app;
  float *b3, *w3;
};
struct DynG {
  RdrfVM density, blending, app;
  float *rbv, *rwv, *l5b, *db2, *bb2;
  float *l5w, *dw2, *bw2;   // small layers of the density phase: weight gradients formed in k_dyn_density_bwd (ray path)
};

// ------------------------------------------------------------------------------------------------
// VM gather backward for one quad: scatter into plane / line (atomics) + coordinate gradients
// ------------------------------------------------------------------------------------------------
// DPP lane movement (VALU rate, no LDS crossbar).  ctrl: quad_perm 0x00-0xFF, row_shr:n 0x110+n,
// wave_shr:1 0x138, row_bcast:15 0x142.  Lanes without a valid source read 0.
// bound_ctrl:1 makes lanes without a valid source read 0 WITHOUT a `v_mov dst, 0` preload, and lets
// LLVM's DPP combiner fold the move into the consuming v_add / v_cndmask (one VALU op per step).
template <int CTRL, int ROW_MASK = 0xF>
RDRF_D float dppf(float v) {
  if constexpr (ROW_MASK == 0xF)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
  else  // masked rows keep `old`: pass the lane's own value so that no zero has to be materialised
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF>
RDRF_D int dppi(int v) {
  if constexpr (ROW_MASK == 0xF)
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
  else
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);
}

// fp32 atomics on MI355X: the L2 retires ~20 G atomic REQUESTS/s, where a request is one
// (instruction, <=64-byte line) pair -- not one lane (tools/ubench/atomics.hip: one component per
// lane per instruction 20 G updates/s; 4 adjacent lanes covering a 16-byte quad 83 G/s; 16 lanes on
// a 64-byte texel 322 G/s).  So a quad is never sent as 4 instructions x 1 component: each group of
// 4 adjacent lanes transposes its 4x4 (lane x component) block with two rounds of quad_perm
// exchanges (8 v_cndmask_dpp), so that instruction k carries, in lanes 4t..4t+3, components 0..3 of
// lane 4t+k's quad: one request per live quad.  `off` = float offset from `base` (uniform over the
// 4-lane group; 0xffffffff = nothing to add).  Must be called by ALL lanes of the wave.
template <int K>
RDRF_D void atomic_quad_k(float* base, unsigned off, float val, int c) {
  constexpr int QP = K * 0x55;  // quad_perm:[K,K,K,K]
  const unsigned o = (unsigned)dppi<QP>((int)off);
  if (__ballot(o != 0xffffffffu) == 0ull) return;
  if (o != 0xffffffffu) grad_add(base + (size_t)o + c, val);
}
RDRF_D void atomic_add4(float* p_base, size_t p_off, f32x4 v, bool ok) {
#if defined(RDRF_ABL_NOATOM) || defined(RDRF_ABL_NOGLOBAL)
  return;
#endif
  if (__ballot(ok) == 0ull) return;
  const int lane = threadIdx.x, c = lane & 3;
  const bool a = lane & 1, b = lane & 2;
  // round 1: 2x2 blocks between lanes i and i^1   (quad_perm [1,0,3,2] = 0xB1)
  // (DPP moves are convergent: issue them for ALL lanes, select afterwards)
  const float px = dppf<0xB1>(v.x), py = dppf<0xB1>(v.y), pz = dppf<0xB1>(v.z), pw = dppf<0xB1>(v.w);
  const float n0 = a ? py : v.x, n1 = a ? v.y : px, n2 = a ? pw : v.z, n3 = a ? v.w : pz;
  // round 2: between lanes i and i^2               (quad_perm [2,3,0,1] = 0x4E)
  const float q0 = dppf<0x4E>(n0), q1 = dppf<0x4E>(n1), q2 = dppf<0x4E>(n2), q3 = dppf<0x4E>(n3);
  const float t0 = b ? q2 : n0, t1 = b ? q3 : n1, t2 = b ? n2 : q0, t3 = b ? n3 : q1;
  const unsigned off = ok ? (unsigned)p_off : 0xffffffffu;
  atomic_quad_k<0>(p_base, off, t0, c);
  atomic_quad_k<1>(p_base, off, t1, c);
  atomic_quad_k<2>(p_base, off, t2, c);
  atomic_quad_k<3>(p_base, off, t3, c);
}
RDRF_D float dot4(f32x4 a, f32x4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// Segmented run reduction over the 32 lanes of a half-wave: lanes are consecutive samples of one
// ray, so equal keys (same texel / line entry) form CONTIGUOUS runs.  After the inclusive segmented
// scan the last lane of each run holds the run's sum and is the only one that issues the atomic:
// fp32 L2 atomics sustain only ~10-20 G/s on MI355X and serialise on hot addresses, so combining
// in registers first is worth ~5 DPP steps per value.
struct Run {
  int start;   // first lane (0..31) of the maximal contiguous equal-key stretch this lane is in
  bool tail;   // this lane is the last of its run
};
RDRF_D Run run_of(int key, int s) {
  const int prev = dppi<0x138>(key);  // wave_shr:1
  const bool head = (s == 0) || (prev != key);
  const unsigned long long b = __ballot(head);
  const unsigned m = (unsigned)(b >> (32 * ((threadIdx.x & 63) >> 5)));
  Run r;
  r.start = 31 - __clz((int)(m & (0xffffffffu >> (31 - s))));
  r.tail = (s == 31) || ((m >> (s + 1)) & 1u);
  return r;
}
// inclusive segmented scan over the 32 lanes of a half-wave, entirely in DPP: four row_shr steps
// inside each 16-lane row, then lane 15's row total is added to the lanes of the next row whose
// run started at or before lane 15 (row_bcast:15, written to rows 1 and 3 only).
RDRF_D f32x4 run_scan4(f32x4 v, int start, int s) {
#ifdef RDRF_ABL_NOSCAN
  return v;
#endif
  const int sr = s & 15;
#define RDRF_SCAN_STEP(D)                                                                   \
  {                                                                                         \
    const float ox = dppf<0x110 + D>(v.x), oy = dppf<0x110 + D>(v.y);                       \
    const float oz = dppf<0x110 + D>(v.z), ow = dppf<0x110 + D>(v.w);                       \
    const bool take = sr >= D && s - D >= start;                                            \
    const float tx_ = v.x + ox, ty_ = v.y + oy, tz_ = v.z + oz, tw_ = v.w + ow;             \
    v.x = take ? tx_ : v.x; v.y = take ? ty_ : v.y; v.z = take ? tz_ : v.z; v.w = take ? tw_ : v.w; \
  }
  RDRF_SCAN_STEP(1)
  RDRF_SCAN_STEP(2)
  RDRF_SCAN_STEP(4)
  RDRF_SCAN_STEP(8)
#undef RDRF_SCAN_STEP
  {
    const float ox = dppf<0x142, 0xA>(v.x), oy = dppf<0x142, 0xA>(v.y);
    const float oz = dppf<0x142, 0xA>(v.z), ow = dppf<0x142, 0xA>(v.w);
    const bool take = s >= 16 && start <= 15;
    const float tx_ = v.x + ox, ty_ = v.y + oy, tz_ = v.z + oz, tw_ = v.w + ow;
    v.x = take ? tx_ : v.x; v.y = take ? ty_ : v.y; v.z = take ? tz_ : v.z; v.w = take ? tw_ : v.w;
  }
  return v;
}
RDRF_D bool nz4(f32x4 v) { return v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f; }

// MODE 0: every lane is an independent sample (compacted appearance tiles): plain atomics.
// MODE 1: lanes of a half-wave walk one ray in order: run-reduce first.  ALL lanes of the wave
//         must call (shuffles); `live` = this lane really has a gradient to scatter.
// Line gradients are tiny tensors hammered by every sample (the z line has no runs along a ray), so
// when they fit they are accumulated in LDS (ds_add_f32) by the whole workgroup and flushed to
// global memory once per block: `ll` = LDS accumulator of this factor set or nullptr.
struct LdsLines {
  float* base;   // LDS accumulator (nullptr: scatter straight to global memory); holds doubles when f64 != 0
  int off[3];    // ELEMENT offset of line 0/1/2 inside it
  int f64;       // element type of the accumulator: 1 = double (ds_add_f64), 0 = float (ds_add_f32)
  int direct;    // 1: every live lane adds its own line taps (no run reduction): the sorted passes, where the line index of
                 //    consecutive entries is random and a ds_add_f64 costs less than the DPP scan that would precede it
};
// Element type.  ds_add_f32 is the slowest LDS atomic of gfx950 by an order of magnitude (tools/micro/lds_atomic_rate.hip,
// the access pattern below, 24 waves per CU): 193 cycles per 64-lane instruction = 0.33 lane-updates per clock and CU,
// against 17.9 cycles for ds_add_f64 (3.6 / clk), 10.9 for ds_add_u64, 9.1 for ds_add_u32 and 11.6 for a plain
// ds_write_b32.  So the accumulators are DOUBLES whenever they fit (twice the LDS, 11 x the update rate, and the line
// sums of ~1e5 terms are formed in fp64 before their one conversion to fp32 at the flush); fp32 accumulators remain for
// lines too long for that (final-stage appearance lines in the ray-tile kernel).
// LDS accumulator layout: entry l of a line with C components starts at element l*(C+4): with the natural
// stride (16 floats for C=16) every entry maps to the same two banks and a z-line update (32 distinct
// entries per half-wave) serialises ~16-fold; stride 20 (and 52 for C=48) walks all eight 4-bank
// sets.  Updates use the same quad transposition as the global atomics (4 adjacent lanes = 4
// adjacent banks).
RDRF_D int lds_stride(int C) { return C + 4; }
template <int K>
RDRF_D void lds_quad_k(float* base, int f64, int addr, float val, int c) {
  constexpr int QP = K * 0x55;
  const int ad = dppi<QP>(addr);
  if (__ballot(ad >= 0) == 0ull) return;
  if (ad >= 0) {
    if (f64) __hip_atomic_fetch_add(reinterpret_cast<double*>(base) + ad + c, (double)val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else atomicAdd(base + ad + c, val);
  }
}
// all lanes of the wave must call; `addr` = ELEMENT offset of the lane's quad inside the accumulator `ll`
RDRF_D void lds_add4(const LdsLines& ll, int addr, f32x4 v, bool ok) {
#if defined(RDRF_ABL_NOATOM) || defined(RDRF_ABL_NOLDS)
  return;
#endif
  if (__ballot(ok) == 0ull) return;
  const int lane = threadIdx.x, c = lane & 3;
  const bool a = lane & 1, b = lane & 2;
  // (DPP moves are convergent: issue them for ALL lanes, select afterwards)
  const float px = dppf<0xB1>(v.x), py = dppf<0xB1>(v.y), pz = dppf<0xB1>(v.z), pw = dppf<0xB1>(v.w);
  const float n0 = a ? py : v.x, n1 = a ? v.y : px, n2 = a ? pw : v.z, n3 = a ? v.w : pz;
  const float q0 = dppf<0x4E>(n0), q1 = dppf<0x4E>(n1), q2 = dppf<0x4E>(n2), q3 = dppf<0x4E>(n3);
  const float t0 = b ? q2 : n0, t1 = b ? q3 : n1, t2 = b ? n2 : q0, t3 = b ? n3 : q1;
  const int ad = ok ? addr : -1;
  lds_quad_k<0>(ll.base, ll.f64, ad, t0, c);
  lds_quad_k<1>(ll.base, ll.f64, ad, t1, c);
  lds_quad_k<2>(ll.base, ll.f64, ad, t2, c);
  lds_quad_k<3>(ll.base, ll.f64, ad, t3, c);
}
RDRF_D int lines_floats(const RdrfVM& vm) {
  return vm.L[0] * lds_stride(vm.C[0]) + vm.L[1] * lds_stride(vm.C[1]) + vm.L[2] * lds_stride(vm.C[2]);
}
// all three lines of a factor set, starting at element `first` of the accumulator (ray-tile kernel)
RDRF_D LdsLines make_lds_lines(float* base, int first, int f64, const RdrfVM& vm) {
  LdsLines l;
  l.base = base;
  l.f64 = f64;
  l.direct = 0;
  l.off[0] = first;
  l.off[1] = first + vm.L[0] * lds_stride(vm.C[0]);
  l.off[2] = l.off[1] + vm.L[1] * lds_stride(vm.C[1]);
  return l;
}
// flush `n_entries` x C components of one line (accumulator elements first ..) into its global gradient
RDRF_D void flush_lds_line(const float* acc, int f64, int first, int L, int C, float* __restrict__ gline) {
  const int st = lds_stride(C), n = L * C;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int l = i / C, c = i - l * C;
    const float v = f64 ? (float)reinterpret_cast<const double*>(acc)[first + l * st + c] : acc[first + l * st + c];
    if (v != 0.f) grad_add(gline + i, v);
  }
}
RDRF_D void flush_lds_lines(const float* acc, int f64, int first, const RdrfVM& vm, const RdrfVM& gvm) {
  for (int li = 0; li < 3; ++li) {
    flush_lds_line(acc, f64, first, vm.L[li], vm.C[li], gvm.line[li]);
    first += vm.L[li] * lds_stride(vm.C[li]);
  }
}

// LDS tile of one factor set's gradient PLANE for the chunk of plane cells a workgroup of the tiled sorted scatter is
// working on (k_scatter_tiled): per stride level a small window [y0, y0 + ny) x [x0, x0 + nx) of that level's sub-grid,
// `C` doubles per texel.  A tap inside the window is a ds_add_f64 (18 cycles per 64-lane instruction); a tap outside it
// (entries whose key was clamped, float rounding at a window edge) takes the global atomic as before -- the window only
// decides WHERE a sum is formed, never whether.  geo (LDS, written once per chunk): [lv][0..4] = x0, y0, nx, ny, element
// offset of the level's window inside `base`.
struct PlaneTile {
  float* base;      // LDS, holds doubles (nullptr: no tile)
  const int* geo;   // LDS
  int C;
};
RDRF_D void lds_add4_f64(float* base, int addr, f32x4 v, bool ok) {
  LdsLines l;
  l.base = base; l.f64 = 1; l.direct = 0; l.off[0] = l.off[1] = l.off[2] = 0;
  lds_add4(l, addr, v, ok);
}
template <bool TILED>
RDRF_D void plane_add4(const PlaneTile& T, int lv, int ixs, int iys, int qo, float* GP, size_t goff, f32x4 v, bool ok) {
  if constexpr (!TILED) {
    atomic_add4(GP, goff, v, ok);
  } else {
    // the window of this level: five wave-uniform ints, moved to scalar registers (a VGPR copy per tap would wait for the
    // LDS atomics in front of it: lgkmcnt is in-order)
    const int* g = T.geo + lv * 8;
    const int gx0 = __builtin_amdgcn_readfirstlane(g[0]), gy0 = __builtin_amdgcn_readfirstlane(g[1]);
    const int gnx = __builtin_amdgcn_readfirstlane(g[2]), gny = __builtin_amdgcn_readfirstlane(g[3]);
    const int gof = __builtin_amdgcn_readfirstlane(g[4]);
    const int dx = ixs - gx0, dy = iys - gy0;
    const bool in = ok && (unsigned)dx < (unsigned)gnx && (unsigned)dy < (unsigned)gny;
    lds_add4_f64(T.base, gof + (dy * gnx + dx) * T.C + qo, v, in);
    atomic_add4(GP, goff, v, ok && !in);
  }
}

template <int C0Q, int C1Q, int MODE>
RDRF_D void gather_quad_bwd(const RdrfVM& vm, const RdrfVM& gvm, int g, float x0, float x1,
                            float x2, f32x4 dq, bool live, int s, float& dx0, float& dx1,
                            float& dx2, const LdsLines ll = LdsLines{nullptr, {0, 0, 0}, 0, 0}) {
#ifdef RDRF_ABL_NOGBWD
  dx0 += dq.x; return;
#endif
  QuadSel<C0Q, C1Q> sl = quad_sel<C0Q, C1Q>(g);
  const int pi = sl.pi;
  const float cx = pi == 2 ? x1 : x0;
  const float cy = pi == 0 ? x1 : x2;
  const float cl = pi == 0 ? x2 : (pi == 1 ? x1 : x0);
  const float* P = pi == 0 ? vm.plane[0] : (pi == 1 ? vm.plane[1] : vm.plane[2]);
  const float* Lp = pi == 0 ? vm.line[0] : (pi == 1 ? vm.line[1] : vm.line[2]);
  float* GP = pi == 0 ? gvm.plane[0] : (pi == 1 ? gvm.plane[1] : gvm.plane[2]);
  float* GL = pi == 0 ? gvm.line[0] : (pi == 1 ? gvm.line[1] : gvm.line[2]);
  const int H = pi == 0 ? vm.H[0] : (pi == 1 ? vm.H[1] : vm.H[2]);
  const int W = pi == 0 ? vm.W[0] : (pi == 1 ? vm.W[1] : vm.W[2]);
  const int L = pi == 0 ? vm.L[0] : (pi == 1 ? vm.L[1] : vm.L[2]);
  const int sH = pi == 0 ? vm.sH[0] : (pi == 1 ? vm.sH[1] : vm.sH[2]);
  const int sW = pi == 0 ? vm.sW[0] : (pi == 1 ? vm.sW[1] : vm.sW[2]);
  const int lv = sl.level, st = 1 << lv;
  const int Ws = (W + st - 1) >> lv, Hs = (H + st - 1) >> lv, Ls = (L + st - 1) >> lv;
  Tap1 tx = tap1d(cx, Ws), ty = tap1d(cy, Hs), tl = tap1d(cl, Ls);
  const int C = sl.C, qo = 4 * sl.q;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const size_t o00 = (size_t)((ty.i0 << lv) * sH + (tx.i0 << lv) * sW) + qo;
  const size_t o01 = (size_t)((ty.i0 << lv) * sH + ((tx.i0 + 1) << lv) * sW) + qo;
  const size_t o10 = (size_t)(((ty.i0 + 1) << lv) * sH + (tx.i0 << lv) * sW) + qo;
  const size_t o11 = (size_t)(((ty.i0 + 1) << lv) * sH + ((tx.i0 + 1) << lv) * sW) + qo;
  const bool k00 = live && ty.ok0 && tx.ok0, k01 = live && ty.ok0 && tx.ok1,
             k10 = live && ty.ok1 && tx.ok0, k11 = live && ty.ok1 && tx.ok1;
  const bool m0 = live && tl.ok0, m1 = live && tl.ok1;
  // unconditional loads from clamped addresses (no per-tap branch + wait); out-of-range taps are
  // zeroed afterwards, exactly like zero padding
  const int x0c = min(max(tx.i0, 0), Ws - 1) << lv, x1c = min(max(tx.i0 + 1, 0), Ws - 1) << lv;
  const int y0c = min(max(ty.i0, 0), Hs - 1) << lv, y1c = min(max(ty.i0 + 1, 0), Hs - 1) << lv;
  const int l0c = min(max(tl.i0, 0), Ls - 1) << lv, l1c = min(max(tl.i0 + 1, 0), Ls - 1) << lv;
  f32x4 v00 = ld4(P + (size_t)(y0c * sH + x0c * sW) + qo), v01 = ld4(P + (size_t)(y0c * sH + x1c * sW) + qo);
  f32x4 v10 = ld4(P + (size_t)(y1c * sH + x0c * sW) + qo), v11 = ld4(P + (size_t)(y1c * sH + x1c * sW) + qo);
  const size_t l0 = (size_t)(tl.i0 << lv) * C + qo, l1 = (size_t)((tl.i0 + 1) << lv) * C + qo;
  f32x4 a0 = ld4(Lp + (size_t)l0c * C + qo), a1 = ld4(Lp + (size_t)l1c * C + qo);
  if (!k00) v00 = zero;
  if (!k01) v01 = zero;
  if (!k10) v10 = zero;
  if (!k11) v11 = zero;
  if (!m0) a0 = zero;
  if (!m1) a1 = zero;
  const f32x4 pv = v00 * (tx.w0 * ty.w0) + v01 * (tx.w1 * ty.w0) + v10 * (tx.w0 * ty.w1) +
                   v11 * (tx.w1 * ty.w1);
  const f32x4 lvv = a0 * tl.w0 + a1 * tl.w1;
  const f32x4 dp = live ? dq * lvv : zero;  // grad wrt the interpolated plane quad
  const f32x4 dl = live ? dq * pv : zero;   // grad wrt the interpolated line quad
  // coordinate gradients (grid_sampler_2d_backward: piecewise-linear in the fractional part), before the
  // atomics: see gather_xy4_bwd
  const float gcx = 0.5f * (float)(Ws - 1) * dot4(dp, (v01 - v00) * ty.w0 + (v11 - v10) * ty.w1);
  const float gcy = 0.5f * (float)(Hs - 1) * dot4(dp, (v10 - v00) * tx.w0 + (v11 - v01) * tx.w1);
  const float gcl = 0.5f * (float)(Ls - 1) * dot4(dl, a1 - a0);
  if (MODE == 0) {
    atomic_add4(GP, o00, dp * (tx.w0 * ty.w0), k00);
    atomic_add4(GP, o01, dp * (tx.w1 * ty.w0), k01);
    atomic_add4(GP, o10, dp * (tx.w0 * ty.w1), k10);
    atomic_add4(GP, o11, dp * (tx.w1 * ty.w1), k11);
    atomic_add4(GL, l0, dl * tl.w0, m0);
    atomic_add4(GL, l1, dl * tl.w1, m1);
  } else {
    // the quad / plane selection is uniform over a half-wave, so the (iy, ix) pair keys the run.
    // Keys are purely geometric (a dead sample inside a run contributes zeros, it must not split
    // the run); the run's last lane issues the atomics whatever its own liveness.
    const bool g00 = ty.ok0 && tx.ok0, g01 = ty.ok0 && tx.ok1, g10 = ty.ok1 && tx.ok0,
               g11 = ty.ok1 && tx.ok1;
    const int pkey = ((ty.i0 + 4) << 16) | ((tx.i0 + 4) & 0xffff);
    const Run pr = run_of(pkey, s);
    f32x4 r00 = run_scan4(k00 ? dp * (tx.w0 * ty.w0) : zero, pr.start, s);
    f32x4 r01 = run_scan4(k01 ? dp * (tx.w1 * ty.w0) : zero, pr.start, s);
    const f32x4 r10 = run_scan4(k10 ? dp * (tx.w0 * ty.w1) : zero, pr.start, s);
    const f32x4 r11 = run_scan4(k11 ? dp * (tx.w1 * ty.w1) : zero, pr.start, s);
    // cross-run merge: consecutive runs of a ray almost always differ by ONE texel in x or in y and
    // then share two of their four bilinear taps (same memory locations).  The later run absorbs the
    // earlier run's sums for the shared taps and the earlier run skips those two atomics: ~2 requests
    // per run instead of 4.  Directions (this run relative to the previous one), tap bits 00=1 01=2
    // 10=4 11=8 (first digit = row):   +y: prev.10->00, prev.11->01     -y: prev.00->10, prev.01->11
    //                                  +x: prev.01->00, prev.11->10     -x: prev.00->01, prev.10->11
    // All lanes act on the RAW scanned sums simultaneously, so a tap that a run has itself received
    // must not be forwarded again (multi-hop): the taps moved across a boundary are
    // skip(direction out) & ~receive(direction in of the earlier run).
    auto recv_mask = [](int d) { return d == 65536 ? 3 : (d == -65536 ? 12 : (d == 1 ? 5 : (d == -1 ? 10 : 0))); };
    auto skip_mask = [](int d) { return d == 65536 ? 12 : (d == -65536 ? 3 : (d == 1 ? 10 : (d == -1 ? 5 : 0))); };
    int skip_out = 0;
    {
      const int pl = pr.start > 0 ? pr.start - 1 : 0;           // tail lane of the previous run
      const int pk = __shfl(pkey, pl, 32);
      const int din = pr.start > 0 ? pkey - pk : 0;             // direction INTO this run
      const int pdin = __shfl(din, pl, 32);                     // direction into the previous run
      const int min_ = skip_mask(din) & ~recv_mask(pdin);       // prev-run taps moved into this run
      const int nk = dppi<0x130>(pkey);                         // wave_shl:1 -> key of lane s+1
      const int dout = s < 31 ? nk - pkey : 0;
      skip_out = skip_mask(dout) & ~recv_mask(din);             // own taps the next run takes over
      if (pi == 0) {
        (void)min_;
        const f32x4 hA = dout == 65536 ? ((skip_out & 4) ? r10 : zero) : (dout == -65536 ? ((skip_out & 1) ? r00 : zero)
                       : (dout == 1 ? ((skip_out & 2) ? r01 : zero) : ((skip_out & 1) ? r00 : zero)));
        const f32x4 hB = dout == 65536 ? ((skip_out & 8) ? r11 : zero) : (dout == -65536 ? ((skip_out & 2) ? r01 : zero)
                       : (dout == 1 ? ((skip_out & 8) ? r11 : zero) : ((skip_out & 4) ? r10 : zero)));
        f32x4 pA, pB;
        pA.x = __shfl(hA.x, pl, 32); pA.y = __shfl(hA.y, pl, 32); pA.z = __shfl(hA.z, pl, 32); pA.w = __shfl(hA.w, pl, 32);
        pB.x = __shfl(hB.x, pl, 32); pB.y = __shfl(hB.y, pl, 32); pB.z = __shfl(hB.z, pl, 32); pB.w = __shfl(hB.w, pl, 32);
        f32x4 a00 = zero, a01 = zero, a10 = zero, a11 = zero;
        if (din == 65536) { a00 = pA; a01 = pB; }
        else if (din == -65536) { a10 = pA; a11 = pB; }
        else if (din == 1) { a00 = pA; a10 = pB; }
        else if (din == -1) { a01 = pA; a11 = pB; }
        r00 = r00 + a00; r01 = r01 + a01;
        f32x4 t10 = r10 + a10, t11 = r11 + a11;
        // (r10 / r11 are const above: rebuild the outputs)
        atomic_add4(GP, o00, r00, pr.tail && g00 && nz4(r00) && !(skip_out & 1));
        atomic_add4(GP, o01, r01, pr.tail && g01 && nz4(r01) && !(skip_out & 2));
        atomic_add4(GP, o10, t10, pr.tail && g10 && nz4(t10) && !(skip_out & 4));
        atomic_add4(GP, o11, t11, pr.tail && g11 && nz4(t11) && !(skip_out & 8));
      } else {
        // XZ / YZ (the non-split path of the appearance / static scatter): +y chains only
        const float ux = __shfl(r10.x, pl, 32), uy = __shfl(r10.y, pl, 32), uz = __shfl(r10.z, pl, 32),
                    uw = __shfl(r10.w, pl, 32);
        const float vx_ = __shfl(r11.x, pl, 32), vy_ = __shfl(r11.y, pl, 32), vz_ = __shfl(r11.z, pl, 32),
                    vw_ = __shfl(r11.w, pl, 32);
        if (din == 65536) {
          r00.x += ux; r00.y += uy; r00.z += uz; r00.w += uw;
          r01.x += vx_; r01.y += vy_; r01.z += vz_; r01.w += vw_;
        }
        const bool up_ok = dout != 65536;
        atomic_add4(GP, o00, r00, pr.tail && g00 && nz4(r00));
        atomic_add4(GP, o01, r01, pr.tail && g01 && nz4(r01));
        atomic_add4(GP, o10, r10, pr.tail && up_ok && g10 && nz4(r10));
        atomic_add4(GP, o11, r11, pr.tail && up_ok && g11 && nz4(r11));
      }
    }
    f32x4 r;
    const Run lr = run_of(tl.i0 + 4, s);
    const bool LL = ll.base != nullptr;
    const int lo_ = pi == 0 ? ll.off[0] : (pi == 1 ? ll.off[1] : ll.off[2]);
    const int lst = lds_stride(C);
    r = run_scan4(m0 ? dl * tl.w0 : zero, lr.start, s);
    {
      const bool okl = lr.tail && tl.ok0 && nz4(r);
      if (LL) lds_add4(ll, lo_ + (tl.i0 << lv) * lst + qo, r, okl); else atomic_add4(GL, l0, r, okl);
    }
    r = run_scan4(m1 ? dl * tl.w1 : zero, lr.start, s);
    {
      const bool okl = lr.tail && tl.ok1 && nz4(r);
      if (LL) lds_add4(ll, lo_ + ((tl.i0 + 1) << lv) * lst + qo, r, okl); else atomic_add4(GL, l1, r, okl);
    }
  }
  // plane 0 = (x, y | z), 1 = (x, z | y), 2 = (y, z | x).  Selects, not branches: pi differs between the lane
  // halves of the appearance scatter, and the branchy form made the compiler keep dx0..2 in a scratch array
  // indexed per lane (scratch load + vmcnt(0) + store per update, draining the atomics in flight).
  dx0 += pi == 2 ? gcl : gcx;
  dx1 += pi == 0 ? gcy : (pi == 1 ? gcl : gcx);
  dx2 += pi == 0 ? gcl : gcy;
}

// XY quads, four at a time: the wave works on 16 samples (sub-tile j of the 32-sample tile) and the
// four quads 4*grp .. 4*grp+3 of the XY plane at one level: lane = (q = lane>>4, s16 = lane&15).
// Run structure and tail positions are identical in the four 16-lane rows (same samples), so in each
// atomic instruction the four rows carry the four quads of the SAME texel: 64 contiguous bytes
// (16 components), which the L2 coalescer turns into one request -- the (quad | quad) half-wave
// pairing of gather_quad_bwd needed two.  A 16-lane run-scan is four row_shr steps.
// xs/live: coordinates and liveness of THIS lane's sample (sub-tile j); q_is_owner: this lane also
// owns that sample in the (half, sample) mapping of the caller's dx accumulators.
RDRF_D Run run_of16(int key, int s16) {
  const int prev = dppi<0x111>(key);  // row_shr:1 (lane 0 of a row reads 0)
  const bool head = (s16 == 0) || (prev != key);
  const unsigned long long b = __ballot(head);
  const unsigned m = (unsigned)(b >> (16 * ((threadIdx.x & 63) >> 4))) & 0xffffu;
  Run r;
  r.start = 31 - __clz((int)(m & (0xffffu >> (15 - s16))));
  r.tail = (s16 == 15) || ((m >> (s16 + 1)) & 1u);
  return r;
}
RDRF_D f32x4 run_scan4_16(f32x4 v, int start, int s16) {
#define RDRF_SCAN_STEP(D)                                                                   \
  {                                                                                         \
    const float ox = dppf<0x110 + D>(v.x), oy = dppf<0x110 + D>(v.y);                       \
    const float oz = dppf<0x110 + D>(v.z), ow = dppf<0x110 + D>(v.w);                       \
    const bool take = s16 >= D && s16 - D >= start;                                         \
    const float tx_ = v.x + ox, ty_ = v.y + oy, tz_ = v.z + oz, tw_ = v.w + ow;             \
    v.x = take ? tx_ : v.x; v.y = take ? ty_ : v.y; v.z = take ? tz_ : v.z; v.w = take ? tw_ : v.w; \
  }
  RDRF_SCAN_STEP(1)
  RDRF_SCAN_STEP(2)
  RDRF_SCAN_STEP(4)
  RDRF_SCAN_STEP(8)
#undef RDRF_SCAN_STEP
  return v;
}
RDRF_D f32x4 shfl4_row(f32x4 v, int src_lane) {
  f32x4 r;
  r.x = __shfl(v.x, src_lane, 64); r.y = __shfl(v.y, src_lane, 64);
  r.z = __shfl(v.z, src_lane, 64); r.w = __shfl(v.w, src_lane, 64);
  return r;
}
template <int C0Q, int C1Q, bool TILED = false>
RDRF_D void gather_xy4_bwd(const RdrfVM& vm, const RdrfVM& gvm, int lv, int q4, float x0, float x1, float x2,
                           f32x4 dq, bool live, bool q_is_owner, float& dx0, float& dx1, float& dx2,
                           const LdsLines ll, const PlaneTile T = PlaneTile{nullptr, nullptr, 0}) {
  const int lane = threadIdx.x & 63, s16 = lane & 15, rowbase = lane & ~15;
  const float* P = vm.plane[0];
  const float* Lp = vm.line[0];
  float* GP = gvm.plane[0];
  float* GL = gvm.line[0];
  const int H = vm.H[0], W = vm.W[0], L = vm.L[0], sH = vm.sH[0], sW = vm.sW[0];
  const int st = 1 << lv;
  const int Ws = (W + st - 1) >> lv, Hs = (H + st - 1) >> lv, Ls = (L + st - 1) >> lv;
  Tap1 tx = tap1d(x0, Ws), ty = tap1d(x1, Hs), tl = tap1d(x2, Ls);
  constexpr int C = 4 * C0Q;
  const int qo = 4 * q4;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const size_t o00 = (size_t)((ty.i0 << lv) * sH + (tx.i0 << lv) * sW) + qo;
  const size_t o01 = (size_t)((ty.i0 << lv) * sH + ((tx.i0 + 1) << lv) * sW) + qo;
  const size_t o10 = (size_t)(((ty.i0 + 1) << lv) * sH + (tx.i0 << lv) * sW) + qo;
  const size_t o11 = (size_t)(((ty.i0 + 1) << lv) * sH + ((tx.i0 + 1) << lv) * sW) + qo;
  const bool g00 = ty.ok0 && tx.ok0, g01 = ty.ok0 && tx.ok1, g10 = ty.ok1 && tx.ok0, g11 = ty.ok1 && tx.ok1;
  const bool k00 = live && g00, k01 = live && g01, k10 = live && g10, k11 = live && g11;
  const bool m0 = live && tl.ok0, m1 = live && tl.ok1;
  const int x0c = min(max(tx.i0, 0), Ws - 1) << lv, x1c = min(max(tx.i0 + 1, 0), Ws - 1) << lv;
  const int y0c = min(max(ty.i0, 0), Hs - 1) << lv, y1c = min(max(ty.i0 + 1, 0), Hs - 1) << lv;
  const int l0c = min(max(tl.i0, 0), Ls - 1) << lv, l1c = min(max(tl.i0 + 1, 0), Ls - 1) << lv;
  f32x4 v00 = ld4(P + (size_t)(y0c * sH + x0c * sW) + qo), v01 = ld4(P + (size_t)(y0c * sH + x1c * sW) + qo);
  f32x4 v10 = ld4(P + (size_t)(y1c * sH + x0c * sW) + qo), v11 = ld4(P + (size_t)(y1c * sH + x1c * sW) + qo);
  f32x4 a0 = ld4(Lp + (size_t)l0c * C + qo), a1 = ld4(Lp + (size_t)l1c * C + qo);
  if (!k00) v00 = zero;
  if (!k01) v01 = zero;
  if (!k10) v10 = zero;
  if (!k11) v11 = zero;
  if (!m0) a0 = zero;
  if (!m1) a1 = zero;
  const f32x4 pv = v00 * (tx.w0 * ty.w0) + v01 * (tx.w1 * ty.w0) + v10 * (tx.w0 * ty.w1) +
                   v11 * (tx.w1 * ty.w1);
  const f32x4 lvv = a0 * tl.w0 + a1 * tl.w1;
  const f32x4 dp = live ? dq * lvv : zero;
  const f32x4 dl = live ? dq * pv : zero;
  // coordinate gradients FIRST: they are the last consumers of the gathered taps.  Computed after the atomics
  // (as the formulas read), the wait for the taps sat behind 16 conditional atomics -- vmcnt is one in-order
  // counter on gfx9, and with conditional issues the compiler must assume the smallest count -- so every
  // iteration waited for all of its own atomics to be acknowledged by the memory side.
  float gcx = 0.5f * (float)(Ws - 1) * dot4(dp, (v01 - v00) * ty.w0 + (v11 - v10) * ty.w1);
  float gcy = 0.5f * (float)(Hs - 1) * dot4(dp, (v10 - v00) * tx.w0 + (v11 - v01) * tx.w1);
  float gcl = 0.5f * (float)(Ls - 1) * dot4(dl, a1 - a0);
  // sum over the four quads (rows) of this sample, then hand it to the lane that owns the sample
  gcx += __shfl_xor(gcx, 16, 64); gcy += __shfl_xor(gcy, 16, 64); gcl += __shfl_xor(gcl, 16, 64);
  gcx += __shfl_xor(gcx, 32, 64); gcy += __shfl_xor(gcy, 32, 64); gcl += __shfl_xor(gcl, 32, 64);
  const int pkey = ((ty.i0 + 4) << 16) | ((tx.i0 + 4) & 0xffff);
  const Run pr = run_of16(pkey, s16);
  f32x4 r00 = run_scan4_16(k00 ? dp * (tx.w0 * ty.w0) : zero, pr.start, s16);
  f32x4 r01 = run_scan4_16(k01 ? dp * (tx.w1 * ty.w0) : zero, pr.start, s16);
  f32x4 r10 = run_scan4_16(k10 ? dp * (tx.w0 * ty.w1) : zero, pr.start, s16);
  f32x4 r11 = run_scan4_16(k11 ? dp * (tx.w1 * ty.w1) : zero, pr.start, s16);
  {  // cross-run merge of shared taps, all four directions (see gather_quad_bwd)
    auto recv_mask = [](int d) { return d == 65536 ? 3 : (d == -65536 ? 12 : (d == 1 ? 5 : (d == -1 ? 10 : 0))); };
    auto skip_mask = [](int d) { return d == 65536 ? 12 : (d == -65536 ? 3 : (d == 1 ? 10 : (d == -1 ? 5 : 0))); };
    const int pl = rowbase | (pr.start > 0 ? pr.start - 1 : 0);
    const int pk = __shfl(pkey, pl, 64);
    const int din = pr.start > 0 ? pkey - pk : 0;
    const int pdin = __shfl(din, pl, 64);
    const int min_ = skip_mask(din) & ~recv_mask(pdin);
    const int nk = dppi<0x101>(pkey);                        // row_shl:1 -> key of lane s16+1
    const int dout = s16 < 15 ? nk - pkey : 0;
    const int skip_out = skip_mask(dout) & ~recv_mask(din);
    // the EARLIER run prepares the two taps it hands over (its direction out = the later run's
    // direction in), so the later run pulls two quads instead of four
    (void)min_;
    const f32x4 hA = dout == 65536 ? ((skip_out & 4) ? r10 : zero) : (dout == -65536 ? ((skip_out & 1) ? r00 : zero)
                   : (dout == 1 ? ((skip_out & 2) ? r01 : zero) : ((skip_out & 1) ? r00 : zero)));
    const f32x4 hB = dout == 65536 ? ((skip_out & 8) ? r11 : zero) : (dout == -65536 ? ((skip_out & 2) ? r01 : zero)
                   : (dout == 1 ? ((skip_out & 8) ? r11 : zero) : ((skip_out & 4) ? r10 : zero)));
    const f32x4 pA = shfl4_row(hA, pl), pB = shfl4_row(hB, pl);
    if (din == 65536) { r00 = r00 + pA; r01 = r01 + pB; }
    else if (din == -65536) { r10 = r10 + pA; r11 = r11 + pB; }
    else if (din == 1) { r00 = r00 + pA; r10 = r10 + pB; }
    else if (din == -1) { r01 = r01 + pA; r11 = r11 + pB; }
    plane_add4<TILED>(T, lv, tx.i0, ty.i0, qo, GP, o00, r00, pr.tail && g00 && nz4(r00) && !(skip_out & 1));
    plane_add4<TILED>(T, lv, tx.i0 + 1, ty.i0, qo, GP, o01, r01, pr.tail && g01 && nz4(r01) && !(skip_out & 2));
    plane_add4<TILED>(T, lv, tx.i0, ty.i0 + 1, qo, GP, o10, r10, pr.tail && g10 && nz4(r10) && !(skip_out & 4));
    plane_add4<TILED>(T, lv, tx.i0 + 1, ty.i0 + 1, qo, GP, o11, r11, pr.tail && g11 && nz4(r11) && !(skip_out & 8));
  }
  {
    const Run lr = run_of16(tl.i0 + 4, s16);
    const bool LL = ll.base != nullptr;
    const int lst = lds_stride(C);
    if (LL && ll.direct) {
      const f32x4 r0 = m0 ? dl * tl.w0 : zero, r1 = m1 ? dl * tl.w1 : zero;
      lds_add4(ll, ll.off[0] + (tl.i0 << lv) * lst + qo, r0, m0 && nz4(r0));
      lds_add4(ll, ll.off[0] + ((tl.i0 + 1) << lv) * lst + qo, r1, m1 && nz4(r1));
    } else {
    f32x4 r = run_scan4_16(m0 ? dl * tl.w0 : zero, lr.start, s16);
    bool okl = lr.tail && tl.ok0 && nz4(r);
    if (LL) lds_add4(ll, ll.off[0] + (tl.i0 << lv) * lst + qo, r, okl); else atomic_add4(GL, (size_t)(tl.i0 << lv) * C + qo, r, okl);
    r = run_scan4_16(m1 ? dl * tl.w1 : zero, lr.start, s16);
    okl = lr.tail && tl.ok1 && nz4(r);
    if (LL) lds_add4(ll, ll.off[0] + ((tl.i0 + 1) << lv) * lst + qo, r, okl); else atomic_add4(GL, (size_t)((tl.i0 + 1) << lv) * C + qo, r, okl);
    }
  }
  if (q_is_owner) { dx0 += gcx; dx1 += gcy; dx2 += gcl; }
}

// XZ / YZ quads of the ray-tile scatter, column-split: BOTH half-waves work on the same quad g of the
// same 32 samples; half h owns the bilinear column ix + h (its lower and upper row taps) and the line
// tap h.  The run structure is identical in the two halves, so in every atomic instruction the lanes
// of half 0 carry texel (iy, ix) and the same lanes of half 1 carry texel (iy, ix + 1): with x-fastest
// plane storage these are 16 bytes apart and the L2 coalescer (which merges same-line lanes across
// the whole wave, tools/ubench/atomics.hip kernels G/H) makes ONE request of them -- half the atomic
// requests of the XZ / YZ planes, which were ~1/3 of the density scatter's time.
// Coordinate gradients are computed by both halves and halved (x*0.5 + x*0.5 is exact).
template <int C0Q, int C1Q, bool TILED = false>
RDRF_D void gather_zquad_bwd(const RdrfVM& vm, const RdrfVM& gvm, int g, int h, float x0, float x1, float x2,
                             f32x4 dq, bool live, int s, float& dx0, float& dx1, float& dx2,
                             const LdsLines ll, const PlaneTile T = PlaneTile{nullptr, nullptr, 0}) {
  QuadSel<C0Q, C1Q> sl = quad_sel<C0Q, C1Q>(g);
  const int pi = sl.pi;   // 1 or 2 (wave-uniform)
  const float cx = pi == 2 ? x1 : x0;
  const float cy = x2;
  const float cl = pi == 1 ? x1 : x0;
  const float* P = pi == 1 ? vm.plane[1] : vm.plane[2];
  const float* Lp = pi == 1 ? vm.line[1] : vm.line[2];
  float* GP = pi == 1 ? gvm.plane[1] : gvm.plane[2];
  float* GL = pi == 1 ? gvm.line[1] : gvm.line[2];
  const int H = pi == 1 ? vm.H[1] : vm.H[2], W = pi == 1 ? vm.W[1] : vm.W[2], L = pi == 1 ? vm.L[1] : vm.L[2];
  const int sH = pi == 1 ? vm.sH[1] : vm.sH[2], sW = pi == 1 ? vm.sW[1] : vm.sW[2];
  const int lv = sl.level, st = 1 << lv;
  const int Ws = (W + st - 1) >> lv, Hs = (H + st - 1) >> lv, Ls = (L + st - 1) >> lv;
  Tap1 tx = tap1d(cx, Ws), ty = tap1d(cy, Hs), tl = tap1d(cl, Ls);
  const int C = sl.C, qo = 4 * sl.q;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const int x0c = min(max(tx.i0, 0), Ws - 1) << lv, x1c = min(max(tx.i0 + 1, 0), Ws - 1) << lv;
  const int y0c = min(max(ty.i0, 0), Hs - 1) << lv, y1c = min(max(ty.i0 + 1, 0), Hs - 1) << lv;
  const int l0c = min(max(tl.i0, 0), Ls - 1) << lv, l1c = min(max(tl.i0 + 1, 0), Ls - 1) << lv;
  f32x4 v00 = ld4(P + (size_t)(y0c * sH + x0c * sW) + qo), v01 = ld4(P + (size_t)(y0c * sH + x1c * sW) + qo);
  f32x4 v10 = ld4(P + (size_t)(y1c * sH + x0c * sW) + qo), v11 = ld4(P + (size_t)(y1c * sH + x1c * sW) + qo);
  f32x4 a0 = ld4(Lp + (size_t)l0c * C + qo), a1 = ld4(Lp + (size_t)l1c * C + qo);
  if (!(live && ty.ok0 && tx.ok0)) v00 = zero;
  if (!(live && ty.ok0 && tx.ok1)) v01 = zero;
  if (!(live && ty.ok1 && tx.ok0)) v10 = zero;
  if (!(live && ty.ok1 && tx.ok1)) v11 = zero;
  if (!(live && tl.ok0)) a0 = zero;
  if (!(live && tl.ok1)) a1 = zero;
  const f32x4 pv = v00 * (tx.w0 * ty.w0) + v01 * (tx.w1 * ty.w0) + v10 * (tx.w0 * ty.w1) +
                   v11 * (tx.w1 * ty.w1);
  const f32x4 lvv = a0 * tl.w0 + a1 * tl.w1;
  const f32x4 dp = live ? dq * lvv : zero;
  const f32x4 dl = live ? dq * pv : zero;
  // (coordinate gradients before the atomics: see gather_xy4_bwd)
  const float gcx = 0.25f * (float)(Ws - 1) * dot4(dp, (v01 - v00) * ty.w0 + (v11 - v10) * ty.w1);
  const float gcy = 0.25f * (float)(Hs - 1) * dot4(dp, (v10 - v00) * tx.w0 + (v11 - v01) * tx.w1);
  const float gcl = 0.25f * (float)(Ls - 1) * dot4(dl, a1 - a0);
  // this half's column
  const float wxc = h ? tx.w1 : tx.w0;
  const bool okc = h ? tx.ok1 : tx.ok0;
  const int ixc = tx.i0 + h;
  const bool g0 = ty.ok0 && okc, g1 = ty.ok1 && okc;
  const size_t o0 = (size_t)((ty.i0 << lv) * sH + (ixc << lv) * sW) + qo;
  const size_t o1 = (size_t)(((ty.i0 + 1) << lv) * sH + (ixc << lv) * sW) + qo;
  const int pkey = ((ty.i0 + 4) << 16) | ((tx.i0 + 4) & 0xffff);   // same key in both halves
  const Run pr = run_of(pkey, s);
  f32x4 r0 = run_scan4((live && g0) ? dp * (wxc * ty.w0) : zero, pr.start, s);
  const f32x4 r1 = run_scan4((live && g1) ? dp * (wxc * ty.w1) : zero, pr.start, s);
  // cross-run merge along the row axis (see gather_quad_bwd)
  const int pl = pr.start > 0 ? pr.start - 1 : 0;
  const int pk = __shfl(pkey, pl, 32);
  const bool chain_prev = pr.start > 0 && pk == pkey - (1 << 16);
  const float ux = __shfl(r1.x, pl, 32), uy = __shfl(r1.y, pl, 32), uz = __shfl(r1.z, pl, 32),
              uw = __shfl(r1.w, pl, 32);
  if (chain_prev) { r0.x += ux; r0.y += uy; r0.z += uz; r0.w += uw; }
  const int nk = dppi<0x130>(pkey);
  const bool up_ok = !(s < 31 && nk == pkey + (1 << 16));
  plane_add4<TILED>(T, lv, ixc, ty.i0, qo, GP, o0, r0, pr.tail && g0 && nz4(r0));
  plane_add4<TILED>(T, lv, ixc, ty.i0 + 1, qo, GP, o1, r1, pr.tail && up_ok && g1 && nz4(r1));
  // line tap h
  if (ll.base && ll.direct) {
    const bool okl = live && (h ? tl.ok1 : tl.ok0);
    const f32x4 r = okl ? dl * (h ? tl.w1 : tl.w0) : zero;
    lds_add4(ll, (pi == 1 ? ll.off[1] : ll.off[2]) + ((tl.i0 + h) << lv) * lds_stride(C) + qo, r, okl && nz4(r));
  } else {
    const Run lr = run_of(tl.i0 + 4, s);
    const bool okl = h ? tl.ok1 : tl.ok0;
    const f32x4 r = run_scan4((live && okl) ? dl * (h ? tl.w1 : tl.w0) : zero, lr.start, s);
    const bool doit = lr.tail && okl && nz4(r);
    const int li = tl.i0 + h;
    if (ll.base) lds_add4(ll, (pi == 1 ? ll.off[1] : ll.off[2]) + (li << lv) * lds_stride(C) + qo, r, doit);
    else atomic_add4(GL, (size_t)(li << lv) * C + qo, r, doit);
  }
  dx0 += pi == 1 ? gcx : gcl;   // plane 1 = (x, z | y), 2 = (y, z | x)
  dx1 += pi == 1 ? gcl : gcx;
  dx2 += gcy;
}

// d(X0)/d(xn): X0 = [xn, t | (sin q, cos q) pairs], q_j = xn[j/10] * 2^(j%10); returns this lane
// half's partial (combine with __shfl_xor 32)
RDRF_D void x0_bwd(const float (&X0)[32], const float (&dX0)[32], int h, float& d0, float& d1,
                   float& d2) {
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    if (o == 0 && h == 0) {
      d0 += dX0[0]; d1 += dX0[1]; d2 += dX0[2];
    } else {
      const int k = 2 * o + h - 1;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int j = 2 * k + p;
        const int d = j / 10, f = j - d * 10;
        const float sv = X0[o * 4 + 2 * p], cv = X0[o * 4 + 2 * p + 1];
        const float dq = ldexpf(dX0[o * 4 + 2 * p] * cv - dX0[o * 4 + 2 * p + 1] * sv, f);
        // (selects: d differs between the lane halves, and a branchy update makes d0..2 a scratch array indexed per lane)
        d0 += d == 0 ? dq : 0.f; d1 += d == 1 ? dq : 0.f; d2 += d == 2 ? dq : 0.f;
      }
    }
  }
}

RDRF_D float act_grad(float f, int act, float shift) {
  return act == RDRF_ACT_RELU ? (f > 0.0f ? 1.0f : 0.0f) : sigmoidf_(f + shift);
}

// backward of a small output layer kept on the VALU (NO <= 6 outputs): dz[kk] = relu'(H[kk]) * sum_o W[o][kk] dzo[o]
// for this lane half's KK inputs.  ws = [NO][2][KK] in LDS, read as 16-byte quads (element-wise `lds[...]` reads
// compiled to one ds_read_b32 + lgkmcnt(0) wait per weight).
template <int KK, int NO>
RDRF_D void small_layer_bwd(float (&dz)[KK], const float (&H)[KK], const float* __restrict__ ws, int h,
                            const float (&dzo)[NO]) {
#pragma unroll
  for (int q = 0; q < KK / 4; ++q) {
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(ws + o * 2 * KK + h * KK + 4 * q);
      d.x = fmaf(wv.x, dzo[o], d.x); d.y = fmaf(wv.y, dzo[o], d.y);
      d.z = fmaf(wv.z, dzo[o], d.z); d.w = fmaf(wv.w, dzo[o], d.w);
    }
    dz[4 * q + 0] = H[4 * q + 0] > 0.f ? d.x : 0.f; dz[4 * q + 1] = H[4 * q + 1] > 0.f ? d.y : 0.f;
    dz[4 * q + 2] = H[4 * q + 2] > 0.f ? d.z : 0.f; dz[4 * q + 3] = H[4 * q + 3] > 0.f ? d.w : 0.f;
  }
}

template <int NB>
RDRF_D void acc_zero(f32x16 (&acc)[NB]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
}

// ------------------------------------------------------------------------------------------------
// appearance phase backward-data (dynamic: MLP_Fea_late_view; static: MLP_Fea | TimeEmbedding)
// ------------------------------------------------------------------------------------------------
// FEAT: the features ARE the basis output: d(features) arrive in g_feat, only the basis backward runs
template <bool FEAT>
RDRF_D void feat_dF(float (&dF)[16], const float* g_feat, int idx, bool act, int h) {
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const int e = elem_of(kk, h);
    dF[kk] = (act && e < 27) ? g_feat[(size_t)idx * 27 + e] : 0.f;
  }
}

// REC: d(app features) go out as sample-major records for the sorted scatter (a.dfa) instead of DA rows

// one backward-data layer of the appearance phases: NBI input blocks from one dz vector, fp32 pipe (A/B builds) or split-storage
// bf16 x 3 (the lo pieces of step 0 are requested here: callers place the call so that row loads / VALU work follow the request)
template <int NBI, int KK>
RDRF_D void app_bwd_seg(f32x16 (&acc)[NBI], const float (&dz)[KK], const float* __restrict__ whm, const float* __restrict__ pk,
                        int lo_reg, int lo_off, int lane) {
#ifdef RDRF_APP_F32
  mfma_seg<NBI, KK>(acc, dz, whm, lane);
#else
  const B3sLo st = b3s_lo_stream(pk + lo_reg, lane);
  u32x4 lo[NBI];
  b3s_lo_load<NBI>(lo, st, lo_off, KK / 8, 0);
  mfma_seg_b3s<NBI, KK, 0>(acc, dz, whm, st, lo_off, 0, lo, lane);
#endif
}
template <bool FEAT, bool REC = false>
__global__ __launch_bounds__(64 * RDRF_MAXW) void k_dyn_app_bwd(BwdArgs a, DynW w, DynG gw) {
  __shared__ int s_next;
  if (threadIdx.x == 0) s_next = blockDim.x >> 6;   // (before the barrier of lds_fill)
  __shared__ __attribute__((aligned(16))) float lds[pkb::K3_SIZE];
  lds_fill(lds, a.pk + pkb::REG_K3, pkb::K3_SIZE);
  const float* basisT = lds + pkb::K3_BASIST;
  const int lane = threadIdx.x & 63, h = lane >> 5, s = lane & 31;
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int count = FEAT ? a.M : a.sp.hdr->count;
  const int ntiles = (count + 31) >> 5;
  for (int k = wave, tile; (tile = blockIdx.x + k * gridDim.x) < ntiles; k = tile_queue_next(&s_next, k, nwaves, a.dynq != 0)) {
    const int li = tile * 32 + s;
    const bool act = li < count;
    const int idx = act ? (FEAT ? li : a.sp.list[li]) : 0;
    const int n = idx / a.S;
    const float* svb = a.sp.act3 + (size_t)tile * sv::K3_ROWS * 32;
    float* gb = a.grows3 + (size_t)tile * sv::K3G_ROWS * 32;
    if constexpr (FEAT) {
      float dF[16];
      feat_dF<true>(dF, a.g_feat, idx, act, h);
      save_rows<16>(gb, sv::K3G_DF, dF, s, h);
      f32x16 acc[7];
      acc_zero<7>(acc);
      app_bwd_seg<7, 16>(acc, dF, basisT, a.pk, pkb::REG_K3_LO, pkb::K3_LO_BASIST, lane);
      float dA[112];
      acc_copy<7>(dA, acc);
      save_rows<112>(gb, sv::K3G_DA, dA, s, h);
      continue;
    }
    float vx, vy, vz;
    ray_norm(a.rays, n, a.ray_type, vx, vy, vz);
    // ---- output layer: v = Wv [H2, vd] + b ; rgb = sigmoid(v)
    float H2[64];
    load_rows<64>(svb, sv::K3_H2, H2, s, h);
    float dzv[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float v = dot_small<64>(H2, lds + pkb::K3_RGBV + o * 128, h) + w.rbv[o];
      v += w.rwv[o * 131 + 128] * vx + w.rwv[o * 131 + 129] * vy + w.rwv[o * 131 + 130] * vz;
      const float r = sigmoidf_(v);
      const float g = (act && a.g_rgb) ? a.g_rgb[(size_t)idx * 3 + o] : 0.f;
      dzv[o] = g * r * (1.0f - r);
      if (h == 0) gb[(size_t)(sv::K3G_DZV + o) * 32 + s] = dzv[o];
    }
    float dz2[64];
    small_layer_bwd<64, 3>(dz2, H2, lds + pkb::K3_RGBV, h, dzv);
    save_rows<64>(gb, sv::K3G_DZ2, dz2, s, h);
    // ---- layer 2 backward: dH1 = W2^T dz2
    float dz1[64];
    {
      f32x16 acc[4];
      acc_zero<4>(acc);
      app_bwd_seg<4, 64>(acc, dz2, lds + pkb::K3_RGB2T, a.pk, pkb::REG_K3_LO, pkb::K3_LO_RGB2T, lane);
      float H1[64];
      load_rows<64>(svb, sv::K3_H1, H1, s, h);
#pragma unroll
      for (int kk = 0; kk < 64; ++kk) dz1[kk] = H1[kk] > 0.f ? acc[kk >> 4][kk & 15] : 0.f;
    }
    save_rows<64>(gb, sv::K3G_DZ1, dz1, s, h);
    // ---- layer 1 backward: feature block and X0 block (t / PE(t) carry no gradient)
    float dF[16], dX0[32];
    {
#ifdef RDRF_APP_F32
      f32x16 acc[1];
      acc_zero<1>(acc);
      mfma_seg<1, 64>(acc, dz1, lds + pkb::K3_RGB1T_F, lane);
      acc_copy<1>(dF, acc);
      f32x16 acc2[2];
      acc_zero<2>(acc2);
      mfma_seg<2, 64>(acc2, dz1, lds + pkb::K3_RGB1T_X0, lane);
      acc_copy<2>(dX0, acc2);
#else   // the F block and the two X0 blocks are one three-block image: one split of dz1
      f32x16 acc[3];
      acc_zero<3>(acc);
      app_bwd_seg<3, 64>(acc, dz1, lds + pkb::K3_RGB1T_F, a.pk, pkb::REG_K3_LO, pkb::K3_LO_RGB1T, lane);
#pragma unroll
      for (int i = 0; i < 16; ++i) dF[i] = acc[0][i];
#pragma unroll
      for (int i = 0; i < 32; ++i) dX0[i] = acc[1 + (i >> 4)][i & 15];
#endif
    }
    save_rows<16>(gb, sv::K3G_DF, dF, s, h);
    float dn0 = 0.f, dn1 = 0.f, dn2 = 0.f;
    {
      float X0[32];
      load_rows<32>(svb, sv::K3_X0, X0, s, h);
      x0_bwd(X0, dX0, h, dn0, dn1, dn2);
    }
    dn0 += __shfl_xor(dn0, 32, 64); dn1 += __shfl_xor(dn1, 32, 64); dn2 += __shfl_xor(dn2, 32, 64);
    // ---- basis backward: d(app features) rows for the scatter kernel
    {
      f32x16 acc[7];
      acc_zero<7>(acc);
      app_bwd_seg<7, 16>(acc, dF, basisT, a.pk, pkb::REG_K3_LO, pkb::K3_LO_BASIST, lane);
      float dA[112];
      acc_copy<7>(dA, acc);
      if constexpr (REC) {
        // sample-major record for the sorted scatter: this lane half holds the feature quads Q = 2 m + h (slots
        // 4m..4m+3) of compacted sample li; one 16-byte store per quad, the two halves write adjacent quads
        if (act) {
          // quad 2 m + 1 sits 4 floats after quad 2 m in the record, except for the (XZ quad 2 | YZ quad 0) pair of
          // every level: one base pointer per half + immediate offsets
          float* rec = a.dfa + (size_t)li * DFA_FLOATS + 4 * h;
          float* recx = a.dfa + (size_t)li * DFA_FLOATS + (h ? dfa_off(15) - dfa_off(14) : 0);
#pragma unroll
          for (int m = 0; m < 27; ++m) {
            static_assert(dfa_off(1) == dfa_off(0) + 4 && dfa_off(13) == dfa_off(12) + 4 && dfa_off(17) == dfa_off(16) + 4, "record layout");
            float* dst = (m % 9 == 7 ? recx : rec) + dfa_off(2 * m);
            *reinterpret_cast<f32x4*>(dst) = f32x4{dA[4 * m], dA[4 * m + 1], dA[4 * m + 2], dA[4 * m + 3]};
          }
        }
      } else {
        save_rows<112>(gb, sv::K3G_DA, dA, s, h);
      }
    }
    if (act && h == 0) {
      a.dxn_app[(size_t)idx * 3 + 0] = dn0; a.dxn_app[(size_t)idx * 3 + 1] = dn1;
      a.dxn_app[(size_t)idx * 3 + 2] = dn2;
    }
  }
}

template <int HEAD, bool FEAT>
__global__ __launch_bounds__(64 * RDRF_MAXW) void k_static_app_bwd(BwdArgs a, StaticW w, StaticG gw) {
  __shared__ int s_next;
  if (threadIdx.x == 0) s_next = blockDim.x >> 6;   // (before the barrier of lds_fill)
  __shared__ __attribute__((aligned(16))) float lds[pkb::S3_SIZE];
  lds_fill(lds, a.pk + pkb::REG_S3, pkb::S3_SIZE);
  const int lane = threadIdx.x & 63, h = lane >> 5, s = lane & 31;
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int count = FEAT ? a.M : a.sp.hdr->count;
  const int ntiles = (count + 31) >> 5;
  for (int k = wave, tile; (tile = blockIdx.x + k * gridDim.x) < ntiles; k = tile_queue_next(&s_next, k, nwaves, a.dynq != 0)) {
    const int li = tile * 32 + s;
    const bool act = li < count;
    const int idx = act ? (FEAT ? li : a.sp.list[li]) : 0;
    const int n = idx / a.S;
    const float* svb = a.sp.act3 + (size_t)tile * sv::S3_ROWS * 32;
    float* gb = a.grows3 + (size_t)tile * sv::K3G_ROWS * 32;
    if constexpr (FEAT) {
      float dF[16];
      feat_dF<true>(dF, a.g_feat, idx, act, h);
      save_rows<16>(gb, sv::K3G_DF, dF, s, h);
      f32x16 acc[3];
      acc_zero<3>(acc);
      app_bwd_seg<3, 16>(acc, dF, lds + pkb::S3_BASIST, a.pk, pkb::REG_S3_LO, pkb::S3_LO_BASIST, lane);
      float dG[48];
      acc_copy<3>(dG, acc);
      save_rows<48>(gb, sv::K3G_DA, dG, s, h);
      continue;
    }
    float vx, vy, vz;
    ray_norm(a.rays, n, a.ray_type, vx, vy, vz);
    float H2[64];
    load_rows<64>(svb, sv::S3_H2, H2, s, h);
    float dzv[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float v = dot_small<64>(H2, lds + pkb::S3_W3 + o * 128, h) + w.b3[o];
      if (HEAD == RDRF_HEAD_MLP_FEA_TIMEEMBEDDING)
        v += w.w3[o * 131 + 128] * vx + w.w3[o * 131 + 129] * vy + w.w3[o * 131 + 130] * vz;
      const float r = sigmoidf_(v);
      const float g = (act && a.g_rgb) ? a.g_rgb[(size_t)idx * 3 + o] : 0.f;
      dzv[o] = g * r * (1.0f - r);
      if (h == 0) gb[(size_t)(sv::K3G_DZV + o) * 32 + s] = dzv[o];
    }
    float dz2[64];
    small_layer_bwd<64, 3>(dz2, H2, lds + pkb::S3_W3, h, dzv);
    save_rows<64>(gb, sv::K3G_DZ2, dz2, s, h);
    float dz1[64];
    {
      f32x16 acc[4];
      acc_zero<4>(acc);
      app_bwd_seg<4, 64>(acc, dz2, lds + pkb::S3_W2T, a.pk, pkb::REG_S3_LO, pkb::S3_LO_W2T, lane);
      float H1[64];
      load_rows<64>(svb, sv::S3_H1, H1, s, h);
#pragma unroll
      for (int kk = 0; kk < 64; ++kk) dz1[kk] = H1[kk] > 0.f ? acc[kk >> 4][kk & 15] : 0.f;
    }
    save_rows<64>(gb, sv::K3G_DZ1, dz1, s, h);
    float dF[16];
    {
#ifdef RDRF_APP_F32
      f32x16 acc[1];
      acc_zero<1>(acc);
      mfma_seg<1, 64>(acc, dz1, lds + pkb::S3_W1T_F, lane);
      acc_copy<1>(dF, acc);
      f32x16 accp[4];
      acc_zero<4>(accp);
      mfma_seg<4, 64>(accp, dz1, lds + pkb::S3_W1T_P, lane);
#else   // the F block and the four PE blocks are one five-block image: one split of dz1
      f32x16 acc5[5];
      acc_zero<5>(acc5);
      app_bwd_seg<5, 64>(acc5, dz1, lds + pkb::S3_W1T_F, a.pk, pkb::REG_S3_LO, pkb::S3_LO_W1T, lane);
#pragma unroll
      for (int i = 0; i < 16; ++i) dF[i] = acc5[0][i];
      f32x16 (&accp)[4] = *reinterpret_cast<f32x16 (*)[4]>(&acc5[1]);
#endif
      float P[64];
      load_rows<64>(svb, sv::S3_P, P, s, h);
      // PE2 backward: P[4r..4r+3] = (sin f, cos f, sin 2f, cos 2f) of feature slot r
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d0 = accp[r >> 2][(r & 3) * 4 + 0], d1 = accp[r >> 2][(r & 3) * 4 + 1];
        const float d2 = accp[r >> 2][(r & 3) * 4 + 2], d3 = accp[r >> 2][(r & 3) * 4 + 3];
        dF[r] += d0 * P[4 * r + 1] - d1 * P[4 * r + 0] + 2.0f * (d2 * P[4 * r + 3] - d3 * P[4 * r + 2]);
      }
    }
    // the view-direction slots (27..29) of the feature block are not features: they carry
    // d(loss)/d(viewdir); viewdir = d/|d|  =>  g_d = (g_v - (g_v.v) v) / |d|
    {
      float gv0 = 0.f, gv1 = 0.f, gv2 = 0.f;
      if (HEAD == RDRF_HEAD_MLP_FEA) {
        if (h == 0) { gv0 = dF[15]; dF[15] = 0.f; }
        else { gv1 = dF[12]; gv2 = dF[13]; dF[12] = 0.f; dF[13] = 0.f; }
        gv0 += __shfl_xor(gv0, 32, 64); gv1 += __shfl_xor(gv1, 32, 64); gv2 += __shfl_xor(gv2, 32, 64);
      } else {
#pragma unroll
        for (int o = 0; o < 3; ++o) {
          gv0 += w.w3[o * 131 + 128] * dzv[o]; gv1 += w.w3[o * 131 + 129] * dzv[o];
          gv2 += w.w3[o * 131 + 130] * dzv[o];
        }
      }
      if (a.g_rays && act && h == 0 && a.ray_type != RDRF_RAY_OTHER) {
        const float rx = a.rays[(size_t)n * 6 + 3], ry = a.rays[(size_t)n * 6 + 4], rz = a.rays[(size_t)n * 6 + 5];
        const float nr = sqrtf(rx * rx + ry * ry + rz * rz);
        const float dotv = gv0 * vx + gv1 * vy + gv2 * vz;
        atomicAdd(a.g_rays + (size_t)n * 6 + 3, (gv0 - dotv * vx) / nr);
        atomicAdd(a.g_rays + (size_t)n * 6 + 4, (gv1 - dotv * vy) / nr);
        atomicAdd(a.g_rays + (size_t)n * 6 + 5, (gv2 - dotv * vz) / nr);
      } else if (a.g_rays && act && h == 0) {
        atomicAdd(a.g_rays + (size_t)n * 6 + 3, gv0);
        atomicAdd(a.g_rays + (size_t)n * 6 + 4, gv1);
        atomicAdd(a.g_rays + (size_t)n * 6 + 5, gv2);
      }
    }
    save_rows<16>(gb, sv::K3G_DF, dF, s, h);
    {
      f32x16 acc[3];
      acc_zero<3>(acc);
      app_bwd_seg<3, 16>(acc, dF, lds + pkb::S3_BASIST, a.pk, pkb::REG_S3_LO, pkb::S3_LO_BASIST, lane);
      float dG[48];
      acc_copy<3>(dG, acc);
      save_rows<48>(gb, sv::K3G_DA, dG, s, h);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// d(weight)/d(alpha) along a ray.  weight_k = alpha_k T_k, T_k = prod_{j<k} p_j, p = 1-alpha+1e-10
//   dL/dalpha_k = gw_k T_k - (sum_{m>k} gw_m w_m) / p_k
// ------------------------------------------------------------------------------------------------

// static field, density phase backward: wave per ray, lane per sample.  The suffix sum is formed
// directly by walking the ray LAST tile first (transmittance carries of each tile start come from a
// forward pre-pass); total - prefix would cancel catastrophically when p -> 1e-10.
__global__ __launch_bounds__(64) void k_static_density_bwd(BwdArgs a, StaticW w, float* __restrict__ gf_rows) {
  __shared__ float carr[64];   // transmittance carries at each 64-sample step (S <= 4096)
  const int lane = threadIdx.x;
  const int n = blockIdx.x;
  if (n >= a.N) return;
  float vx, vy, vz;
  const float nrm = ray_norm(a.rays, n, a.ray_type, vx, vy, vz);
  if (a.g_weight) {
    float carry = 1.0f;
    for (int j0 = 0; j0 < a.S; j0 += 64) {
      if (lane == 0) carr[j0 >> 6] = carry;
      const int j = j0 + lane;
      const bool act = j < a.S;
      const int idx = n * a.S + (act ? j : 0);
      const bool vld = act && a.valid[idx] != 0;
      const float sigma = vld ? density_act(a.sp.raw[idx], a.act, a.density_shift) : 0.f;
      const float zj = act ? a.z[idx] : 0.f;
      const float zn = (j + 1 < a.S) ? a.z[idx + 1] : zj;
      const float ds = ((j + 1 < a.S) ? (zn - zj) : 0.0f) * nrm * a.distance_scale;
      const float alpha = 1.0f - expf(-sigma * ds);
      const float p = act ? one_minus_alpha_eps(alpha) : 1.0f;
      carry *= __shfl(scan_mul64(p, lane), 63, 64);
    }
  }
  __syncthreads();
  float sufcarry = 0.f, g_nrm = 0.f;
  const int ntile = (a.S + 63) >> 6;
  for (int tl = ntile - 1; tl >= 0; --tl) {
    const int j0 = tl << 6;
    const int j = j0 + lane;
    const bool act = j < a.S;
    const int idx = n * a.S + (act ? j : 0);
    const bool vld = act && a.valid[idx] != 0;
    const float f = a.sp.raw[idx];
    const float sigma = vld ? density_act(f, a.act, a.density_shift) : 0.f;
    const float zj = act ? a.z[idx] : 0.f;
    const float zn = (j + 1 < a.S) ? a.z[idx + 1] : zj;
    const float ds = ((j + 1 < a.S) ? (zn - zj) : 0.0f) * nrm * a.distance_scale;
    const float alpha = 1.0f - expf(-sigma * ds);
    const float p = act ? one_minus_alpha_eps(alpha) : 1.0f;
    float g_alpha = 0.f;
    if (a.g_weight) {
      const float incl = scan_mul64(p, lane);
      float excl = __shfl_up(incl, 1, 64);
      if (lane == 0) excl = 1.0f;
      const float T = carr[tl] * excl;
      const float gwv = act ? a.g_weight[idx] : 0.f;
      float rinc = gwv * alpha * T;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_down(rinc, d, 64);
        if (lane + d < 64) rinc += o;
      }
      float rex = __shfl_down(rinc, 1, 64);
      if (lane == 63) rex = 0.f;
      g_alpha = gwv * T - (sufcarry + rex) / p;
      sufcarry += __shfl(rinc, 0, 64);
    }
    float g_sigma = (act && a.g_sigma) ? a.g_sigma[idx] : 0.f;
    g_sigma += g_alpha * ds * (1.0f - alpha);
    if ((a.g_rays || a.g_z) && act) {  // dists = dz * |d| * scale: d(loss)/d|d| and d(loss)/dz
      const float g_ds = (a.g_dists ? a.g_dists[idx] : 0.f) + g_alpha * sigma * (1.0f - alpha);
      if (a.g_rays) g_nrm += g_ds * ((j + 1 < a.S) ? (zn - zj) : 0.0f) * a.distance_scale;
      if (a.g_z && j + 1 < a.S) {
        const float gz = g_ds * nrm * a.distance_scale;
        atomicAdd(a.g_z + idx, -gz);
        atomicAdd(a.g_z + idx + 1, gz);
      }
    }
    const float gf = vld ? g_sigma * act_grad(f, a.act, a.density_shift) : 0.f;
    // d(loss)/d(density feature) per sample; the VM gather backward (scatter + coordinate gradients)
    // runs in k_scatter with LDS line accumulators (bcast mode: the feature is the plain sum of the
    // 24 products, so every component has this same gradient)
    if (act) gf_rows[(size_t)n * (((a.S + 31) >> 5) << 5) + j] = gf;
  }
  if (a.g_rays && a.ray_type != RDRF_RAY_OTHER) {
    g_nrm = wave_sum(g_nrm);
    if (lane < 3) atomicAdd(a.g_rays + (size_t)n * 6 + 3 + lane, g_nrm * (lane == 0 ? vx : (lane == 1 ? vy : vz)));
  }
}

// ------------------------------------------------------------------------------------------------
// generic scatter kernel: VM gather backward of one or two factor sets from feature-gradient rows.
// No MFMA state and no weight image -> ~100 VGPRs, several workgroups per CU: the dependent
// shuffle-scan / atomic latency chains of different waves overlap (inside the fused MLP kernels,
// at 2 waves/SIMD, they were 85 % of the backward-data time).
// ------------------------------------------------------------------------------------------------
#ifndef SC_LINES_MAX_BYTES
/* LDS line-gradient accumulators per workgroup, sized per launch (dynamic LDS): up to 80 KB keeps two
   256-thread workgroups per CU; up to 152 KB runs one 512-thread workgroup per CU, still faster than
   sending the line gradients to global atomics (final stage appearance: 3.9 -> 2.6 ms) */
#define SC_LINES_MAX_BYTES (152 * 1024)
#endif
struct ScatterArgs {
  RdrfVM vm[2], gvm[2];
  int nsets;
  const float* rows;   // d(feature) rows: tile t, row r at rows + (t*stride + row0[set] + r)*32
  int stride, row0[2];
  const float* xw;     // [idx][3] normalised coordinates, or nullptr -> normalise xyz
  const float* xyz;
  Box box;
  const int* list;     // compacted mode: sample ids (+ device count); nullptr -> ray tiles
  const int* count;
  const uint8_t* valid;
  int N, S;
  float* dxw;          // [idx][3] coordinate gradients (nullable)
  int dxw_accumulate;
  float* g_xyz;        // static field: g_xyz += dw * inv (nullable)
  int lds_bytes;       // dynamic LDS for the line accumulators (0: lines go to global memory)
  int lds_f64;         // the accumulators are doubles (ds_add_f64: 11 x the update rate of ds_add_f32) / floats
  int bcast;           // 1: every component's gradient is row 0 of the tile (static density: the
                       //    feature is the plain sum of the 24 products)
};

template <int C0Q, int C1Q, int NQ>
__global__ __launch_bounds__(512, 3) void k_scatter(ScatterArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lacc[];
  const int nl0 = lines_floats(a.vm[0]), nl1 = a.nsets > 1 ? lines_floats(a.vm[1]) : 0;
  const bool use_lacc = a.lds_bytes > 0;   // host: (nl0 + nl1) elements of 8 (lds_f64) or 4 bytes if they fit SC_LINES_MAX_BYTES, else 0
  if (use_lacc)
    for (int i = threadIdx.x; i < (nl0 + nl1) * (a.lds_f64 ? 2 : 1); i += blockDim.x) lacc[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, h = lane >> 5, s = lane & 31;
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int tpr = (a.S + 31) >> 5;
  const int count = a.list ? *a.count : 0;
  const int ntiles = a.list ? ((count + 31) >> 5) : a.N * tpr;
  for (int t = blockIdx.x * nwaves + wave; t < ntiles; t += gridDim.x * nwaves) {
    int idx;
    bool live;
    if (a.list) {
      const int li = t * 32 + s;
      live = li < count;
      idx = live ? a.list[li] : 0;
    } else {
      const int n = t / tpr, j = (t - n * tpr) * 32 + s;
      const bool act = j < a.S;
      idx = n * a.S + (act ? j : 0);
      live = act && a.valid[idx] != 0;
    }
    float x0, x1, x2;
    if (a.xw) {
      x0 = a.xw[(size_t)idx * 3 + 0]; x1 = a.xw[(size_t)idx * 3 + 1]; x2 = a.xw[(size_t)idx * 3 + 2];
    } else {
      x0 = norm_c(a.xyz[(size_t)idx * 3 + 0], a.box.lo[0], a.box.inv[0]);
      x1 = norm_c(a.xyz[(size_t)idx * 3 + 1], a.box.lo[1], a.box.inv[1]);
      x2 = norm_c(a.xyz[(size_t)idx * 3 + 2], a.box.lo[2], a.box.inv[2]);
    }
    // coordinates / liveness of the samples this lane handles in the (quad, 16-sample) mapping of
    // the XY iterations: sample 16 j + (lane & 15), j = 0, 1 (lanes 0..31 hold samples 0..31)
    const float xa0 = __shfl(x0, lane & 15, 64), xa1 = __shfl(x1, lane & 15, 64), xa2 = __shfl(x2, lane & 15, 64);
    const float xb0 = __shfl(x0, 16 + (lane & 15), 64), xb1 = __shfl(x1, 16 + (lane & 15), 64),
                xb2 = __shfl(x2, 16 + (lane & 15), 64);
    const bool livea = __shfl((int)live, lane & 15, 64) != 0, liveb = __shfl((int)live, 16 + (lane & 15), 64) != 0;
    float dw0 = 0.f, dw1 = 0.f, dw2 = 0.f;
    for (int set = 0; set < a.nsets; ++set) {
      const float* rb = a.rows + ((size_t)t * a.stride + a.row0[set]) * 32;
      const LdsLines ll = make_lds_lines(use_lacc ? lacc : nullptr, set ? nl0 : 0, a.lds_f64, a.vm[set]);
#ifndef RDRF_SC_UNROLL
#define RDRF_SC_UNROLL 1
#endif
      // Per level: the XY plane's C0Q quads go four at a time over 16-sample sub-tiles
      // (gather_xy4_bwd: a texel's four quads in one instruction = one 64-byte request); then the
      // XZ / YZ quads, either column-split over the half-waves (ZSPLIT: density / blending, one
      // quad per wave iteration, two bilinear columns = one request) or one quad per half-wave.
      constexpr bool ZSPLIT = C0Q <= 4;
      constexpr int QPL = C0Q + 2 * C1Q, NLV = (2 * NQ) / QPL;
      static_assert(C0Q % 4 == 0 && NLV * QPL == 2 * NQ, "quad layout");
      const int q = lane >> 4, s16 = lane & 15;
#ifndef RDRF_ABL_SC_NLV
#define RDRF_ABL_SC_NLV 99
#endif
#pragma unroll 1
      for (int lv = 0; lv < NLV && lv < RDRF_ABL_SC_NLV; ++lv) {
#ifndef RDRF_ABL_SC_NOXY
#pragma unroll 1
        for (int it = 0; it < C0Q / 2; ++it) {
          const int grp = it >> 1, j = it & 1;
          const int g = lv * QPL + 4 * grp + q;
          const int sidx = 16 * j + s16;
          const float* rq = rb + (a.bcast ? (size_t)0 : (size_t)(4 * g) * 32) + sidx;
          const int rs = a.bcast ? 0 : 32;
          const f32x4 dq = {rq[0], rq[rs], rq[2 * rs], rq[3 * rs]};
          gather_xy4_bwd<C0Q, C1Q>(a.vm[set], a.gvm[set], lv, 4 * grp + q, j ? xb0 : xa0, j ? xb1 : xa1,
                                   j ? xb2 : xa2, dq, j ? liveb : livea, lane < 32 && q == j, dw0, dw1, dw2, ll);
        }
#endif
#ifndef RDRF_ABL_SC_NOZ
        if constexpr (ZSPLIT) {
#pragma unroll 1
          for (int zq = 0; zq < 2 * C1Q; ++zq) {
            const int g = lv * QPL + C0Q + zq;
            const float* rq = rb + (a.bcast ? (size_t)0 : (size_t)(4 * g) * 32) + s;
            const int rs = a.bcast ? 0 : 32;
            const f32x4 dq = {rq[0], rq[rs], rq[2 * rs], rq[3 * rs]};
            gather_zquad_bwd<C0Q, C1Q>(a.vm[set], a.gvm[set], g, h, x0, x1, x2, dq, live, s, dw0, dw1, dw2, ll);
          }
        } else {
#pragma unroll 1
          for (int zp = 0; zp < C1Q; ++zp) {
            const int g = lv * QPL + C0Q + 2 * zp + h;
            const float* rq = rb + (a.bcast ? (size_t)0 : (size_t)(4 * g) * 32) + s;
            const int rs = a.bcast ? 0 : 32;
            const f32x4 dq = {rq[0], rq[rs], rq[2 * rs], rq[3 * rs]};
            gather_quad_bwd<C0Q, C1Q, 1>(a.vm[set], a.gvm[set], g, x0, x1, x2, dq, live, s, dw0, dw1, dw2, ll);
          }
        }
#endif
      }
    }
    dw0 += __shfl_xor(dw0, 32, 64); dw1 += __shfl_xor(dw1, 32, 64); dw2 += __shfl_xor(dw2, 32, 64);
    if (live && h == 0) {
      if (a.dxw) {
        float* d = a.dxw + (size_t)idx * 3;
        if (a.dxw_accumulate) { d[0] += dw0; d[1] += dw1; d[2] += dw2; }
        else { d[0] = dw0; d[1] = dw1; d[2] = dw2; }
      }
      if (a.g_xyz) {
        a.g_xyz[(size_t)idx * 3 + 0] += dw0 * a.box.inv[0];
        a.g_xyz[(size_t)idx * 3 + 1] += dw1 * a.box.inv[1];
        a.g_xyz[(size_t)idx * 3 + 2] += dw2 * a.box.inv[2];
      }
    }
  }
  if (use_lacc) {
    __syncthreads();
    flush_lds_lines(lacc, a.lds_f64, 0, a.vm[0], a.gvm[0]);
    if (a.nsets > 1) flush_lds_lines(lacc, a.lds_f64, nl0, a.vm[1], a.gvm[1]);
  }
}

// ------------------------------------------------------------------------------------------------
// SORTED scatter of the density / blending gradients (dynamic field, ray path).
//
// With the reference initialiser the warp MLP moves the warped point by about a texel between consecutive samples of
// a ray, so the ray-tile scatter above finds runs of 1-2 samples and pays ~13 M memory-side atomic requests per launch
//  for gradient planes of a few MB.  Here the live samples are first grouped by the plane CELL they fall
// into (one stable device-wide radix sort of (plane | level-0 cell) keys, rdrf_sort.hip), once per plane, and each
// plane is scattered in that order by the SAME per-quad device functions: consecutive lanes now hold samples of the
// same or the neighbouring cell (~20-40 samples per level-0 cell at the benchmark shapes), so the in-register run
// reduction collapses them and a run of lanes issues one request per tap.  Per plane: XY = 16 samples x 4 quads per
// wave step (gather_xy4_bwd), XZ / YZ = 32 samples, the half-waves taking the two bilinear columns (gather_zquad_bwd).
// d(features) come from the sample-major records the heads kernel writes (BwdArgs::dfs); coordinate gradients are
// accumulated into dxw by the three launches in turn (a sample is owned by one lane per launch: no atomics).
// ------------------------------------------------------------------------------------------------
struct SortKeyArgs {
  const float* xw;
  const uint8_t* valid;
  const float* grows1;   // K1G_SM rows 3 / 4 hold g_fd / g_fb: a sample with both zero scatters nothing
  const int* list;       // appearance: the compacted sample ids (entry li of the key arrays = compacted sample li) ...
  const int* count;      // ... and their device count; nullptr: every sample of the batch, liveness from valid / grows1
  int N, S;
  int W[3], H[3];        // level-0 plane sizes
  int kb;                // bits of the cell part of the key
  unsigned* keys;        // [3][N*S]; compact: [3][*count]
  int* counts;           // [3] live entries per plane
  int compact;           // list mode: the key arrays hold the *count compacted entries of each plane back to back (stride *count
                         // instead of N*S), so that the sort and the key generation touch live entries only (round 6)
};

RDRF_D int cell_axis(float c, int L, bool& any) {
  // level-0 tap index clamped to [-2, L]; `any`: some stride level has an in-range tap on this axis
  const Tap1 t0 = tap1d(c, L), t1 = tap1d(c, (L + 1) >> 1), t2 = tap1d(c, (L + 3) >> 2);
  any = t0.ok0 || t0.ok1 || t1.ok0 || t1.ok1 || t2.ok0 || t2.ok1;
  return min(max(t0.i0, -2), L) + 2;
}

__global__ __launch_bounds__(256) void k_sort_keys(SortKeyArgs a) {
  const int NS = a.N * a.S, tpr = (a.S + 31) >> 5;
  const int count = a.list ? *a.count : 0;
  const int nent = (a.list && a.compact) ? count : NS;   // entries per plane = the stride of the key arrays
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nent; e += gridDim.x * blockDim.x) {
    bool live = false;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    if (a.list) {
      live = e < count;
      const int idx = live ? a.list[e] : 0;
      x0 = a.xw[(size_t)idx * 3 + 0]; x1 = a.xw[(size_t)idx * 3 + 1]; x2 = a.xw[(size_t)idx * 3 + 2];
    } else {
      const int idx = e;
      const int n = idx / a.S, j = idx - n * a.S;
      const float* sm = a.grows1 + ((size_t)(n * tpr + (j >> 5)) * sv::K1G_ROWS + sv::K1G_SM) * 32 + (j & 31);
      live = a.valid[idx] != 0 && (sm[3 * 32] != 0.f || sm[4 * 32] != 0.f);
      x0 = a.xw[(size_t)idx * 3 + 0]; x1 = a.xw[(size_t)idx * 3 + 1]; x2 = a.xw[(size_t)idx * 3 + 2];
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const float cx = p == 2 ? x1 : x0, cy = p == 0 ? x1 : x2;
      bool ax, ay;
      const int ix = cell_axis(cx, a.W[p], ax), iy = cell_axis(cy, a.H[p], ay);
      const bool in = live && ax && ay;
      const unsigned cell = in ? (unsigned)(iy * (a.W[p] + 3) + ix) : ((1u << a.kb) - 1u);
      a.keys[(size_t)p * nent + e] = ((unsigned)p << a.kb) | cell;
    }
  }
}

// live entries per plane = position of the first dropped key of the plane in the sorted array (a per-wave atomic
// counter in k_sort_keys serialised 66 k same-address atomics: 240 us)
__global__ void k_sort_counts(const unsigned* __restrict__ keys_sorted, int NS, int kb, int* __restrict__ counts,
                              const int* __restrict__ seg) {
  const int p = threadIdx.x;
  if (p >= 3) return;
  const int stride = seg ? *seg : NS;   // compact key arrays: the planes' segments are *seg entries long
  const unsigned drop = ((unsigned)p << kb) | ((1u << kb) - 1u);
  const unsigned* k = keys_sorted + (size_t)p * stride;
  int lo = 0, hi = stride;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (k[mid] < drop) lo = mid + 1; else hi = mid;
  }
  counts[p] = lo;
}

#ifndef RDRF_LINE_DIRECT_DEFAULT
#define RDRF_LINE_DIRECT_DEFAULT 1   // measured: -0.1 ms / step on top of the double accumulators (profiles/r05_ab_lds_f64.txt)
#endif
struct SortedScatterArgs {
  RdrfVM vm[2], gvm[2];
  int set_mask;            // bit k: set k has a gradient
  const unsigned* order;   // [NS] sorted positions of THIS plane (value = plane * NS + entry index)
  const int* count;        // live entries of this plane
  unsigned base;           // plane * NS
  const int* seg;          // compact key arrays: `order` is the base of the whole array, this plane's positions start at
                           // PLANE * *seg and carry that base (nullptr: order / base as given)
  const float* dfs;        // records: entry e at dfs + e * rec_floats (+ set * floats per set)
  int rec_floats;
  const int* list;         // appearance: entry e is compacted sample e, its sample id is list[e]; nullptr: entry = sample id
  const float* xw;
  float* dxw;              // += coordinate gradients
  int lds_bytes, lds_f64;  // line accumulator of this pass in LDS: bytes (0: global atomics), doubles / floats
  int line_direct;         // no run reduction in front of the LDS line updates (see LdsLines::direct)
  // k_scatter_tiled: the sorted keys of this plane (a slice's window is anchored at its first key), bits of the cell part
  // of a key, key-space row length W + 3, window width in cells, wave steps per slice, windows allocated (factor sets)
  const unsigned* keys;
  int kb, Wk, tw, slice_steps, tile_sets;
};

// C0Q / C1Q: quads of an XY / XZ-YZ texel: <4, 1> density and blending ({16,4,4} components, two sets per record),
// <12, 3> appearance ({48,12,12}, one set, entries = the compacted list)
template <int PLANE, int C0Q, int C1Q>
__global__ __launch_bounds__(512, 3) void k_scatter_sorted(SortedScatterArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lacc[];
  // pass PLANE touches line PLANE only (the partner of its plane): one line per factor set in the accumulator
  const int nl0 = (a.set_mask & 1) ? a.vm[0].L[PLANE] * lds_stride(a.vm[0].C[PLANE]) : 0;   // (no accumulator for a factor set
  const int nl1 = (a.set_mask & 2) ? a.vm[1].L[PLANE] * lds_stride(a.vm[1].C[PLANE]) : 0;   //  without a gradient in this launch)
  const bool use_lacc = a.lds_bytes > 0;
  if (use_lacc)
    for (int i = threadIdx.x; i < (nl0 + nl1) * (a.lds_f64 ? 2 : 1); i += blockDim.x) lacc[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, h = lane >> 5, s = lane & 31, q = lane >> 4, s16 = lane & 15;
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int count = *a.count;
  const unsigned obase = a.seg ? (unsigned)PLANE * (unsigned)*a.seg : a.base;   // compact key arrays (SortKeyArgs::compact)
  const unsigned* order = a.seg ? a.order + obase : a.order;
  constexpr int SPT = PLANE == 0 ? 16 : 32;   // samples per wave step
  // record of one set: [XY level 0 | 1 | 2 (XYF floats each)] [XZ: 3 x ZF] [YZ: 3 x ZF]
  constexpr int XYF = 4 * C0Q, ZF = 4 * C1Q, SETF = 3 * XYF + 6 * ZF, QPL = C0Q + 2 * C1Q;
  const int ntiles = (count + SPT - 1) / SPT;
  for (int t = blockIdx.x * nwaves + wave; t < ntiles; t += gridDim.x * nwaves) {
    const int pos = t * SPT + (PLANE == 0 ? s16 : s);
    const bool live = pos < count;
    const int ent = live ? (int)(order[pos] - obase) : 0;
    const int idx = a.list ? a.list[ent] : ent;
    const float x0 = a.xw[(size_t)idx * 3 + 0], x1 = a.xw[(size_t)idx * 3 + 1], x2 = a.xw[(size_t)idx * 3 + 2];
    float dw0 = 0.f, dw1 = 0.f, dw2 = 0.f;
#pragma unroll 1
    for (int set = 0; set < 2; ++set) {
      if (!((a.set_mask >> set) & 1)) continue;
      const int first = set ? nl0 : 0;
      const LdsLines ll = LdsLines{use_lacc ? lacc : nullptr, {first, first, first}, a.lds_f64, a.line_direct};
      const float* rec = a.dfs + (size_t)ent * a.rec_floats + set * SETF;
#pragma unroll 1
      for (int lv = 0; lv < 3; ++lv) {
        if constexpr (PLANE == 0) {
#pragma unroll 1
          for (int grp = 0; grp < C0Q / 4; ++grp) {
            const f32x4 dq = live ? ld4(rec + lv * XYF + grp * 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            gather_xy4_bwd<C0Q, C1Q>(a.vm[set], a.gvm[set], lv, 4 * grp + q, x0, x1, x2, dq, live, q == 0, dw0, dw1, dw2, ll);
          }
        } else {
#pragma unroll 1
          for (int zq = 0; zq < C1Q; ++zq) {
            const f32x4 dq = live ? ld4(rec + 3 * XYF + (PLANE - 1) * 3 * ZF + lv * ZF + 4 * zq) : f32x4{0.f, 0.f, 0.f, 0.f};
            gather_zquad_bwd<C0Q, C1Q>(a.vm[set], a.gvm[set], lv * QPL + C0Q + (PLANE - 1) * C1Q + zq, h, x0, x1, x2, dq, live, s,
                                       dw0, dw1, dw2, ll);
          }
        }
      }
    }
    if constexpr (PLANE != 0) {   // both halves computed half of every coordinate gradient
      dw0 += __shfl_xor(dw0, 32, 64); dw1 += __shfl_xor(dw1, 32, 64); dw2 += __shfl_xor(dw2, 32, 64);
    }
    if (live && (PLANE == 0 ? q == 0 : h == 0)) {
      float* d = a.dxw + (size_t)idx * 3;
      d[0] += dw0; d[1] += dw1; d[2] += dw2;
    }
  }
  if (use_lacc) {
    __syncthreads();
    if (a.set_mask & 1) flush_lds_line(lacc, a.lds_f64, 0, a.vm[0].L[PLANE], a.vm[0].C[PLANE], a.gvm[0].line[PLANE]);
    if (a.set_mask & 2) flush_lds_line(lacc, a.lds_f64, nl0, a.vm[1].L[PLANE], a.vm[1].C[PLANE], a.gvm[1].line[PLANE]);
  }
}

// ------------------------------------------------------------------------------------------------
// TILED sorted scatter.  After the LDS line accumulators became doubles (ds_add_f64), what was left of the sorted passes
// was their memory-side plane atomics: 1.08 of 1.85 ms per step for the density / blending scatter (ablation builds,
// profiles/r05_scatter_ablation.txt) although the sort had already cut the REQUESTS tenfold -- every run tail still pays a
// round trip to the memory side, and vmcnt is one in-order counter, so the next gathers wait behind it.  Neighbouring
// plane cells are neighbours in the sorted array, so a SLICE of consecutive sorted entries (slice_steps wave steps of
// one workgroup) touches a window of ~(tw + 1) x 2 texels per stride level, anchored at the slice's first cell: the
// window lives in LDS as doubles, the run tails add into it with ds_add_f64, and the workgroup flushes it with one
// coalesced sweep (64-byte texels = one request per 16 lanes).  Slices have equal size (static stride over the workgroups:
// the load is balanced by construction; windows by CELL ranges with a dynamic queue were 2 x slower than the untiled
// kernel, the ray density per cell varies too much).  Everything per entry (taps, run reduction, cross-run merge,
// coordinate gradients, line updates) is the code of k_scatter_sorted; only the destination of a plane sum differs
// (plane_add4), and a tap outside the window (the slice ran into the next key row, a sparse region, a clamped key)
// still goes to global memory, so the result is the same sum in a different order.
// ------------------------------------------------------------------------------------------------
RDRF_HD int tile_texels(int tw) { return (tw + 1) * 2 + (tw / 2 + 3) * 3 + (tw / 4 + 3) * 3; }

template <int PLANE, int C0Q, int C1Q>
__global__ __launch_bounds__(512, 4) void k_scatter_tiled(SortedScatterArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lacc[];   // doubles: [line set 0 | line set 1 | tile set 0 | tile set 1]
  __shared__ int s_geo[24];
  constexpr int CT = PLANE == 0 ? 4 * C0Q : 4 * C1Q;   // components per texel of this plane
  // (a factor set without a gradient in this launch gets no accumulator)
  const int nl0 = (a.set_mask & 1) ? a.vm[0].L[PLANE] * lds_stride(a.vm[0].C[PLANE]) : 0;
  const int nl1 = (a.set_mask & 2) ? a.vm[1].L[PLANE] * lds_stride(a.vm[1].C[PLANE]) : 0;
  const int tile_elems = tile_texels(a.tw) * CT;
  for (int i = threadIdx.x; i < (nl0 + nl1 + a.tile_sets * tile_elems) * 2; i += blockDim.x) lacc[i] = 0.f;
  const int lane = threadIdx.x & 63, h = lane >> 5, s = lane & 31, q = lane >> 4, s16 = lane & 15;
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int count = *a.count;
  constexpr int SPT = PLANE == 0 ? 16 : 32;
  constexpr int XYF = 4 * C0Q, ZF = 4 * C1Q, SETF = 3 * XYF + 6 * ZF, QPL = C0Q + 2 * C1Q;
  const int W = a.vm[0].W[PLANE], H = a.vm[0].H[PLANE];
  const unsigned cmask = (1u << a.kb) - 1u;
  const int E = a.slice_steps * SPT;   // entries per slice
  for (int sl = blockIdx.x; (long)sl * E < count; sl += gridDim.x) {
    __syncthreads();   // (first pass: the zero fill; later: the previous slice's flush has read s_geo)
    const int e_lo = sl * E, e_hi = min(e_lo + E, count);
    if (threadIdx.x == 0) {   // the slice's windows, one per stride level (PlaneTile), anchored at its first (= smallest) cell
      const int cell = (int)(a.keys[e_lo] & cmask), r = cell / a.Wk, kx = cell - r * a.Wk;
      const int ix_lo = kx - 2, ix_hi = min(kx + a.tw, a.Wk) - 3, iy = r - 2;
      int off = 0;
      for (int lv = 0; lv < 3; ++lv) {
        const int st = 1 << lv, Ws = (W + st - 1) >> lv, Hs = (H + st - 1) >> lv;
        int x0, x1, y0, y1;
        if (lv == 0) {
          x0 = max(ix_lo, 0); x1 = min(ix_hi + 1, Ws - 1); y0 = max(iy, 0); y1 = min(iy + 1, Hs - 1);
        } else {   // f_lv = f_0 (Ws - 1) / (W - 1): the taps of f_0 in [i, j + 1) are floor(i rho) .. floor((j + 1) rho) + 1
          const float rx = W > 1 ? (float)(Ws - 1) / (float)(W - 1) : 0.f, ry = H > 1 ? (float)(Hs - 1) / (float)(H - 1) : 0.f;
          x0 = max((int)floorf((float)ix_lo * rx), 0); x1 = min((int)floorf((float)(ix_hi + 1) * rx) + 1, Ws - 1);
          y0 = max((int)floorf((float)iy * ry), 0); y1 = min((int)floorf((float)(iy + 1) * ry) + 1, Hs - 1);
        }
        const int nxcap = lv == 0 ? a.tw + 1 : (a.tw >> lv) + 3, nycap = lv == 0 ? 2 : 3;
        const int nx = min(max(x1 - x0 + 1, 0), nxcap), ny = min(max(y1 - y0 + 1, 0), nycap);
        s_geo[lv * 8 + 0] = x0; s_geo[lv * 8 + 1] = y0; s_geo[lv * 8 + 2] = nx; s_geo[lv * 8 + 3] = ny; s_geo[lv * 8 + 4] = off;
        off += nx * ny * CT;
      }
    }
    __syncthreads();
    for (int t = wave; t * SPT < e_hi - e_lo; t += nwaves) {
      const int pos = e_lo + t * SPT + (PLANE == 0 ? s16 : s);
      const bool live = pos < e_hi;
      const int ent = live ? (int)(a.order[pos] - a.base) : 0;
      const int idx = a.list ? a.list[ent] : ent;
      const float x0 = a.xw[(size_t)idx * 3 + 0], x1 = a.xw[(size_t)idx * 3 + 1], x2 = a.xw[(size_t)idx * 3 + 2];
      float dw0 = 0.f, dw1 = 0.f, dw2 = 0.f;
#pragma unroll 1
      for (int set = 0; set < 2; ++set) {
        if (!((a.set_mask >> set) & 1)) continue;
        const int first = set ? nl0 : 0;
        const LdsLines ll = LdsLines{lacc, {first, first, first}, 1, a.line_direct};
        const int tslot = set == 1 && (a.set_mask & 1) ? 1 : 0;   // windows are allocated for the live sets only
        const PlaneTile T = PlaneTile{lacc + 2 * (nl0 + nl1 + tslot * tile_elems), s_geo, CT};
        const float* rec = a.dfs + (size_t)ent * a.rec_floats + set * SETF;
#pragma unroll 1
        for (int lv = 0; lv < 3; ++lv) {
          if constexpr (PLANE == 0) {
#pragma unroll 1
            for (int grp = 0; grp < C0Q / 4; ++grp) {
              const f32x4 dq = live ? ld4(rec + lv * XYF + grp * 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
              gather_xy4_bwd<C0Q, C1Q, true>(a.vm[set], a.gvm[set], lv, 4 * grp + q, x0, x1, x2, dq, live, q == 0, dw0, dw1, dw2, ll, T);
            }
          } else {
#pragma unroll 1
            for (int zq = 0; zq < C1Q; ++zq) {
              const f32x4 dq = live ? ld4(rec + 3 * XYF + (PLANE - 1) * 3 * ZF + lv * ZF + 4 * zq) : f32x4{0.f, 0.f, 0.f, 0.f};
              gather_zquad_bwd<C0Q, C1Q, true>(a.vm[set], a.gvm[set], lv * QPL + C0Q + (PLANE - 1) * C1Q + zq, h, x0, x1, x2, dq, live, s,
                                               dw0, dw1, dw2, ll, T);
            }
          }
        }
      }
      if constexpr (PLANE != 0) {
        dw0 += __shfl_xor(dw0, 32, 64); dw1 += __shfl_xor(dw1, 32, 64); dw2 += __shfl_xor(dw2, 32, 64);
      }
      if (live && (PLANE == 0 ? q == 0 : h == 0)) {
        float* d = a.dxw + (size_t)idx * 3;
        d[0] += dw0; d[1] += dw1; d[2] += dw2;
      }
    }
    __syncthreads();
    // flush the windows: component fastest, so 16 consecutive lanes cover one 64-byte texel (one request); the slots are
    // left zeroed for the next chunk
#pragma unroll 1
    for (int set = 0; set < 2; ++set) {
      if (!((a.set_mask >> set) & 1)) continue;
      double* tile = reinterpret_cast<double*>(lacc) + nl0 + nl1 + (set == 1 && (a.set_mask & 1) ? 1 : 0) * tile_elems;
      float* GP = a.gvm[set].plane[PLANE];
      const int sH = a.vm[set].sH[PLANE], sW = a.vm[set].sW[PLANE];
      for (int lv = 0; lv < 3; ++lv) {
        const int gx0 = s_geo[lv * 8 + 0], gy0 = s_geo[lv * 8 + 1], nx = s_geo[lv * 8 + 2], ny = s_geo[lv * 8 + 3], off = s_geo[lv * 8 + 4];
        const int n = nx * ny * CT;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
          const double v = tile[off + i];
          if (v != 0.0) {
            tile[off + i] = 0.0;
            const int t = i / CT, cc = i - t * CT, ty = t / nx, tx = t - ty * nx;
            grad_add(GP + (size_t)(((gy0 + ty) << lv) * sH + ((gx0 + tx) << lv) * sW) + cc, (float)v);
          }
        }
      }
    }
  }
  __syncthreads();
  if (a.set_mask & 1) flush_lds_line(lacc, 1, 0, a.vm[0].L[PLANE], a.vm[0].C[PLANE], a.gvm[0].line[PLANE]);
  if (a.set_mask & 2) flush_lds_line(lacc, 1, nl0, a.vm[1].L[PLANE], a.vm[1].C[PLANE], a.gvm[1].line[PLANE]);
}

// ------------------------------------------------------------------------------------------------
// dynamic field, density / blending / warp backward-data: wave per ray, 32-sample tiles, in two
// phases around the scatter kernel:
//   PHASE 0 (heads): weight/sigma/blending backward, density + blending head backward ->
//                    d(feature) rows for k_scatter, d(X0) partial rows, dz rows
//   PHASE 1 (warp) : coordinate gradients (appearance + density + blending scatter) ->
//                    warp MLP backward, positional-encoding backward, g_xyz, d(tout)
// ------------------------------------------------------------------------------------------------
// Reduce-scatter over the 32 lanes of a half-wave: on return lane s holds the sum over the half's 32 lanes of p[s]
// (five butterfly stages: 31 lane exchanges + 31 adds).  Used for the weight gradients of the 3- and 1-row layers of the
// density phase (layer5, density / blending layer2): dW[e] = sum over the tile's samples of dz(sample) * in_e(sample),
// with dz a per-lane scalar and in_e the 32 slots the lane already holds -- as MFMA products in k_dw2 these were 6 of
// the 40 per tile, each 27/32 empty.  ALL lanes of the wave must call.
RDRF_D float reduce_scatter32(const float (&p)[32], int s) {
  const bool b4 = s & 16, b3 = s & 8, b2 = s & 4, b1 = s & 2, b0 = s & 1;
  float q[16], r[8], t[4], u[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) q[i] = (b4 ? p[i + 16] : p[i]) + __shfl_xor(b4 ? p[i] : p[i + 16], 16, 64);
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (b3 ? q[i + 8] : q[i]) + __shfl_xor(b3 ? q[i] : q[i + 8], 8, 64);
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = (b2 ? r[i + 4] : r[i]) + __shfl_xor(b2 ? r[i] : r[i + 4], 4, 64);
#pragma unroll
  for (int i = 0; i < 2; ++i) u[i] = (b1 ? t[i + 2] : t[i]) + __shfl_xor(b1 ? t[i] : t[i + 2], 2, 64);
  return (b0 ? u[1] : u[0]) + __shfl_xor(b0 ? u[0] : u[1], 1, 64);
}

template <int PHASE, bool FEAT>
__global__ __launch_bounds__(64 * RDRF_MAXW) void k_dyn_density_bwd(BwdArgs a, DynW w, DynG gw) {
  __shared__ __attribute__((aligned(16))) float lds[PHASE == 0 ? pkb::K1H_SIZE : pkb::K1W_SIZE];
  __shared__ float carr[8][128];  // per-wave transmittance carries at tile starts (S <= 4096)
  lds_fill(lds, a.pk + (PHASE == 0 ? pkb::REG_K1H : pkb::REG_K1W), PHASE == 0 ? pkb::K1H_SIZE : pkb::K1W_SIZE);
  const int lane = threadIdx.x & 63, h = lane >> 5, s = lane & 31;
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int tpr = (a.S + 31) >> 5;
  // small-layer weight gradients of this wave (ray path): lane (s, h) holds input element elem_of(s, h) of up to three
  // output rows (PHASE 0: density / blending layer2; PHASE 1: the three rows of layer5) + the rows' bias sums per lane
  float sw[3] = {0.f, 0.f, 0.f}, sb[3] = {0.f, 0.f, 0.f};
  for (int n = blockIdx.x * nwaves + wave; n < a.N; n += gridDim.x * nwaves) {
    float vx = 0.f, vy = 0.f, vz = 0.f;
    float nrm = 1.0f;
    if constexpr (!FEAT) nrm = ray_norm(a.rays, n, a.ray_type, vx, vy, vz);
    if (!FEAT && PHASE == 0 && a.g_weight) {  // pre-pass: transmittance at each tile start
      float carry = 1.0f;
      for (int j0 = 0; j0 < a.S; j0 += 32) {
        if (lane == 0) carr[wave][j0 >> 5] = carry;
        const int j = j0 + s;
        const bool act = j < a.S;
        const int idx = n * a.S + (act ? j : 0);
        const bool vld = act && a.valid[idx] != 0;
        const float sigma = vld ? density_act(a.sp.raw[(size_t)idx * 2], a.act, a.density_shift) : 0.f;
        const float zj = act ? a.z[idx] : 0.f;
        const float zn = (j + 1 < a.S) ? a.z[idx + 1] : zj;
        const float ds = ((j + 1 < a.S) ? (zn - zj) : 0.0f) * nrm * a.distance_scale;
        const float alpha = 1.0f - expf(-sigma * ds);
        const float p = act ? one_minus_alpha_eps(alpha) : 1.0f;
        carry *= __shfl(scan_mul32(p, s), 31, 32);
      }
    }
    float sufcarry = 0.f, g_nrm = 0.f;
    float dTacc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) dTacc[i] = 0.f;
    for (int tli = tpr - 1; tli >= 0; --tli) {  // LAST tile first: direct suffix sums
      const int j0 = tli << 5;
      const int j = j0 + s;
      const bool act = j < a.S && (!FEAT || n * a.S + j < a.M);
      const int idx = n * a.S + (act ? j : 0);
      const bool vld = act && (FEAT || a.valid[idx] != 0);
      const size_t tl = (size_t)n * tpr + tli;
      const float* svb = a.sp.act1 + tl * sv::K1_ROWS * 32;
      float* gb = a.grows1 + tl * sv::K1G_ROWS * 32;
      if (PHASE == 0) {
        float g_fd, g_fb;
        if constexpr (FEAT) {  // the gradients of the raw head outputs arrive directly
          g_fd = (act && a.g_sigma) ? a.g_sigma[idx] : 0.f;
          g_fb = (act && a.g_blending) ? a.g_blending[idx] : 0.f;
        } else {
        const float fd = a.sp.raw[(size_t)idx * 2], fb = a.sp.raw[(size_t)idx * 2 + 1];
        const float sigma = vld ? density_act(fd, a.act, a.density_shift) : 0.f;
        const float zj = act ? a.z[idx] : 0.f;
        const float zn = (j + 1 < a.S) ? a.z[idx + 1] : zj;
        const float ds = ((j + 1 < a.S) ? (zn - zj) : 0.0f) * nrm * a.distance_scale;
        const float alpha = 1.0f - expf(-sigma * ds);
        const float p = act ? one_minus_alpha_eps(alpha) : 1.0f;
        float g_alpha = 0.f;
        if (a.g_weight) {
          const float incl = scan_mul32(p, s);
          float excl = __shfl_up(incl, 1, 32);
          if (s == 0) excl = 1.0f;
          const float T = carr[wave][tli] * excl;
          const float gwv = act ? a.g_weight[idx] : 0.f;
          float rinc = gwv * alpha * T;
#pragma unroll
          for (int d = 1; d < 32; d <<= 1) {
            const float o = __shfl_down(rinc, d, 32);
            if (s + d < 32) rinc += o;
          }
          float rex = __shfl_down(rinc, 1, 32);
          if (s == 31) rex = 0.f;
          g_alpha = gwv * T - (sufcarry + rex) / p;
          sufcarry += __shfl(rinc, 0, 32);
        }
        float g_sigma = (act && a.g_sigma) ? a.g_sigma[idx] : 0.f;
        g_sigma += g_alpha * ds * (1.0f - alpha);
        if ((a.g_rays || a.g_z) && act && h == 0) {
          const float g_ds = (a.g_dists ? a.g_dists[idx] : 0.f) + g_alpha * sigma * (1.0f - alpha);
          if (a.g_rays) g_nrm += g_ds * ((j + 1 < a.S) ? (zn - zj) : 0.0f) * a.distance_scale;
          if (a.g_z && j + 1 < a.S) {
            const float gz = g_ds * nrm * a.distance_scale;
            atomicAdd(a.g_z + idx, -gz);
            atomicAdd(a.g_z + idx + 1, gz);
          }
        }
        g_fd = vld ? g_sigma * act_grad(fd, a.act, a.density_shift) : 0.f;
        const float bl = sigmoidf_(fb);
        g_fb = (vld && a.g_blending) ? a.g_blending[idx] * bl * (1.0f - bl) : 0.f;
        }
        f32x16 accX[2];  // d(X0) of the density and blending heads
        acc_zero<2>(accX);
        // head liveness (wave-uniform): a head whose output no loss reaches is not differentiated -- the reference's
        // autograd never visits that subgraph either (passes B-D of the trainer: no term depends on the blending head
        // before the late mask terms) -- so its 160 MFMAs per tile, its dz / d(feature) rows, its scatter set and its
        // dW products (host side) all drop out
        const bool live_d = FEAT ? a.g_sigma != nullptr : (a.g_sigma != nullptr || a.g_weight != nullptr);
        const bool live_b = a.g_blending != nullptr;
#pragma unroll
        for (int head = 0; head < 2; ++head) {
          if (!(head == 0 ? live_d : live_b)) continue;
          const float gfh = head == 0 ? g_fd : g_fb;
          float Hh[32], dzh[32];
          load_rows<32>(svb, head == 0 ? sv::K1_HD : sv::K1_HB, Hh, s, h);
          const float* w2 = lds + (head == 0 ? pkb::K1H_DEN2 : pkb::K1H_BLE2) + h * 32;
          float w2r[32];   // eight 16-byte LDS reads up front (element-wise reads were 32 x {ds_read_b32, lgkmcnt(0)})
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(w2 + 4 * q);
            w2r[4 * q] = v.x; w2r[4 * q + 1] = v.y; w2r[4 * q + 2] = v.z; w2r[4 * q + 3] = v.w;
          }
#pragma unroll
          for (int kk = 0; kk < 32; ++kk) dzh[kk] = Hh[kk] > 0.f ? w2r[kk] * gfh : 0.f;
          save_rows<32>(gb, head == 0 ? sv::K1G_DZD : sv::K1G_DZB, dzh, s, h);
          if (!FEAT && a.small_dw) {   // d(layer2 weight) = sum over the samples of g * (layer2 input = the saved activation)
            float pw[32];
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) pw[kk] = gfh * Hh[kk];
            sw[head] += reduce_scatter32(pw, s);
            sb[head] += gfh;
          }
          f32x16 accF[3];
          acc_zero<3>(accF);
#ifdef RDRF_HEADS_BWD_F32
          mfma_seg<3, 32>(accF, dzh, lds + (head == 0 ? pkb::K1H_DEN1T_F : pkb::K1H_BLE1T_F), lane);
          mfma_seg<2, 32>(accX, dzh, lds + (head == 0 ? pkb::K1H_DEN1T_X0 : pkb::K1H_BLE1T_X0), lane);
#else
          mfma_seg_b3_pair<3, 2, 32>(accF, accX, dzh, lds + (head == 0 ? pkb::K1H_DEN1T_F : pkb::K1H_BLE1T_F),
                                     lds + (head == 0 ? pkb::K1H_DEN1T_X0 : pkb::K1H_BLE1T_X0), lane);
#endif
          float dFh[48];
          acc_copy<3>(dFh, accF);
          if (!FEAT && a.dfs != nullptr) {
            // sample-major record for the sorted scatter: this lane half holds the feature quads Q = 2 m + h
            // (slots 4m..4m+3), Q = 6 level + {0..3: XY quad, 4: XZ, 5: YZ}; one 16-byte store per quad
            if (act) {
              float* rec = a.dfs + (size_t)idx * DFS_FLOATS + head * 72;
#pragma unroll
              for (int m = 0; m < 9; ++m) {
                const int off = h ? dfs_off(2 * m + 1) : dfs_off(2 * m);
                *reinterpret_cast<f32x4*>(rec + off) = f32x4{dFh[4 * m], dFh[4 * m + 1], dFh[4 * m + 2], dFh[4 * m + 3]};
              }
            }
          } else {
            save_rows<48>(gb, head == 0 ? sv::K1G_DFD : sv::K1G_DFB, dFh, s, h);
          }
        }
        float dXh[32];
        acc_copy<2>(dXh, accX);
        save_rows<32>(gb, sv::K1G_DX0, dXh, s, h);
        if (h == 0) {
          gb[(size_t)(sv::K1G_SM + 3) * 32 + s] = g_fd;
          gb[(size_t)(sv::K1G_SM + 4) * 32 + s] = g_fb;
        }
      } else {
        // coordinate gradients: appearance phase + density/blending scatter (already summed)
        float dw0 = act ? a.dxw_app[(size_t)idx * 3 + 0] : 0.f, dw1 = act ? a.dxw_app[(size_t)idx * 3 + 1] : 0.f,
              dw2 = act ? a.dxw_app[(size_t)idx * 3 + 2] : 0.f;
        float dn0 = act ? a.dxn_app[(size_t)idx * 3 + 0] : 0.f, dn1 = act ? a.dxn_app[(size_t)idx * 3 + 1] : 0.f,
              dn2 = act ? a.dxn_app[(size_t)idx * 3 + 2] : 0.f;
        // xw = normalize(unnormalize(xn) + delta); xyz_prime = xyz + delta
        float dd0 = dw0 * a.box.inv[0], dd1 = dw1 * a.box.inv[1], dd2 = dw2 * a.box.inv[2];
        float gp0 = 0.f, gp1 = 0.f, gp2 = 0.f;
        if (act && a.g_xyz_prime) {
          gp0 = a.g_xyz_prime[(size_t)idx * 3 + 0]; gp1 = a.g_xyz_prime[(size_t)idx * 3 + 1];
          gp2 = a.g_xyz_prime[(size_t)idx * 3 + 2];
        }
        dd0 += gp0; dd1 += gp1; dd2 += gp2;
        if (!act) { dd0 = dd1 = dd2 = 0.f; }
        if (h == 0) {  // small-layer dz rows 0..2 (rows 3,4 were written by phase 0)
          gb[(size_t)(sv::K1G_SM + 0) * 32 + s] = dd0; gb[(size_t)(sv::K1G_SM + 1) * 32 + s] = dd1;
          gb[(size_t)(sv::K1G_SM + 2) * 32 + s] = dd2;
        }
        // ---- warp MLP backward: layer5 (VALU) -> layer4 -> layer3
        float dz4[32];
        {
          float H4[32];
          load_rows<32>(svb, sv::K1_H4, H4, s, h);
          const float* w5 = lds + pkb::K1W_W5 + h * 32;
#pragma unroll
          for (int q = 0; q < 8; ++q) {   // 16-byte LDS reads (see the heads above)
            const f32x4 wa = *reinterpret_cast<const f32x4*>(w5 + 4 * q);
            const f32x4 wb = *reinterpret_cast<const f32x4*>(w5 + 64 + 4 * q);
            const f32x4 wc = *reinterpret_cast<const f32x4*>(w5 + 128 + 4 * q);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float d = wa[c] * dd0 + wb[c] * dd1 + wc[c] * dd2;
              dz4[4 * q + c] = H4[4 * q + c] > 0.f ? d : 0.f;
            }
          }
          if (!FEAT && a.small_dw) {   // d(layer5 weight): three output rows against the 64 saved inputs
            sb[0] += dd0; sb[1] += dd1; sb[2] += dd2;
#pragma unroll 1
            for (int o = 0; o < 3; ++o) {   // one row at a time: the three butterflies unrolled together spill
              const float dd = o == 0 ? dd0 : (o == 1 ? dd1 : dd2);
              float pw[32];
#pragma unroll
              for (int kk = 0; kk < 32; ++kk) pw[kk] = dd * H4[kk];
              const float r = reduce_scatter32(pw, s);
              sw[0] += o == 0 ? r : 0.f; sw[1] += o == 1 ? r : 0.f; sw[2] += o == 2 ? r : 0.f;
            }
          }
        }
        save_rows<32>(gb, sv::K1G_DZ4, dz4, s, h);
        float dz3[32];
        {
          f32x16 acc[2];
          acc_zero<2>(acc);
#ifdef RDRF_HEADS_BWD_F32
          mfma_seg<2, 32>(acc, dz4, lds + pkb::K1W_W4T, lane);
#else
          mfma_seg_b3<2, 32>(acc, dz4, lds + pkb::K1W_W4T, lane);
#endif
          float H3[32];
          load_rows<32>(svb, sv::K1_H3, H3, s, h);
#pragma unroll
          for (int kk = 0; kk < 32; ++kk) dz3[kk] = H3[kk] > 0.f ? acc[kk >> 4][kk & 15] : 0.f;
        }
        save_rows<32>(gb, sv::K1G_DZ3, dz3, s, h);
        f32x16 accX[2];  // d(X0): heads (from rows) + warp layer 3
        {
          float dXh[32];
          load_rows<32>(gb, sv::K1G_DX0, dXh, s, h);
#pragma unroll
          for (int kk = 0; kk < 32; ++kk) accX[kk >> 4][kk & 15] = dXh[kk];
        }
#ifdef RDRF_HEADS_BWD_F32
        mfma_seg<2, 32>(accX, dz3, lds + pkb::K1W_W3T_X0, lane);
#endif
        {
          f32x16 accT[1];
          acc_zero<1>(accT);
#ifdef RDRF_HEADS_BWD_F32
          mfma_seg<1, 32>(accT, dz3, lds + pkb::K1W_W3T_T, lane);
#else
          mfma_seg_b3_pair<2, 1, 32>(accX, accT, dz3, lds + pkb::K1W_W3T_X0, lds + pkb::K1W_W3T_T, lane);
#endif
#pragma unroll
          for (int i = 0; i < 16; ++i) dTacc[i] += accT[0][i];
        }
        // ---- positional encoding backward -> d(xn); plus the identity path xw <- xn
        {
          float X0[32], dX0[32];
          load_rows<32>(svb, sv::K1_X0, X0, s, h);
          acc_copy<2>(dX0, accX);
          float e0 = 0.f, e1 = 0.f, e2 = 0.f;
          x0_bwd(X0, dX0, h, e0, e1, e2);
          e0 += __shfl_xor(e0, 32, 64); e1 += __shfl_xor(e1, 32, 64); e2 += __shfl_xor(e2, 32, 64);
          dn0 += e0 + dw0; dn1 += e1 + dw1; dn2 += e2 + dw2;
        }
        if (act && h == 0 && a.g_xyz) {
          if (FEAT && a.in_norm) {   // compute_*: gradient wrt the NORMALISED input coordinates
            a.g_xyz[(size_t)idx * 3 + 0] += dn0; a.g_xyz[(size_t)idx * 3 + 1] += dn1;
            a.g_xyz[(size_t)idx * 3 + 2] += dn2;
          } else {
            a.g_xyz[(size_t)idx * 3 + 0] += dn0 * a.box.inv[0] + gp0;
            a.g_xyz[(size_t)idx * 3 + 1] += dn1 * a.box.inv[1] + gp1;
            a.g_xyz[(size_t)idx * 3 + 2] += dn2 * a.box.inv[2] + gp2;
          }
        }
        if constexpr (FEAT) {  // time is per point: d(tout) of this sample, no reduction over the tile
          if (act) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a.dtout[(size_t)idx * 32 + elem_of(i, h)] = dTacc[i];
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) dTacc[i] = 0.f;
        }
      }
    }
    if (PHASE == 1 && !FEAT) {  // per-ray d(tout): sum over the samples (lanes of each half)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = dTacc[i];
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if (s == 0) a.dtout[(size_t)n * 32 + elem_of(i, h)] = v;
      }
    }
    if (!FEAT && PHASE == 0 && a.g_rays && a.ray_type != RDRF_RAY_OTHER) {
      g_nrm = wave_sum(g_nrm);
      if (lane < 3) atomicAdd(a.g_rays + (size_t)n * 6 + 3 + lane, g_nrm * (lane == 0 ? vx : (lane == 1 ? vy : vz)));
    }
  }
  if (!FEAT && a.small_dw) {
    // small-layer gradients: sum the workgroup's waves in LDS, then one atomic per entry and workgroup
    __shared__ float red[8][3][65];
    __syncthreads();   // (also: every wave is done with `carr`)
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      red[wave][o][lane] = sw[o];
      const float bs = wave_sum(h == 0 ? sb[o] : 0.f);   // both halves hold the same samples
      if (lane == 0) red[wave][o][64] = bs;
    }
    __syncthreads();
    if (wave == 0) {
      const bool live_d = a.g_sigma != nullptr || a.g_weight != nullptr, live_b = a.g_blending != nullptr;
#pragma unroll
      for (int o = 0; o < 3; ++o) {
        float v = 0.f, bv = 0.f;
        for (int wv = 0; wv < nwaves; ++wv) { v += red[wv][o][lane]; bv += red[wv][o][64]; }
        float* gwt = PHASE == 1 ? gw.l5w + o * 64 : (o == 0 ? gw.dw2 : gw.bw2);
        float* gbs = PHASE == 1 ? gw.l5b + o : (o == 0 ? gw.db2 : gw.bb2);
        const bool on = PHASE == 1 ? true : (o == 0 ? live_d : (o == 1 ? live_b : false));
        if (on) {
          grad_add(gwt + elem_of(s, h), v);
          if (lane == 0) grad_add(gbs, bv);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// time branch backward: [t, PE8(t)] -> 64 -> relu -> 30.  32 rays per 128-thread block, 4 lanes per
// ray (16 hidden neurons each); parameter gradients are reduced inside the block through LDS, then
// one atomic per entry.  (128 rays per block / thread per ray used 32 CUs: 56 us per launch.)
// ------------------------------------------------------------------------------------------------
#define TB_RPB 32
__global__ __launch_bounds__(128) void k_time_branch_bwd(const float* __restrict__ ts, DynW w, int N,
                                                         const float* __restrict__ dtout,
                                                         float* __restrict__ g_l1w,
                                                         float* __restrict__ g_l1b,
                                                         float* __restrict__ g_l2w,
                                                         float* __restrict__ g_l2b) {
  __shared__ float s_tin[TB_RPB][17];
  __shared__ float s_h[TB_RPB][65];
  __shared__ float s_dz1[TB_RPB][65];
  __shared__ float s_dz2[TB_RPB][31];
  const int tid = threadIdx.x;
  const int r = tid >> 2, q = tid & 3;
  const int n = blockIdx.x * TB_RPB + r;
  const bool act = n < N;
  float tin[17];
  const float t = act ? ts[n] : 0.f;
  tin[0] = t;
#pragma unroll
  for (int f = 0; f < 8; ++f) sincosf(ldexpf(t, f), &tin[1 + f], &tin[9 + f]);
  if (q == 0)
    for (int i = 0; i < 17; ++i) s_tin[r][i] = tin[i];
  for (int o = q; o < 30; o += 4) s_dz2[r][o] = act ? dtout[(size_t)n * 32 + o] : 0.f;
  __syncthreads();
  for (int k = 16 * q; k < 16 * q + 16; ++k) {
    float hk = w.l1b[k];
#pragma unroll
    for (int i = 0; i < 17; ++i) hk = fmaf(w.l1w[k * 17 + i], tin[i], hk);
    float dh = 0.f;
    for (int o = 0; o < 30; ++o) dh = fmaf(w.l2w[o * 64 + k], s_dz2[r][o], dh);
    s_h[r][k] = fmaxf(hk, 0.f);
    s_dz1[r][k] = hk > 0.f ? dh : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < 30 * 64; e += 128) {
    const int o = e / 64, k = e - o * 64;
    float a = 0.f;
    for (int rr = 0; rr < TB_RPB; ++rr) a = fmaf(s_dz2[rr][o], s_h[rr][k], a);
    grad_add(g_l2w + e, a);
  }
  for (int e = tid; e < 64 * 17; e += 128) {
    const int k = e / 17, i = e - k * 17;
    float a = 0.f;
    for (int rr = 0; rr < TB_RPB; ++rr) a = fmaf(s_dz1[rr][k], s_tin[rr][i], a);
    grad_add(g_l1w + e, a);
  }
  if (tid < 30) {
    float a = 0.f;
    for (int rr = 0; rr < TB_RPB; ++rr) a += s_dz2[rr][tid];
    grad_add(g_l2b + tid, a);
  }
  if (tid < 64) {
    float a = 0.f;
    for (int rr = 0; rr < TB_RPB; ++rr) a += s_dz1[rr][tid];
    grad_add(g_l1b + tid, a);
  }
}

// ------------------------------------------------------------------------------------------------
// scene flow backward-data
// ------------------------------------------------------------------------------------------------
RDRF_D void sf_x_bwd(const float (&X)[20], const float (&dX)[20], int h, float& d0, float& d1,
                     float& d2) {
#pragma unroll
  for (int o = 0; o < 5; ++o) {
    if (o == 0 && h == 0) {
      d0 += dX[0]; d1 += dX[1]; d2 += dX[2];
    } else {
      const int k = 2 * o + h - 1;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int pr = 2 * k + p;
        if (pr < 12) {
          const int d = pr >> 2, f = pr & 3;
          const float dq = ldexpf(dX[o * 4 + 2 * p] * X[o * 4 + 2 * p + 1] -
                                  dX[o * 4 + 2 * p + 1] * X[o * 4 + 2 * p], f);
          d0 += d == 0 ? dq : 0.f; d1 += d == 1 ? dq : 0.f; d2 += d == 2 ? dq : 0.f;   // selects, see x0_bwd
        }
      }
    }
  }
}

__global__ __launch_bounds__(64 * RDRF_MAXW) void k_scene_flow_bwd(int N, int S, Box box,
                                                        const float* __restrict__ pkg,
                                                        const float* __restrict__ act_rows,
                                                        float* __restrict__ grows,
                                                        const float* __restrict__ g_f,
                                                        const float* __restrict__ g_b,
                                                        float* __restrict__ g_sfb6,
                                                        float* __restrict__ g_pts) {
  __shared__ __attribute__((aligned(16))) float lds[pkb::SF_SIZE];
  lds_fill(lds, pkg + pkb::REG_SF, pkb::SF_SIZE);
  const int lane = threadIdx.x & 63, h = lane >> 5, s = lane & 31;
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int total = N * S;
  const int ntiles = (total + 31) >> 5;
  for (int tile = blockIdx.x * nwaves + wave; tile < ntiles; tile += gridDim.x * nwaves) {
    const int li = tile * 32 + s;
    const bool act = li < total;
    const int idx = act ? li : 0;
    const float* svb = act_rows + (size_t)tile * sv::SF_ROWS * 32;
    float* gb = grows + (size_t)tile * sv::SFG_ROWS * 32;
    float dz6[6];
#pragma unroll
    for (int o = 0; o < 6; ++o) {
      const float* gsrc = o < 3 ? g_f : g_b;
      dz6[o] = (act && gsrc) ? gsrc[(size_t)idx * 3 + (o % 3)] : 0.f;
      if (h == 0) gb[(size_t)(sv::SFG_DZ6 + o) * 32 + s] = dz6[o];
    }
    float dz[32], Hh[32];
    load_rows<32>(svb, sv::SF_H4, Hh, s, h);
    small_layer_bwd<32, 6>(dz, Hh, lds + pkb::SF_W6, h, dz6);
    save_rows<32>(gb, sv::SFG_DZ4, dz, s, h);
    f32x16 acc[2];
    acc_zero<2>(acc);
    mfma_seg<2, 32>(acc, dz, lds + pkb::SF_W4T, lane);
    load_rows<32>(svb, sv::SF_H2, Hh, s, h);
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) dz[kk] = Hh[kk] > 0.f ? acc[kk >> 4][kk & 15] : 0.f;
    save_rows<32>(gb, sv::SFG_DZ2, dz, s, h);
    acc_zero<2>(acc);
    mfma_seg<2, 32>(acc, dz, lds + pkb::SF_W2T, lane);
    load_rows<32>(svb, sv::SF_H0, Hh, s, h);
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) dz[kk] = Hh[kk] > 0.f ? acc[kk >> 4][kk & 15] : 0.f;
    save_rows<32>(gb, sv::SFG_DZ0, dz, s, h);
    if (g_pts) {
      acc_zero<2>(acc);
      mfma_seg<2, 32>(acc, dz, lds + pkb::SF_W0T, lane);
      float X[20], dX[20];
      load_rows<20>(svb, sv::SF_X, X, s, h);
#pragma unroll
      for (int kk = 0; kk < 20; ++kk) dX[kk] = acc[kk >> 4][kk & 15];
      float d0 = 0.f, d1 = 0.f, d2 = 0.f;
      sf_x_bwd(X, dX, h, d0, d1, d2);
      d0 += __shfl_xor(d0, 32, 64); d1 += __shfl_xor(d1, 32, 64); d2 += __shfl_xor(d2, 32, 64);
      if (act && h == 0) {
        g_pts[(size_t)idx * 3 + 0] += d0 * box.inv[0];
        g_pts[(size_t)idx * 3 + 1] += d1 * box.inv[1];
        g_pts[(size_t)idx * 3 + 2] += d2 * box.inv[2];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// generic dW kernel: dW[out][col(e)] += sum_tiles sum_samples dz[out][s] * in[e][s]
// ------------------------------------------------------------------------------------------------
struct DwJob {
  const float* A;   // dz rows: tile t, row r at A + (t*A_stride + A_row0 + r)*32
  int A_stride, A_row0, nbo, out_dim, out_row0;
  const float* B;   // input rows
  int B_stride;
  int nblk;         // number of 32-row input blocks
  int blk_row0[8], blk_seg[8], blk_e0[8];
  int in_dim, ld;
  float* dW;
  float* db;        // bias gradient (nullable), indexed like the out rows
  const int* count; // device sample count (compacted phases) or nullptr
  int ntiles;
};
#define RDRF_MAX_DW_JOBS 12
struct DwJobs {
  DwJob j[RDRF_MAX_DW_JOBS];
  int n;
};

static void dw_add(DwJobs& D, const float* A, int A_stride, int A_row0, int nbo, int out_dim,
                   int out_row0, const float* B, int B_stride, int in_dim, int ld, float* dW,
                   float* db, const int* count, int ntiles) {
  DwJob& j = D.j[D.n];
  memset(&j, 0, sizeof(j));
  j.A = A; j.A_stride = A_stride; j.A_row0 = A_row0; j.nbo = nbo; j.out_dim = out_dim;
  j.out_row0 = out_row0; j.B = B; j.B_stride = B_stride; j.nblk = 0; j.in_dim = in_dim; j.ld = ld;
  j.dW = dW; j.db = db; j.count = count; j.ntiles = ntiles;
  D.n++;
}
static void dw_blk(DwJobs& D, int row0, int seg, int e0) {
  DwJob& j = D.j[D.n - 1];
  j.blk_row0[j.nblk] = row0; j.blk_seg[j.nblk] = seg; j.blk_e0[j.nblk] = e0;
  j.nblk++;
}
// ------------------------------------------------------------------------------------------------
// cooperative dW kernel (k_dw2).  k_dw above gives every (out-block, in-group) item its own wave and its
// own copy of the rows it needs: a density-phase tile is requested 55 blocks at a time for 27 unique ones,
// and each wave's 20 KB stage is written and read once per item.  Here ONE 8-wave workgroup owns a tile:
// it stages every unique 32-row block of the tile in LDS once (27 x 4 KB = 108 KB for the density phase,
// 30 for the appearance phases), then its waves form all (dz block) x (input block) products of all the
// jobs from that stage -- 5 or 6 products per wave, accumulators resident for the whole launch.  The
// global loads of the next tile are issued right after the stage is written (into registers: 14 x 16 B
// per thread) and land while the MFMAs of the current tile run.  Per tile and CU: 108 KB of HBM traffic
// (was ~220 KB of L2 traffic), 640 MFMAs spread over the four SIMDs.
// ------------------------------------------------------------------------------------------------
#define DW2_MAX_BLK 30
#define DW2_MAX_PROD 4
#define DW2_WAVES 12
#define DW2_MAX_SEG 2
struct Dw2Prod {     // 32-bit fields: scalar loads (see blk_meta)
  int a, b;          // staged block indices of the dz block / the input block
  int job, bo, k;    // write-out: job, out-block of the job, in-block index of the job
  int bias;          // this product also accumulates the bias gradient of its out-block
};
struct Dw2Plan {
  const float* A;    // dz rows: tile t, row r at A + (t*A_stride + r)*32
  const float* B;    // activation rows
  int A_stride, B_stride;
  int nblk;
  int blk_meta[DW2_MAX_BLK];        // (src << 16) | row0, src 0 = A, 1 = B.  32-bit on purpose: the block index is
                                    // wave-uniform, so these are SCALAR loads (lgkmcnt); byte / short fields
                                    // compile to vector loads whose vmcnt(0) waits drain the data loads in flight
  // the staged blocks, sorted by (source, first row), form at most DW2_MAX_SEG runs of consecutive rows: a run
  // is ONE contiguous byte range per tile, so a slot's address needs no per-block metadata (see k_dw2)
  int nseg;
  int seg_blk0[DW2_MAX_SEG];          // first staged block of the run
  int seg_src[DW2_MAX_SEG];           // 0 = A, 1 = B
  int seg_row0[DW2_MAX_SEG];          // first row of the run inside a tile
  int nprod[DW2_WAVES];
  Dw2Prod prod[DW2_WAVES][DW2_MAX_PROD];
  const int* count;
  int ntiles;
  DwJob job[RDRF_MAX_DW_JOBS];
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifdef RDRF_TOOLS   // k_dw2 is the A/B partner of k_dw3 (RDRF_DW3=0) in the tools build; the product launches k_dw3 only
__global__ __launch_bounds__(64 * DW2_WAVES) void k_dw2(Dw2Plan P) {
  extern __shared__ __attribute__((aligned(16))) f32x4 dw2_stage[];   // nblk x 256 float4, XOR-swizzled per block
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, li = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: indexes the plan in the kernel arguments
  const int ntiles = P.count ? ((*P.count + 31) >> 5) : P.ntiles;
  const int nf4 = P.nblk * 256;
  constexpr int NT = 64 * DW2_WAVES, NPF = (DW2_MAX_BLK * 256 + NT - 1) / NT;
  // this thread's slots of the tile image: float4 index i*NT + tid -> (block, row, 16-byte chunk)
  f32x4 pf[NPF];
  // Slot i of this thread is float4 number idx = i*NT + tid of the tile image; its staged block idx >> 8 is
  // wave-uniform.  The blocks are sorted so that they form <= DW2_MAX_SEG runs of consecutive rows, each run one
  // contiguous range of a tile: the address is (run base of this tile, scalar) + idx*16 bytes, selected with
  // scalar compares.  (Looking the block up in the plan inside the loop cost two dependent scalar loads + waits
  // in front of each of the 10 global loads: 2-4 thousand cycles per tile in which the wave issued no MFMA.)
  static_assert(DW2_MAX_SEG == 2, "segment select below");
  // every plan of this path is [dz rows 0..nA) | activation rows 0..nB): two runs (checked on the host)
  const int sb1 = P.nseg > 1 ? P.seg_blk0[1] : 1 << 20;
  const int s1 = P.nseg > 1 ? 1 : 0;
  const float* seg0a = (P.seg_src[0] ? P.B : P.A) + (size_t)P.seg_row0[0] * 32;
  const float* seg0b = (P.seg_src[s1] ? P.B : P.A) + (size_t)P.seg_row0[s1] * 32 - (size_t)P.seg_blk0[s1] * 1024;
  const size_t st0 = (size_t)(P.seg_src[0] ? P.B_stride : P.A_stride) * 32,
               st1 = (size_t)(P.seg_src[s1] ? P.B_stride : P.A_stride) * 32;
  static_assert(NT == 768, "a slot is three 256-float4 blocks: slot i of wave w holds block 3*i + (w >> 2)");
  const int wgrp = wave >> 2;   // scalar
  const int voffb = tid * 16;   // the only per-thread part of an address
  // Buffer loads: descriptor (scalar, rebuilt per tile from the run's tile base) + voffb + a per-slot scalar
  // offset.  No vector address temporaries: with 64-bit flat addresses the compiler built them in the prefetch
  // registers themselves and its waitcnt pass then put `s_waitcnt vmcnt(0)` in front of every load of the batch.
  auto gload = [&](int t) {
    const float* g0 = seg0a + (size_t)t * st0;
    const float* g1 = seg0b + (size_t)t * st1;
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)g0, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)g1, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int blk = 3 * i + wgrp;          // wave-uniform: uniform branches
      if (blk < P.nblk) {
#ifndef RDRF_ABL_DW_NOLOAD
        u32x4 v;
        if (blk < sb1) v = __builtin_amdgcn_raw_buffer_load_b128(r0, voffb, i * (NT * 16), 0);
        else v = __builtin_amdgcn_raw_buffer_load_b128(r1, voffb, i * (NT * 16), 0);
        pf[i] = __builtin_bit_cast(f32x4, v);
#else
        pf[i] = f32x4{(float)t, 1.f, 2.f, (float)i};
#endif
      }
    }
  };
  int rpos[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) rpos[q] = li * 8 + ((4 * h + q) ^ ((li >> 1) & 7));
  const int np = P.nprod[wave];
  int pa[DW2_MAX_PROD], pb[DW2_MAX_PROD];   // staged block offsets of each product, bias flag in bit 30 of pa (scalars)
#pragma unroll
  for (int p = 0; p < DW2_MAX_PROD; ++p) {
    pa[p] = __builtin_amdgcn_readfirstlane(P.prod[wave][p].a * 256 | (P.prod[wave][p].bias << 30));
    pb[p] = __builtin_amdgcn_readfirstlane(P.prod[wave][p].b * 256);
  }
  f32x16 acc[DW2_MAX_PROD];
  float bsum[DW2_MAX_PROD];
#pragma unroll
  for (int p = 0; p < DW2_MAX_PROD; ++p) {
    bsum[p] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
  }
  int t = blockIdx.x;
  if (t < ntiles) gload(t);
  while (t < ntiles) {
    __syncthreads();   // every wave is done reading the previous tile
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      if (3 * i + wgrp < P.nblk) {   // uniform
        const int idx = i * NT + tid;
        const int blk = idx >> 8, w = idx & 255, row = w >> 3;
        dw2_stage[blk * 256 + row * 8 + ((w & 7) ^ ((row >> 1) & 7))] = pf[i];
      }
    }
    __syncthreads();
    const int tn = t + gridDim.x;
    if (tn < ntiles) gload(tn);   // lands while the MFMAs below run
#ifndef RDRF_ABL_DW_NOMFMA
#pragma unroll
    for (int p = 0; p < DW2_MAX_PROD; ++p) {
      if (p < np) {
        int oa = pa[p] & 0xffffff, ob = pb[p];
        // (opaque to the optimiser: with loop-invariant offsets it hoists all 32 LDS read addresses of the
        // four products out of the tile loop and spills them -- reloads whose vmcnt(0) drain the prefetch)
        asm volatile("" : "+s"(oa), "+s"(ob));
        const f32x4* sa = dw2_stage + oa;
        const f32x4* sb = dw2_stage + ob;
        f32x4 av4[4], bv4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { av4[q] = sa[rpos[q]]; bv4[q] = sb[rpos[q]]; }
        if (pa[p] >> 30) {
#pragma unroll
          for (int q = 0; q < 4; ++q) bsum[p] += av4[q].x + av4[q].y + av4[q].z + av4[q].w;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av4[q].x, bv4[q].x, acc[p], 0, 0, 0);
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av4[q].y, bv4[q].y, acc[p], 0, 0, 0);
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av4[q].z, bv4[q].z, acc[p], 0, 0, 0);
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av4[q].w, bv4[q].w, acc[p], 0, 0, 0);
        }
      }
    }
#else
    if (np > 0) acc[0][0] += dw2_stage[tid & 255].x;   // keep the stage alive
#endif
    t = tn;
  }
  // write-out: C row i = (rr&3) + 8*(rr>>2) + 4*h (out neuron), column = li (input element)
#ifdef RDRF_ABL_DW_NOFLUSH
  if (ntiles >= 0) return;
#endif
#pragma unroll
  for (int p = 0; p < DW2_MAX_PROD; ++p) {
    if (p < np) {
      const Dw2Prod pr = P.prod[wave][p];
      const DwJob& J = P.job[pr.job];
      const int col = seg_imap(J.blk_seg[pr.k], J.blk_e0[pr.k] + li, J.in_dim);
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int orow = pr.bo * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h - J.out_row0;
        if (col >= 0 && orow >= 0 && orow < J.out_dim) grad_add(J.dW + (size_t)orow * J.ld + col, acc[p][rr]);
      }
      if (pr.bias && J.db != nullptr) {
        const float b = bsum[p] + __shfl_xor(bsum[p], 32, 64);
        const int orow = pr.bo * 32 + li - J.out_row0;
        if (h == 0 && orow >= 0 && orow < J.out_dim) grad_add(J.db + orow, b);
      }
    }
  }
}

#endif   // RDRF_TOOLS (k_dw2)

// ------------------------------------------------------------------------------------------------
// k_dw3 (round 6): the same plan as k_dw2 on HALF stages with LDS-DMA.  k_dw2 stages a whole tile
// (32 samples x all rows, 108-120 KB of the 160 KB LDS) through registers: barrier, ten ds_write_b128 per thread, barrier, the
// products -- the matrix pipes wait during the write pass and both barriers (3 x (64 MFMA + 4 VALU cycles) account for 0.66
// of its wave cycles), and nothing can be double-buffered.  Here a stage is 16 samples x all rows (64 bytes of every row,
// 54-60 KB), there are two of them, and `buffer_load_dwordx4 ... lds` moves a half stage straight from HBM into the buffer
// that is not being read: step s = barrier; issue the DMA of step s + 1; the products of step s (8 MFMAs each, K = 16
// samples).  One barrier per half stage, no staging registers, no LDS write instructions.
// Measured (profiles/r06_ab_dw3_*.txt): dw launches 2.02 -> 1.91 ms/step at stage 0, 4.40 -> 4.10 at the final stage.  Ablations:
// MFMAs + barriers alone 1.05 ms (dw_dyn, stage 0), DMA + barriers alone 1.29 -- the memory side binds, and NOT through its
// latency: a ring of 80 block slots that keeps two half stages in flight (counted vmcnt, raw s_barrier) left the DMA-only time
// at 1.30 ms and made the kernel slower (1.56).  Nor through the 64-byte half rows: the same bytes fetched as full 128-byte
// rows (RDRF_ABL_DW_FULLROW, wrong arithmetic, timing only: profiles/r06_ab_dw3_fullrow_timing.txt) take 1.18 instead of 1.25 ms
// DMA-only and 1.43 instead of 1.47 ms in the whole kernel.  What is left is the per-step bubble of a 12-wave workgroup:
// vmcnt(0) -> barrier -> ~5 DMA issues per wave, during which this CU has nothing in flight (~0.5 of each 2.7 us step).
//   LDS image of a block (32 rows x 16 samples = 2 KB): float4 position p = row * 4 + (chunk ^ ((row >> 2) & 3)); a DMA
//   instruction fills 1 KB in lane order (base + lane * 16 -- the hardware's layout), so the swizzle sits on the SOURCE
//   address of lane l (row = 16 sub + (l >> 2), chunk = (l & 3) ^ ((row >> 2) & 3)) and on the read (cdna guide, rule 21);
//   the 16 lanes of a ds_read_b128 group then cover all 64 banks.
// ------------------------------------------------------------------------------------------------
#ifndef RDRF_DW_AUX
#define RDRF_DW_AUX 0   // cache policy of the row DMA (2 = nt on gfx950: the rows are read once)
#endif
__global__ __launch_bounds__(64 * DW2_WAVES) void k_dw3(Dw2Plan P) {
  extern __shared__ __attribute__((aligned(16))) f32x4 dw3_stage[];   // 2 buffers x nblk x 128 float4
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, li = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = P.count ? ((*P.count + 31) >> 5) : P.ntiles;
  static_assert(DW2_MAX_SEG == 2 && (DW2_WAVES & 1) == 0, "segment select below; a wave's pieces share their parity");
  const int sb1 = P.nseg > 1 ? P.seg_blk0[1] : 1 << 20;
  const int s1 = P.nseg > 1 ? 1 : 0;
  const float* seg0a = (P.seg_src[0] ? P.B : P.A) + (size_t)P.seg_row0[0] * 32;
  const float* seg0b = (P.seg_src[s1] ? P.B : P.A) + (size_t)P.seg_row0[s1] * 32 - (size_t)P.seg_blk0[s1] * 1024;
  const size_t st0 = (size_t)(P.seg_src[0] ? P.B_stride : P.A_stride) * 32,
               st1 = (size_t)(P.seg_src[s1] ? P.B_stride : P.A_stride) * 32;
  const int npieces = 2 * P.nblk;   // 1 KB pieces (16 rows x 64 B) of a half stage
  constexpr int NPW = (2 * DW2_MAX_BLK + DW2_WAVES - 1) / DW2_WAVES;   // pieces per wave: 5
  // the lane's source offset inside a piece (bytes): row (l >> 2) of the piece's 16, chunk swizzled by the row
  // (a wave's pieces wave, wave + 12, ... all have the parity of the wave: one offset register)
  const int prow_ = 16 * (wave & 1) + (lane >> 2);
  const unsigned voff = (unsigned)(prow_ * 128 + (((lane & 3) ^ ((prow_ >> 2) & 3)) << 4));
  const int hbuf = P.nblk * 128;   // float4 per buffer
  auto dma = [&](int t, int hf, int buf) {
#ifdef RDRF_DW_REV   // A/B: walk the tiles newest first (the rows the backward-data kernels touched last)
    t = ntiles - 1 - t;
#endif
    const float* g0 = seg0a + (size_t)t * st0;
    const float* g1 = seg0b + (size_t)t * st1;
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)g0, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)g1, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int pc = wave + DW2_WAVES * i;   // wave-uniform
      if (pc < npieces) {
#ifndef RDRF_ABL_DW_NOLOAD
        const int blk = pc >> 1;
        __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(dw3_stage + buf * hbuf + pc * 64);
#ifdef RDRF_ABL_DW_FULLROW   // timing experiment (tools): the same bytes as FULL 128-byte rows (rows 16 hf .. 16 hf + 15 of the block,
        const unsigned soff = (unsigned)(blk * 4096 + hf * 2048 + (pc & 1) * 1024);   // 1 KB contiguous per piece); WRONG results
        const unsigned vo = (unsigned)(lane * 16);
        if (blk < sb1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, dst, 16, vo, soff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, dst, 16, vo, soff, 0, 0);
#else
        const unsigned soff = (unsigned)(blk * 4096 + hf * 64);
        if (blk < sb1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, dst, 16, voff, soff, 0, RDRF_DW_AUX);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, dst, 16, voff, soff, 0, RDRF_DW_AUX);
#endif
#endif
      }
    }
  };
  int rpos[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) rpos[q] = li * 4 + ((2 * h + q) ^ ((li >> 2) & 3));
  const int np = P.nprod[wave];
  int pa[DW2_MAX_PROD], pb[DW2_MAX_PROD];   // staged block offsets (float4) of each product, bias flag in bit 30 of pa (scalars)
#pragma unroll
  for (int p = 0; p < DW2_MAX_PROD; ++p) {
    pa[p] = __builtin_amdgcn_readfirstlane(P.prod[wave][p].a * 128 | (P.prod[wave][p].bias << 30));
    pb[p] = __builtin_amdgcn_readfirstlane(P.prod[wave][p].b * 128);
  }
  f32x16 acc[DW2_MAX_PROD];
  float bsum[DW2_MAX_PROD];
#pragma unroll
  for (int p = 0; p < DW2_MAX_PROD; ++p) {
    bsum[p] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
  }
  int t = blockIdx.x, hf = 0, buf = 0;
  if (t < ntiles) dma(t, 0, 0);
  while (t < ntiles) {
    // this wave's DMA pieces of the current half stage have landed; behind the barrier everybody's have, and everybody is
    // done reading the other buffer (the products of the previous step)
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0) (expcnt / lgkmcnt untouched)
    __syncthreads();
    int tn = t, hn = hf ^ 1;
    if (hf) tn = t + gridDim.x;
    if (tn < ntiles) dma(tn, hn, buf ^ 1);   // lands while the MFMAs below run
    __builtin_amdgcn_sched_barrier(0);       // (issued BEFORE the products: hipcc is free to sink it below them otherwise)
#ifndef RDRF_ABL_DW_NOMFMA
    const f32x4* stage = dw3_stage + buf * hbuf;
#pragma unroll
    for (int p = 0; p < DW2_MAX_PROD; ++p) {
      if (p < np) {
        int oa = pa[p] & 0xffffff, ob = pb[p];
        asm volatile("" : "+s"(oa), "+s"(ob));   // (as in k_dw2: keeps the read addresses out of the loop-invariant hoist)
        const f32x4* sa = stage + oa;
        const f32x4* sb = stage + ob;
        f32x4 av4[2], bv4[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { av4[q] = sa[rpos[q]]; bv4[q] = sb[rpos[q]]; }
        if (pa[p] >> 30) {
#pragma unroll
          for (int q = 0; q < 2; ++q) bsum[p] += av4[q].x + av4[q].y + av4[q].z + av4[q].w;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av4[q].x, bv4[q].x, acc[p], 0, 0, 0);
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av4[q].y, bv4[q].y, acc[p], 0, 0, 0);
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av4[q].z, bv4[q].z, acc[p], 0, 0, 0);
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av4[q].w, bv4[q].w, acc[p], 0, 0, 0);
        }
      }
    }
#else
    if (np > 0) acc[0][0] += dw3_stage[buf * hbuf + (tid & 127)].x;
#endif
    t = tn; hf = hn; buf ^= 1;
  }
#ifdef RDRF_ABL_DW_NOFLUSH
  if (ntiles >= 0) return;
#endif
#pragma unroll
  for (int p = 0; p < DW2_MAX_PROD; ++p) {
    if (p < np) {
      const Dw2Prod pr = P.prod[wave][p];
      const DwJob& J = P.job[pr.job];
      const int col = seg_imap(J.blk_seg[pr.k], J.blk_e0[pr.k] + li, J.in_dim);
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int orow = pr.bo * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h - J.out_row0;
        if (col >= 0 && orow >= 0 && orow < J.out_dim) grad_add(J.dW + (size_t)orow * J.ld + col, acc[p][rr]);
      }
      if (pr.bias && J.db != nullptr) {
        const float b = bsum[p] + __shfl_xor(bsum[p], 32, 64);
        const int orow = pr.bo * 32 + li - J.out_row0;
        if (h == 0 && orow >= 0 && orow < J.out_dim) grad_add(J.db + orow, b);
      }
    }
  }
}

// one plan per group of jobs that walk the same rows (same dz array, same activation array, same tile
// space); a group whose products or blocks exceed one plan is cut into several launches
static int dw_launch(DwJobs& D, hipStream_t stream, const char* name) {
  bool done[RDRF_MAX_DW_JOBS] = {false};
  for (int g0 = 0; g0 < D.n; ++g0) {
    if (done[g0]) continue;
    const DwJob& R = D.j[g0];
    Dw2Plan P;
    auto reset = [&]() {
      memset(&P, 0, sizeof(P));
      P.A = R.A; P.B = R.B; P.A_stride = R.A_stride; P.B_stride = R.B_stride;
      P.count = R.count; P.ntiles = R.ntiles;
      for (int i = 0; i < D.n; ++i) P.job[i] = D.j[i];
    };
    auto find_blk = [&](int src, int row) {
      for (int i = 0; i < P.nblk; ++i)
        if (P.blk_meta[i] == ((src << 16) | row)) return i;
      return -1;
    };
    auto flush = [&]() -> int {
      int tot = 0;
      for (int w = 0; w < DW2_WAVES; ++w) tot += P.nprod[w];
      if (tot == 0) return 0;
      {  // one contiguous run per source: a pruned head leaves a hole in the row ranges (its dz / activation rows are
         // not part of any product); the hole's blocks are staged unused so that the two-run addressing holds
        for (int src = 0; src < 2; ++src) {
          int lo = 1 << 30, hi = -1;
          for (int i = 0; i < P.nblk; ++i)
            if ((P.blk_meta[i] >> 16) == src) { const int r = P.blk_meta[i] & 0xffff; lo = r < lo ? r : lo; hi = r > hi ? r : hi; }
          for (int r = lo; r < hi; r += 32)
            if (find_blk(src, r) < 0) {
              RDRF_CHECK(P.nblk < DW2_MAX_BLK, -2, "dw: %d staged blocks are not enough to bridge the row ranges of this plan", DW2_MAX_BLK);
              P.blk_meta[P.nblk++] = (src << 16) | r;
            }
        }
      }
      {  // stage order = (source, first row) order; runs of consecutive rows become segments
        int order[DW2_MAX_BLK], rank[DW2_MAX_BLK], meta[DW2_MAX_BLK];
        for (int i = 0; i < P.nblk; ++i) order[i] = i;
        std::sort(order, order + P.nblk, [&](int x, int y) { return P.blk_meta[x] < P.blk_meta[y]; });
        for (int i = 0; i < P.nblk; ++i) { rank[order[i]] = i; meta[i] = P.blk_meta[order[i]]; }
        for (int i = 0; i < P.nblk; ++i) P.blk_meta[i] = meta[i];
        for (int w = 0; w < DW2_WAVES; ++w)
          for (int k = 0; k < P.nprod[w]; ++k) { P.prod[w][k].a = rank[P.prod[w][k].a]; P.prod[w][k].b = rank[P.prod[w][k].b]; }
        P.nseg = 0;
        for (int i = 0; i < P.nblk; ++i) {
          if (i == 0 || (meta[i] >> 16) != (meta[i - 1] >> 16) || (meta[i] & 0xffff) != (meta[i - 1] & 0xffff) + 32) {
            RDRF_CHECK(P.nseg < DW2_MAX_SEG, -2, "dw: the staged rows form more than %d contiguous runs", DW2_MAX_SEG);
            P.seg_blk0[P.nseg] = i; P.seg_src[P.nseg] = meta[i] >> 16; P.seg_row0[P.nseg] = meta[i] & 0xffff;
            ++P.nseg;
          }
        }
      }
      if (RDRF_ENV("RDRF_DW_DEBUG")) {
        fprintf(stderr, "dw plan %s: nblk %d products %d segs", name, P.nblk, tot);
        for (int g = 0; g < P.nseg; ++g) {
          const int end = g + 1 < P.nseg ? P.seg_blk0[g + 1] : P.nblk;
          fprintf(stderr, " [%c rows %d..%d]", P.seg_src[g] ? 'B' : 'A', P.seg_row0[g], P.seg_row0[g] + 32 * (end - P.seg_blk0[g]) - 1);
        }
        fprintf(stderr, " strides A %d B %d\n", P.A_stride, P.B_stride);
      }
      const size_t lds = (size_t)P.nblk * 4096;   // two half stages of nblk x 2 KB (k_dw2: one whole stage of nblk x 4 KB)
      int grid = 256;
      if (P.count == nullptr && P.ntiles < grid) grid = P.ntiles < 1 ? 1 : P.ntiles;
#ifdef RDRF_TOOLS
      static const int dw3 = RDRF_ENV("RDRF_DW3") ? atoi(RDRF_ENV("RDRF_DW3")) : 1;   // 0: k_dw2
      if (lds > 48 * 1024) {
        if (dw3) RDRF_HIP(hipFuncSetAttribute((const void*)k_dw3, hipFuncAttributeMaxDynamicSharedMemorySize, DW2_MAX_BLK * 4096));
        else RDRF_HIP(hipFuncSetAttribute((const void*)k_dw2, hipFuncAttributeMaxDynamicSharedMemorySize, DW2_MAX_BLK * 4096));
      }
      rdrf_prof_begin(name, stream);
      if (dw3) hipLaunchKernelGGL(k_dw3, dim3(grid), dim3(64 * DW2_WAVES), lds, stream, P);
      else hipLaunchKernelGGL(k_dw2, dim3(grid), dim3(64 * DW2_WAVES), lds, stream, P);
#else
      if (lds > 48 * 1024)
        RDRF_HIP(hipFuncSetAttribute((const void*)k_dw3, hipFuncAttributeMaxDynamicSharedMemorySize, DW2_MAX_BLK * 4096));
      rdrf_prof_begin(name, stream);
      hipLaunchKernelGGL(k_dw3, dim3(grid), dim3(64 * DW2_WAVES), lds, stream, P);
#endif
      rdrf_prof_end(name, stream);
      RDRF_HIP(hipGetLastError());
      return 0;
    };
    reset();
    int nprods = 0;
    for (int ji = g0; ji < D.n; ++ji) {
      const DwJob& J = D.j[ji];
      if (done[ji] || J.A != R.A || J.B != R.B || J.A_stride != R.A_stride || J.B_stride != R.B_stride ||
          J.count != R.count || J.ntiles != R.ntiles)
        continue;
      done[ji] = true;
      for (int bo = 0; bo < J.nbo; ++bo)
        for (int k = 0; k < J.nblk; ++k) {
          int need = (find_blk(0, J.A_row0 + 32 * bo) < 0) + (find_blk(1, J.blk_row0[k]) < 0);
          if (nprods == DW2_WAVES * DW2_MAX_PROD || P.nblk + need > DW2_MAX_BLK) {
            int rc = flush();
            if (rc) return rc;
            reset();
            nprods = 0;
          }
          int a = find_blk(0, J.A_row0 + 32 * bo);
          if (a < 0) { a = P.nblk++; P.blk_meta[a] = J.A_row0 + 32 * bo; }
          int b = find_blk(1, J.blk_row0[k]);
          if (b < 0) { b = P.nblk++; P.blk_meta[b] = (1 << 16) | J.blk_row0[k]; }
          const int w = nprods % DW2_WAVES;   // round robin: consecutive products of a job share their dz block
          Dw2Prod& pr = P.prod[w][P.nprod[w]++];
          pr.a = a; pr.b = b; pr.job = ji; pr.bo = bo; pr.k = k;
          pr.bias = (k == 0 && J.db != nullptr) ? 1 : 0;
          ++nprods;
        }
    }
    int rc = flush();
    if (rc) return rc;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
void fill_static_w(StaticW& w, const RdrfStaticParams* P);
void fill_dyn_w(DynW& w, const RdrfDynamicParams* P);

static void dyn_pack_jobs_bwd(PackJobs& J, const RdrfDynamicParams* P) {
  using namespace pkb;
  J.n = 0;
  const int kh = REG_K1H, kw = REG_K1W, k3 = REG_K3, sf = REG_SF;
  pack_add(J, P->l5w, 64, 3, 64, SEG_IDENT, 1, 3, 32, kw + K1W_W5);
#ifdef RDRF_HEADS_BWD_F32
  const int wm = 2;
#else
  const int wm = 8;   // bf16 x 3 transposed fragments
#endif
  pack_add(J, P->l4w, 64, 64, 64, SEG_IDENT, wm, 2, 32, kw + K1W_W4T);
  pack_add(J, P->l3w, 93, 64, 93, SEG_WARP3_X0, wm, 2, 32, kw + K1W_W3T_X0);
  pack_add(J, P->l3w, 93, 64, 93, SEG_WARP3_T, wm, 1, 32, kw + K1W_W3T_T);
  pack_add(J, P->dw2, 64, 1, 64, SEG_IDENT, 1, 1, 32, kh + K1H_DEN2);
  pack_add(J, P->bw2, 64, 1, 64, SEG_IDENT, 1, 1, 32, kh + K1H_BLE2);
#ifdef RDRF_HEADS_BWD_F32
  const int hm = 2;
#else
  const int hm = 8;   // bf16 x 3 transposed fragments
#endif
  pack_add(J, P->dw1, 152, 64, 72, SEG_IDENT, hm, 3, 32, kh + K1H_DEN1T_F);
  pack_add(J, P->dw1, 152, 64, 152, SEG_DEN1_X0, hm, 2, 32, kh + K1H_DEN1T_X0);
  pack_add(J, P->bw1, 152, 64, 72, SEG_IDENT, hm, 3, 32, kh + K1H_BLE1T_F);
  pack_add(J, P->bw1, 152, 64, 152, SEG_DEN1_X0, hm, 2, 32, kh + K1H_BLE1T_X0);
  pack_add(J, P->rwv, 131, 3, 128, SEG_IDENT, 1, 3, 64, k3 + K3_RGBV);
#ifdef RDRF_APP_F32   // A/B builds: the appearance phases' backward-data products on the fp32 matrix pipe
  pack_add(J, P->rw2, 128, 128, 128, SEG_IDENT, 2, 4, 64, k3 + K3_RGB2T);
  pack_add(J, P->rw1, 107, 128, 107, SEG_RGB1_F, 2, 1, 64, k3 + K3_RGB1T_F);
  pack_add(J, P->rw1, 107, 128, 107, SEG_RGB1_X0, 2, 2, 64, k3 + K3_RGB1T_X0);
  pack_add(J, P->basis, 216, 27, 216, SEG_IDENT, 2, 7, 16, k3 + K3_BASIST);
#else                 // bf16 x 3 with split storage (same image offsets and sizes)
  pack_add_b3s_t(J, P->rw2, 128, 128, 128, SEG_IDENT, 4, 64, k3 + K3_RGB2T, REG_K3_LO + K3_LO_RGB2T);
  pack_add_b3s_t(J, P->rw1, 107, 128, 107, SEG_RGB1_F, 1, 64, k3 + K3_RGB1T_F, REG_K3_LO + K3_LO_RGB1T);
  pack_add_b3s_t(J, P->rw1, 107, 128, 107, SEG_RGB1_X0, 2, 64, k3 + K3_RGB1T_X0, REG_K3_LO + K3_LO_RGB1T + 64 * 32);
  pack_add_b3s_t(J, P->basis, 216, 27, 216, SEG_IDENT, 7, 16, k3 + K3_BASIST, REG_K3_LO + K3_LO_BASIST);
#endif
  pack_add(J, P->sfw[3], 64, 6, 64, SEG_IDENT, 1, 6, 32, sf + SF_W6);
  pack_add(J, P->sfw[2], 64, 64, 64, SEG_IDENT, 2, 2, 32, sf + SF_W4T);
  pack_add(J, P->sfw[1], 64, 64, 64, SEG_IDENT, 2, 2, 32, sf + SF_W2T);
  pack_add(J, P->sfw[0], 36, 64, 36, SEG_SF_X, 2, 2, 32, sf + SF_W0T);
}

static void static_pack_jobs_bwd(PackJobs& J, const RdrfStaticParams* P, int head) {
  using namespace pkb;
  J.n = 0;
  const bool fea = head == RDRF_HEAD_MLP_FEA;
  const int in1 = fea ? 138 : 135;
  pack_add(J, P->w3, fea ? 128 : 131, 3, 128, SEG_IDENT, 1, 3, 64, REG_S3 + S3_W3);
#ifdef RDRF_APP_F32
  pack_add(J, P->w2, 128, 128, 128, SEG_IDENT, 2, 4, 64, REG_S3 + S3_W2T);
  pack_add(J, P->w1, in1, 128, in1, fea ? SEG_STAT1_F_FEA : SEG_STAT1_F_TE, 2, 1, 64, REG_S3 + S3_W1T_F);
  pack_add(J, P->w1, in1, 128, in1, fea ? SEG_STAT1_P_FEA : SEG_STAT1_P_TE, 2, 4, 64, REG_S3 + S3_W1T_P);
  pack_add(J, P->basis, 72, 27, 72, SEG_IDENT, 2, 3, 16, REG_S3 + S3_BASIST);
#else
  pack_add_b3s_t(J, P->w2, 128, 128, 128, SEG_IDENT, 4, 64, REG_S3 + S3_W2T, REG_S3_LO + S3_LO_W2T);
  pack_add_b3s_t(J, P->w1, in1, 128, in1, fea ? SEG_STAT1_F_FEA : SEG_STAT1_F_TE, 1, 64, REG_S3 + S3_W1T_F, REG_S3_LO + S3_LO_W1T);
  pack_add_b3s_t(J, P->w1, in1, 128, in1, fea ? SEG_STAT1_P_FEA : SEG_STAT1_P_TE, 4, 64, REG_S3 + S3_W1T_P, REG_S3_LO + S3_LO_W1T + 64 * 32);
  pack_add_b3s_t(J, P->basis, 72, 27, 72, SEG_IDENT, 3, 16, REG_S3 + S3_BASIST, REG_S3_LO + S3_LO_BASIST);
#endif
}

struct Geo {
  int grid, block;
};
static Geo geo_for_units(long units) {
  Geo g;
  const int ncu = 256;
  int waves = (int)((units + ncu - 1) / ncu);
  waves = waves < 1 ? 1 : (waves > RDRF_MAXW ? RDRF_MAXW : waves);
  g.block = waves * 64;
  long blocks = (units + waves - 1) / waves;
  g.grid = (int)(blocks < 1 ? 1 : (blocks > ncu ? ncu : blocks));
  return g;
}

#define PACK_AREA_FLOATS (1 << 20)
static_assert(pkb::REG_K3_LO + pkb::K3_LO_SIZE <= PACK_AREA_FLOATS && pkb::REG_S3_LO + pkb::S3_LO_SIZE <= PACK_AREA_FLOATS,
              "the backward images and their streamed lo pieces fit the pack area");

static void fill_bwd_common(BwdArgs& a, const RdrfFieldCfg* cfg, const float* rays, const float* ts,
                            const float* xyz, const float* z, const uint8_t* valid, int N, int S) {
  memset(&a, 0, sizeof(a));
  a.rays = rays; a.ts = ts; a.xyz = xyz; a.z = z; a.valid = valid; a.N = N; a.S = S;
  a.box = make_box(cfg);
  a.distance_scale = cfg->distance_scale; a.weight_thres = cfg->weight_thres;
  a.density_shift = cfg->density_shift; a.act = cfg->act; a.ray_type = cfg->ray_type;
  a.static_head = cfg->static_head;
  static const int dynq = RDRF_ENV("RDRF_DYNQ") ? atoi(RDRF_ENV("RDRF_DYNQ")) : 1;   // 0: static tile stride (tools build)
  a.dynq = dynq;
}

// forward calls (either field, scene flow): pack area + counter + tout + xw + list -- what an inference-only caller needs
extern "C" size_t rdrf_forward_workspace_bytes(int N, int S) {
  const size_t ns = (size_t)N * S;
  return (size_t)PACK_AREA_FLOATS * 4 + 256 + (size_t)N * 32 * 4 + ns * 3 * 4 + ns * 4 + (1 << 12);
}

extern "C" size_t rdrf_workspace_bytes(int N, int S) {
  size_t ns = (size_t)N * S, t1 = (size_t)N * ((S + 31) / 32), t3 = (ns + 31) / 32;
  size_t fwd = rdrf_forward_workspace_bytes(N, S);
  // backward: pack area + dz rows of both phases + coordinate-gradient buffers + d(tout)
  size_t bwd = (size_t)PACK_AREA_FLOATS * 4 + t1 * sv::K1G_ROWS * 32 * 4 + t3 * sv::K3G_ROWS * 32 * 4 +
               ns * 3 * 4 * 2 + (size_t)N * 32 * 4 + (1 << 14);
  // sorted scatter: sample-major d(feature) records, keys in / out, sorted positions, counters, radix-sort scratch
  bwd += ns * DFS_FLOATS * 4 + ns * DFA_FLOATS * 4 + 3 * ns * 4 * 3 + 1024 + rdrf_sort_temp_bytes((unsigned)(3 * ns), 32) + (1 << 12);
  size_t sf = (size_t)PACK_AREA_FLOATS * 4 + t3 * sv::SFG_ROWS * 32 * 4 + (1 << 12);
  size_t m = fwd > bwd ? fwd : bwd;
  return m > sf ? m : sf;
}

struct BwdWs {
  float* dfs;            // sorted scatter (dynamic field)
  float* dfa;            // sorted appearance scatter: [N*S] records of DFA_FLOATS (capacity; count <= N*S entries are used)
  unsigned *keys_in, *keys_out, *order;
  int* counts;
  void* sort_tmp;
  size_t sort_tmp_bytes;
  float* pk;
  float* gf;       // static field: d(density feature) per sample, [N][ceil(S/32)*32]
  float* grows1;
  float* grows3;
  float* dxw;
  float* dxn;
  float* dtout;
};
static int carve_bwd(BwdWs& b, void* ws, size_t ws_bytes, int N, int S, int dynamic) {
  WsCarver c(ws, ws_bytes);
  size_t ns = (size_t)N * S, t1 = (size_t)N * ((S + 31) / 32), t3 = (ns + 31) / 32;
  b.pk = c.take<float>(PACK_AREA_FLOATS);
  b.grows3 = c.take<float>(t3 * sv::K3G_ROWS * 32);
  b.grows1 = dynamic ? c.take<float>(t1 * sv::K1G_ROWS * 32) : nullptr;
  b.dxw = dynamic ? c.take<float>(ns * 3) : nullptr;
  b.dxn = dynamic ? c.take<float>(ns * 3) : nullptr;
  b.dtout = dynamic ? c.take<float>((size_t)N * 32) : nullptr;
  b.gf = dynamic ? nullptr : c.take<float>(t1 * 32);
  b.dfs = nullptr;
  b.dfa = nullptr;
  if (dynamic) {
    b.dfs = c.take<float>(ns * DFS_FLOATS);
    b.dfa = c.take<float>(ns * DFA_FLOATS);
    b.keys_in = c.take<unsigned>(3 * ns);
    b.keys_out = c.take<unsigned>(3 * ns);
    b.order = c.take<unsigned>(3 * ns);
    b.counts = c.take<int>(64);
    b.sort_tmp_bytes = rdrf_sort_temp_bytes((unsigned)(3 * ns), 32);
    b.sort_tmp = c.take<char>(b.sort_tmp_bytes);
  }
  RDRF_CHECK(c.ok(), -3, "backward workspace too small: need %zu have %zu", c.off, ws_bytes);
  return 0;
}

static void fill_scatter_common(ScatterArgs& sa, const BwdArgs& a) {
  memset(&sa, 0, sizeof(sa));
  sa.xyz = a.xyz; sa.box = a.box; sa.valid = a.valid; sa.N = a.N; sa.S = a.S;
}
static int lines_floats_host(const RdrfVM& vm) {
  int n = 0;
  for (int i = 0; i < 3; ++i) n += vm.L[i] * (vm.C[i] + 4);   // lds_stride(C) = C + 4
  return n;
}
// LDS line accumulators when both factor sets' lines fit SC_LINES_MAX_BYTES; launches with that much
// dynamic LDS (the attribute call is needed above 64 KB and is idempotent)
#ifndef RDRF_SC_THREADS2_DEFAULT
#define RDRF_SC_THREADS2_DEFAULT 256
#endif
template <typename K>
static int launch_scatter(const char* name, K kern, ScatterArgs& sa, long ntiles, hipStream_t stream) {
  const long n0 = lines_floats_host(sa.vm[0]), n1 = sa.nsets > 1 ? lines_floats_host(sa.vm[1]) : 0;
  // element type of the accumulators: doubles when they fit (ds_add_f64 retires 11 x the updates of ds_add_f32, see
  // LdsLines), also at the price of one launch per factor set; floats for lines too long for that; else global atomics
  static const int f64_env = RDRF_ENV("RDRF_LDS_F64") ? atoi(RDRF_ENV("RDRF_LDS_F64")) : 1;   // 0: floats (tools build, A/B)
  int esz = 0;
  bool split = false;
  for (int e = f64_env ? 8 : 4; e >= 4 && !esz; e -= 4) {
    if (e * (n0 + n1) <= SC_LINES_MAX_BYTES) esz = e;
    else if (sa.nsets == 2 && e * n0 <= SC_LINES_MAX_BYTES && e * n1 <= SC_LINES_MAX_BYTES) { esz = e; split = true; }
  }
#ifdef RDRF_DETERMINISTIC
  esz = 0;   // the LDS accumulators add in wave-arrival order: the line gradients go straight to the fixed-point shadow
  split = false;
#endif
  if (split) {
    // the two factor sets do not fit the LDS together but each does alone: one launch per set
    ScatterArgs s0 = sa, s1 = sa;
    s0.nsets = 1;
    s1.nsets = 1;
    s1.vm[0] = sa.vm[1]; s1.gvm[0] = sa.gvm[1]; s1.row0[0] = sa.row0[1];
    s1.dxw_accumulate = 1;
    int rc = launch_scatter(name, kern, s0, ntiles, stream);
    return rc ? rc : launch_scatter(name, kern, s1, ntiles, stream);
  }
  sa.lds_bytes = (int)(esz * (n0 + n1));
  sa.lds_f64 = esz == 8;
  if (sa.lds_bytes > 48 * 1024)
    RDRF_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SC_LINES_MAX_BYTES));
  // workgroups per CU that the accumulators leave room for (160 KB of LDS): one big workgroup, or two / three of 256
  // threads; the grid never exceeds what is resident at once (the tile loop is a static stride: a workgroup that starts
  // after the others have finished would run its share alone)
  const int per_cu = sa.lds_bytes > 80 * 1024 ? 1 : (sa.lds_bytes > 53 * 1024 ? 2 : 3);
  static const int thr2_env = RDRF_ENV("RDRF_SC_THREADS2") ? atoi(RDRF_ENV("RDRF_SC_THREADS2")) : RDRF_SC_THREADS2_DEFAULT;
  const int threads = per_cu == 1 ? 512 : (per_cu == 2 ? thr2_env : 256);   // two workgroups per CU: 512 threads each = 4 waves per SIMD
  const int wpb = threads / 64;
  long g = (ntiles + wpb - 1) / wpb;
  static const long cap_env = RDRF_ENV("RDRF_SC_CAP") ? atol(RDRF_ENV("RDRF_SC_CAP")) : 0;   // experiments (tools build)
  const long cap = per_cu == 1 ? 256 : (cap_env > 0 ? cap_env : 256 * per_cu);
  g = g < 1 ? 1 : (g > cap ? cap : g);
  rdrf_prof_begin(name, stream);
  hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(threads), (size_t)sa.lds_bytes, stream, sa);
  rdrf_prof_end(name, stream);
  RDRF_HIP(hipGetLastError());
  return 0;
}

// rdrf_set_scatter_mode(RDRF_SCATTER_AUTO (default) | _RAY | _SORTED): how the density / blending gradients of the dynamic
// field's ray path reach the factor planes -- the ray-tile kernel (k_scatter), or samples grouped by plane cell first
// (k_scatter_sorted: ~10x fewer memory-side atomic requests).  Measured on MI355X at the Balloon1 stage-0 shape (DESIGN.md
// 9): the grouping is a fixed cost per launch, the saving grows with the batch.  Kept selectable: sorted is the only path
// whose request count does not depend on the warp field's smoothness, and the parity tests run both at every size.
static int g_scatter_mode = RDRF_SCATTER_AUTO;
extern "C" int rdrf_set_scatter_mode(int mode) {
  RDRF_CHECK(mode == RDRF_SCATTER_AUTO || mode == RDRF_SCATTER_RAY || mode == RDRF_SCATTER_SORTED || mode == RDRF_SCATTER_SORTED_PLAIN, -1,
             "rdrf_set_scatter_mode: mode must be RDRF_SCATTER_AUTO, _RAY, _SORTED or _SORTED_PLAIN");
  g_scatter_mode = mode;
  return 0;
}
static int scatter_mode(size_t ns, hipStream_t stream) {   // 0 ray, 1 sorted
  int m = g_scatter_mode;
  if (const char* e = RDRF_ENV("RDRF_SCATTER")) m = !strcmp(e, "sorted") ? RDRF_SCATTER_SORTED : (!strcmp(e, "ray") ? RDRF_SCATTER_RAY : m);
  // auto: with the hand-written radix sort (72 us for 1.4 M keys; rocPRIM took 150) the sorted path wins from the
  // Balloon1 stage-0 pass (4096 x 115 = 471 k samples: 12.57 vs 12.72 ms/step, interleaved A/B on one box) upwards, and
  // sends a tenth of the atomic requests; below ~300 k samples the fixed cost of its seven extra launches dominates
  // A launch sequence that is being CAPTURED into a HIP graph pays no per-launch host cost at replay, so the grouping wins at
  // every size there: at the S = 13 stages of Nvidia_no_poses.txt / DAVIS.txt (53 k - 106 k samples per pass into a 17 x 19 x 11 /
  // 16^3 grid: every plane fits one LDS window) the captured iteration runs 5.44 -> 5.11 / 9.31 -> 8.93 ms
  // (profiles/r06_graph_scatter_ab.txt).
  if (m == RDRF_SCATTER_AUTO) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) return 1;
    return ns >= (size_t)300000 ? 1 : 0;
  }
  return (m == RDRF_SCATTER_SORTED || m == RDRF_SCATTER_SORTED_PLAIN) ? 1 : 0;
}

template <int PLANE, int C0Q, int C1Q>
static int launch_scatter_sorted(SortedScatterArgs& sa, long max_samples, hipStream_t stream) {
  // pass PLANE accumulates line PLANE only: L x (C + 4) elements per factor set, doubles when they fit (see LdsLines)
  const long n = ((sa.set_mask & 1) ? (long)sa.vm[0].L[PLANE] * (sa.vm[0].C[PLANE] + 4) : 0) +
                 ((sa.set_mask & 2) ? (long)sa.vm[1].L[PLANE] * (sa.vm[1].C[PLANE] + 4) : 0);
  static const int f64_env = RDRF_ENV("RDRF_LDS_F64") ? atoi(RDRF_ENV("RDRF_LDS_F64")) : 1;   // 0: floats (tools build, A/B)
  const int esz = (f64_env && 8 * n <= SC_LINES_MAX_BYTES) ? 8 : (4 * n <= SC_LINES_MAX_BYTES ? 4 : 0);
  sa.lds_bytes = (int)(esz * n);
  sa.lds_f64 = esz == 8;
  static const int direct_env = RDRF_ENV("RDRF_LINE_DIRECT") ? atoi(RDRF_ENV("RDRF_LINE_DIRECT")) : RDRF_LINE_DIRECT_DEFAULT;
  sa.line_direct = sa.lds_f64 && direct_env;
#ifdef RDRF_DETERMINISTIC
  sa.lds_bytes = 0;
#endif
  if (sa.lds_bytes > 48 * 1024)
    RDRF_HIP(hipFuncSetAttribute((const void*)k_scatter_sorted<PLANE, C0Q, C1Q>, hipFuncAttributeMaxDynamicSharedMemorySize, SC_LINES_MAX_BYTES));
  // 512 threads x 2 workgroups per CU = 4 waves per SIMD (the kernels need <= 116 VGPRs): 12.16 -> 11.98 ms / step against
  // 256 x 3 (profiles/r05_ab_sorted_occupancy.txt); one workgroup per CU when the accumulators take more than half the LDS
  int per_cu = sa.lds_bytes > 80 * 1024 ? 1 : 2;
  int threads = 512;
  static const int thr_env = RDRF_ENV("RDRF_SS_THREADS") ? atoi(RDRF_ENV("RDRF_SS_THREADS")) : 0;   // experiments (tools build)
  static const int pcu_env = RDRF_ENV("RDRF_SS_PER_CU") ? atoi(RDRF_ENV("RDRF_SS_PER_CU")) : 0;
  if (thr_env > 0 && pcu_env > 0 && (long)pcu_env * sa.lds_bytes <= 160 * 1024) { threads = thr_env; per_cu = pcu_env; }
  const int wpb = threads / 64;
  const long ntiles = (max_samples + (PLANE == 0 ? 15 : 31)) / (PLANE == 0 ? 16 : 32);
  long g = (ntiles + wpb - 1) / wpb;
  const long cap = 256 * per_cu;
  g = g < 1 ? 1 : (g > cap ? cap : g);
  static const char* names[3] = {"scatter_sorted_xy", "scatter_sorted_xz", "scatter_sorted_yz"};
  rdrf_prof_begin(names[PLANE], stream);
  hipLaunchKernelGGL((k_scatter_sorted<PLANE, C0Q, C1Q>), dim3((unsigned)g), dim3(threads), (size_t)sa.lds_bytes, stream, sa);
  rdrf_prof_end(names[PLANE], stream);
  RDRF_HIP(hipGetLastError());
  return 0;
}

// the tiled form of pass PLANE (k_scatter_tiled) when its line accumulators and plane windows fit the LDS as doubles;
// returns 1 if it was launched, 0 if the caller should take k_scatter_sorted, < 0 on error
#ifndef RDRF_SS_TILED_DEFAULT
#define RDRF_SS_TILED_DEFAULT 1
#endif
template <int PLANE, int C0Q, int C1Q>
static int launch_scatter_tiled(SortedScatterArgs& sa, const unsigned* keys_sorted, int kb, long max_samples, hipStream_t stream) {
#ifdef RDRF_DETERMINISTIC
  return 0;   // LDS sums form in wave-arrival order
#endif
  static const int tiled_env = RDRF_ENV("RDRF_SS_TILED") ? atoi(RDRF_ENV("RDRF_SS_TILED")) : RDRF_SS_TILED_DEFAULT;
  if (!tiled_env || g_scatter_mode == RDRF_SCATTER_SORTED_PLAIN) return 0;
  static const int tw_env = RDRF_ENV("RDRF_SS_TW") ? atoi(RDRF_ENV("RDRF_SS_TW")) : 0;
  static const int steps_env = RDRF_ENV("RDRF_SS_STEPS") ? atoi(RDRF_ENV("RDRF_SS_STEPS")) : 0;
  constexpr int CT = PLANE == 0 ? 4 * C0Q : 4 * C1Q;
  int tw = tw_env > 0 ? tw_env : (CT > 16 ? 16 : 32);   // 48-component texels: narrower windows keep two workgroups per CU
  tw = (tw < 8 ? 8 : (tw > 128 ? 128 : tw)) & ~3;
  const long n0 = (sa.set_mask & 1) ? (long)sa.vm[0].L[PLANE] * (sa.vm[0].C[PLANE] + 4) : 0;
  const long n1 = (sa.set_mask & 2) ? (long)sa.vm[1].L[PLANE] * (sa.vm[1].C[PLANE] + 4) : 0;
  const long n = n0 + n1;
  sa.tile_sets = sa.set_mask == 3 ? 2 : 1;
  const long bytes = 8L * (n + (long)sa.tile_sets * tile_texels(tw) * CT);
  // only where two 512-thread workgroups per CU still fit (at the final grids the z lines of two factor sets alone take
  // 70 KB: one workgroup per CU lost more than the windows gained, 3.46 -> 3.97 ms) and for the 16- / 4-component texels
  // of the density / blending sets (the 48-component appearance windows gained nothing: 0.71 -> 0.71 ms)
  static const long max_env = RDRF_ENV("RDRF_SS_TILED_MAXB") ? atol(RDRF_ENV("RDRF_SS_TILED_MAXB")) : 80 * 1024;
  if (C0Q > 4 && tiled_env < 2) return 0;   // (tiled_env >= 2 exists in the tools build only: RDRF_ENV is null in the product)
  // two workgroups per CU: each needs its dynamic bytes + the kernel's static LDS (s_geo: 96 B), rounded to the 512-byte
  // allocation granule (at exactly 80 KB of dynamic LDS the second workgroup did not fit)
  const long per_wg = ((bytes + 96 + 511) / 512) * 512;
  if (per_wg > max_env) {
    // both factor sets together do not leave room for two workgroups per CU, each alone does (final grids: 35 KB of z
    // line + 20 KB of windows per set).  One tiled launch per set was measured and LOSES to the untiled two-set kernel
    // (final stage, scatter_dyn_density 3.08 -> 3.24 ms / step, profiles/r05_ab_tiled_scatter.txt: every entry's taps and
    // run structure are formed twice), so it stays an experiment switch of the tools build and the caller falls back
    static const int split_env = RDRF_ENV("RDRF_SS_SPLIT") ? atoi(RDRF_ENV("RDRF_SS_SPLIT")) : 0;
    const long one = 8L * ((n0 > n1 ? n0 : n1) + (long)tile_texels(tw) * CT);
    if (sa.set_mask != 3 || one > max_env || !split_env) return 0;
    SortedScatterArgs s0 = sa, s1 = sa;
    s0.set_mask = 1; s1.set_mask = 2;
    int rc = launch_scatter_tiled<PLANE, C0Q, C1Q>(s0, keys_sorted, kb, max_samples, stream);
    if (rc != 1) return rc;
    rc = launch_scatter_tiled<PLANE, C0Q, C1Q>(s1, keys_sorted, kb, max_samples, stream);
    return rc == 0 ? -5 : rc;   // (cannot happen: the same size test passed for the larger of the two)
  }
  sa.lds_bytes = (int)bytes; sa.lds_f64 = 1;
  static const int direct_env = RDRF_ENV("RDRF_LINE_DIRECT") ? atoi(RDRF_ENV("RDRF_LINE_DIRECT")) : RDRF_LINE_DIRECT_DEFAULT;
  sa.line_direct = direct_env;
  sa.keys = keys_sorted; sa.kb = kb; sa.tw = tw;
  sa.slice_steps = steps_env > 0 ? steps_env : 16;   // 2 steps for each of the 8 waves (16: 1.51, 32: 1.55, 64: 1.68 ms / step)
  sa.Wk = sa.vm[0].W[PLANE] + 3;
  if (sa.lds_bytes > 48 * 1024)
    RDRF_HIP(hipFuncSetAttribute((const void*)k_scatter_tiled<PLANE, C0Q, C1Q>, hipFuncAttributeMaxDynamicSharedMemorySize, SC_LINES_MAX_BYTES));
  const int per_cu = per_wg > 80 * 1024 ? 1 : 2;   // 512-thread workgroups, 4 waves per SIMD (__launch_bounds__(512, 4): HIP counts waves per SIMD)
  const long nslices = (max_samples + (long)sa.slice_steps * (PLANE == 0 ? 16 : 32) - 1) / ((long)sa.slice_steps * (PLANE == 0 ? 16 : 32));
  long g = nslices < 256L * per_cu ? nslices : 256L * per_cu;
  g = g < 1 ? 1 : g;
  static const char* names[3] = {"scatter_tiled_xy", "scatter_tiled_xz", "scatter_tiled_yz"};
  rdrf_prof_begin(names[PLANE], stream);
  hipLaunchKernelGGL((k_scatter_tiled<PLANE, C0Q, C1Q>), dim3((unsigned)g), dim3(512), (size_t)sa.lds_bytes, stream, sa);
  rdrf_prof_end(names[PLANE], stream);
  RDRF_HIP(hipGetLastError());
  return 1;
}

// keys of the entries (all samples, or the compacted list), stable sort by (plane | cell), live counts per plane
static int sorted_scatter_prepare(SortKeyArgs& ka, const RdrfVM& vm, const BwdArgs& a, const BwdWs& b, hipStream_t stream) {
  const size_t ns = (size_t)a.N * a.S;
  ka.xw = a.sp.xw; ka.valid = a.valid; ka.N = a.N; ka.S = a.S;
  long maxcells = 0;
  for (int p = 0; p < 3; ++p) {
    ka.W[p] = vm.W[p]; ka.H[p] = vm.H[p];
    const long c = (long)(ka.W[p] + 3) * (ka.H[p] + 3);
    maxcells = c > maxcells ? c : maxcells;
  }
  int kb = 1;
  while (((1L << kb) - 1) < maxcells) ++kb;
  RDRF_CHECK(kb <= 29, -1, "sorted scatter: plane too large for 32-bit keys");
  ka.kb = kb; ka.keys = b.keys_in; ka.counts = b.counts;
  rdrf_prof_begin("scatter_sort", stream);
  {
    long g = ((long)ns + 255) / 256;
    g = g > 2048 ? 2048 : g;
    rdrf_prof_begin("sort_keys", stream);
    hipLaunchKernelGGL(k_sort_keys, dim3((unsigned)g), dim3(256), 0, stream, ka);
    rdrf_prof_end("sort_keys", stream);
  }
  // compact: the sort covers the 3 x *count live-list entries only (launches sized for 3 N S; the appearance list holds
  // 35-60 % of the samples)
  int rc = rdrf_sort_positions(b.keys_in, b.keys_out, b.order, (unsigned)(3 * ns), kb + 2, b.sort_tmp, b.sort_tmp_bytes, stream,
                               ka.compact ? ka.count : nullptr, 3u);
  if (rc) return rc;
  hipLaunchKernelGGL(k_sort_counts, dim3(1), dim3(64), 0, stream, (const unsigned*)b.keys_out, (int)ns, kb, b.counts,
                     ka.compact ? ka.count : (const int*)nullptr);
  rdrf_prof_end("scatter_sort", stream);
  RDRF_HIP(hipGetLastError());
  return 0;
}

// appearance scatter of the dynamic field's ray path with the compacted samples grouped by plane cell: the same
// per-quad device functions as the ray-tile kernel (k_scatter<12, 3, 27>), whose 18 M memory-side atomic requests per
// launch (runs of 1-2 samples along a ray) were the largest share of the step's requests
static int scatter_dyn_app_sorted(const BwdArgs& a, const BwdWs& b, const RdrfDynamicParams* P, const RdrfDynamicParams* G,
                                  hipStream_t stream) {
  const size_t ns = (size_t)a.N * a.S;
  SortKeyArgs ka;
  memset(&ka, 0, sizeof(ka));
  ka.list = a.sp.list; ka.count = &a.sp.hdr->count;
#ifdef RDRF_TOOLS   // (the windows experiment below addresses the sorted keys with the host-side stride N S)
  static const int compact_env = RDRF_ENV("RDRF_SORT_COMPACT") ? atoi(RDRF_ENV("RDRF_SORT_COMPACT")) : 1;
  ka.compact = compact_env && !(RDRF_ENV("RDRF_SS_TILED") && atoi(RDRF_ENV("RDRF_SS_TILED")) >= 2);
#else
  ka.compact = 1;
#endif
  int rc = sorted_scatter_prepare(ka, P->app, a, b, stream);
  if (rc) return rc;
  SortedScatterArgs sa;
  memset(&sa, 0, sizeof(sa));
  sa.vm[0] = P->app; sa.gvm[0] = G->app;
  sa.set_mask = 1; sa.dfs = b.dfa; sa.rec_floats = DFA_FLOATS; sa.list = a.sp.list; sa.xw = a.sp.xw; sa.dxw = b.dxw;
  for (int p = 0; p < 3; ++p) {
    sa.order = b.order + (size_t)p * ns; sa.count = b.counts + p; sa.base = (unsigned)(p * ns);
    if (ka.compact) { sa.order = b.order; sa.seg = ka.count; }   // device-side segment starts (k_scatter_sorted)
    const unsigned* ks = b.keys_out + (size_t)p * ns;
#ifdef RDRF_TOOLS   // the 48-component windows gained nothing (0.713 -> 0.695 ms, profiles/r05_ab_tiled_scatter.txt): an experiment of the
    rc = p == 0 ? launch_scatter_tiled<0, 12, 3>(sa, ks, ka.kb, (long)ns, stream)          // tools build (RDRF_SS_TILED=2), not
                : (p == 1 ? launch_scatter_tiled<1, 12, 3>(sa, ks, ka.kb, (long)ns, stream)   // compiled into the product
                          : launch_scatter_tiled<2, 12, 3>(sa, ks, ka.kb, (long)ns, stream));
    if (rc < 0) return rc;
    if (rc == 1) continue;
#else
    (void)ks;
#endif
    rc = p == 0 ? launch_scatter_sorted<0, 12, 3>(sa, (long)ns, stream)
                : (p == 1 ? launch_scatter_sorted<1, 12, 3>(sa, (long)ns, stream) : launch_scatter_sorted<2, 12, 3>(sa, (long)ns, stream));
    if (rc) return rc;
  }
  return 0;
}

// density / blending scatter of the dynamic field's ray path, samples grouped by plane cell (see k_scatter_sorted)
static int scatter_dyn_density_sorted(const BwdArgs& a, const BwdWs& b, const RdrfDynamicParams* P, const RdrfDynamicParams* G,
                                      int set_mask, hipStream_t stream) {
  const size_t ns = (size_t)a.N * a.S;
  for (int p = 0; p < 3; ++p)
    RDRF_CHECK(P->blending.W[p] == P->density.W[p] && P->blending.H[p] == P->density.H[p], -1,
               "sorted scatter: density and blending planes differ in size");
  SortKeyArgs ka;
  memset(&ka, 0, sizeof(ka));
  ka.grows1 = b.grows1;
  int rc = sorted_scatter_prepare(ka, P->density, a, b, stream);
  if (rc) return rc;
  SortedScatterArgs sa;
  memset(&sa, 0, sizeof(sa));
  sa.vm[0] = P->density; sa.gvm[0] = G->density; sa.vm[1] = P->blending; sa.gvm[1] = G->blending;
  sa.set_mask = set_mask; sa.dfs = b.dfs; sa.rec_floats = DFS_FLOATS; sa.xw = a.sp.xw; sa.dxw = b.dxw;
  for (int p = 0; p < 3; ++p) {
    sa.order = b.order + (size_t)p * ns; sa.count = b.counts + p; sa.base = (unsigned)(p * ns);
    const unsigned* ks = b.keys_out + (size_t)p * ns;
    rc = p == 0 ? launch_scatter_tiled<0, 4, 1>(sa, ks, ka.kb, (long)ns, stream)
                : (p == 1 ? launch_scatter_tiled<1, 4, 1>(sa, ks, ka.kb, (long)ns, stream)
                          : launch_scatter_tiled<2, 4, 1>(sa, ks, ka.kb, (long)ns, stream));
    if (rc < 0) return rc;
    if (rc == 1) continue;
    rc = p == 0 ? launch_scatter_sorted<0, 4, 1>(sa, (long)ns, stream)
                : (p == 1 ? launch_scatter_sorted<1, 4, 1>(sa, (long)ns, stream) : launch_scatter_sorted<2, 4, 1>(sa, (long)ns, stream));
    if (rc) return rc;
  }
  return 0;
}

// dW jobs of the dynamic field's density phase (warp MLP, density / blending heads)
static void add_density_phase_dw(DwJobs& D, const float* grows1, const float* act1, const RdrfDynamicParams* G,
                                 int T1, bool live_d = true, bool live_b = true, bool small_in_kernel = false) {
  // layer3: [X0 | tout]
  dw_add(D, grows1, sv::K1G_ROWS, sv::K1G_DZ3, 2, 64, 0, act1, sv::K1_ROWS, 93, 93, G->l3w, G->l3b, nullptr, T1);
  dw_blk(D, sv::K1_X0, SEG_WARP3_X0, 0);
  dw_blk(D, sv::K1_X0 + 32, SEG_WARP3_X0, 32);
  dw_blk(D, sv::K1_T, SEG_WARP3_T, 0);
  dw_add(D, grows1, sv::K1G_ROWS, sv::K1G_DZ4, 2, 64, 0, act1, sv::K1_ROWS, 64, 64, G->l4w, G->l4b, nullptr, T1);
  dw_blk(D, sv::K1_H3, SEG_IDENT, 0);
  dw_blk(D, sv::K1_H3 + 32, SEG_IDENT, 32);
  // small layers share one dz block: rows 0..2 -> layer5, row 3 -> density_layer2, row 4 -> blending_layer2.
  // On the ray path k_dyn_density_bwd forms these gradients itself (reduce_scatter32): as MFMA products they were 6 of
  // the 40 per tile, 27 of 32 rows empty, and the 12 waves of k_dw2 take 34 products in 3 rounds instead of 4
  if (!small_in_kernel) {
  dw_add(D, grows1, sv::K1G_ROWS, sv::K1G_SM, 1, 3, 0, act1, sv::K1_ROWS, 64, 64, G->l5w, G->l5b, nullptr, T1);
  dw_blk(D, sv::K1_H4, SEG_IDENT, 0);
  dw_blk(D, sv::K1_H4 + 32, SEG_IDENT, 32);
  if (live_d) {
    dw_add(D, grows1, sv::K1G_ROWS, sv::K1G_SM, 1, 1, 3, act1, sv::K1_ROWS, 64, 64, G->dw2, G->db2, nullptr, T1);
    dw_blk(D, sv::K1_HD, SEG_IDENT, 0);
    dw_blk(D, sv::K1_HD + 32, SEG_IDENT, 32);
  }
  if (live_b) {
    dw_add(D, grows1, sv::K1G_ROWS, sv::K1G_SM, 1, 1, 4, act1, sv::K1_ROWS, 64, 64, G->bw2, G->bb2, nullptr, T1);
    dw_blk(D, sv::K1_HB, SEG_IDENT, 0);
    dw_blk(D, sv::K1_HB + 32, SEG_IDENT, 32);
  }
  }
  for (int head = 0; head < 2; ++head) {
    if (!(head ? live_b : live_d)) continue;   // dead head: its dz rows were not written (k_dyn_density_bwd<0>)
    dw_add(D, grows1, sv::K1G_ROWS, head ? sv::K1G_DZB : sv::K1G_DZD, 2, 64, 0, act1, sv::K1_ROWS, 152, 152,
           head ? G->bw1 : G->dw1, head ? G->bb1 : G->db1, nullptr, T1);
    const int f0 = head ? sv::K1_FB : sv::K1_FD;
    dw_blk(D, f0, SEG_IDENT72, 0);
    dw_blk(D, f0 + 32, SEG_IDENT72, 32);
    dw_blk(D, f0 + 64, SEG_IDENT72, 64);
    dw_blk(D, sv::K1_X0, SEG_DEN1_X0, 0);
    dw_blk(D, sv::K1_X0 + 32, SEG_DEN1_X0, 32);
    dw_blk(D, sv::K1_X1, SEG_DEN1_X1, 0);
  }
}

extern "C" int rdrf_static_bwd(const RdrfStaticParams* P, const RdrfFieldCfg* cfg, const float* rays,
                               const float* ts, const float* xyz, const float* z,
                               const uint8_t* valid, int N, int S, const float* g_rgb,
                               const float* g_sigma, const float* g_weight, const float* g_dists,
                               const RdrfStaticParams* G, float* g_xyz, float* g_z, float* g_rays,
                               void* saved, size_t saved_bytes, void* ws, size_t ws_bytes,
                               rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(P && cfg && G && saved && N > 0 && S > 0 && S <= 4096, -1, "static_bwd: bad arguments (S <= 4096)");
  RDRF_CHECK((size_t)N * S * 3 < (size_t)INT32_MAX, -1, "static_bwd: N * S * 3 must stay below 2^31 (32-bit sample indices)");
  // z_vals never depends on a trainable quantity in the reference (linspace + jitter)
  BwdArgs a;
  fill_bwd_common(a, cfg, rays, ts, xyz, z, valid, N, S);
  a.g_rgb = g_rgb; a.g_sigma = g_sigma; a.g_weight = g_weight; a.g_xyz = g_xyz;
  a.g_rays = g_rays; a.g_dists = g_dists; a.g_z = g_z;
  RDRF_CHECK(carve_saved(a.sp, saved, saved_bytes, 0, N, S), -3, "static_bwd: saved buffer too small");
  BwdWs b;
  int rc = carve_bwd(b, ws, ws_bytes, N, S, 0);
  if (rc) return rc;
  a.pk = b.pk; a.grows3 = b.grows3;
  StaticW w;
  fill_static_w(w, P);
  StaticG gw;
  gw.density = G->density; gw.app = G->app; gw.b3 = G->b3; gw.w3 = G->w3;
  if (P->packed_bwd != nullptr) a.pk = P->packed_bwd;   // caller-packed image (rdrf_static_pack)
  else {
    PackJobs J;
    static_pack_jobs_bwd(J, P, cfg->static_head);
    rc = pack_launch(J, b.pk, stream);
    if (rc) return rc;
  }
  const size_t t3 = ((size_t)N * S + 31) / 32;
  if (g_rgb != nullptr) {
    const Geo g = geo_for_units((long)t3);
    if (cfg->static_head == RDRF_HEAD_MLP_FEA)
      RDRF_LAUNCH("static_app_bwd", (k_static_app_bwd<RDRF_HEAD_MLP_FEA, false>), dim3(g.grid), dim3(g.block),
                  stream, a, w, gw);
    else
      RDRF_LAUNCH("static_app_bwd", (k_static_app_bwd<RDRF_HEAD_MLP_FEA_TIMEEMBEDDING, false>), dim3(g.grid),
                  dim3(g.block), stream, a, w, gw);
    {
      ScatterArgs sa;
      fill_scatter_common(sa, a);
      sa.vm[0] = P->app; sa.gvm[0] = G->app; sa.nsets = 1;
      sa.rows = b.grows3; sa.stride = sv::K3G_ROWS; sa.row0[0] = sv::K3G_DA;
      sa.list = a.sp.list; sa.count = &a.sp.hdr->count;
      sa.g_xyz = g_xyz;
      { int rc_ = launch_scatter("scatter_static_app", k_scatter<12, 3, 9>, sa, (long)t3, stream); if (rc_) return rc_; }
    }
    const bool fea = cfg->static_head == RDRF_HEAD_MLP_FEA;
    const int in1 = fea ? 138 : 135;
    const int* cnt = &a.sp.hdr->count;
    DwJobs D;
    D.n = 0;
    dw_add(D, b.grows3, sv::K3G_ROWS, sv::K3G_DZV, 1, 3, 0, a.sp.act3, sv::S3_ROWS, 128,
           fea ? 128 : 131, G->w3, G->b3, cnt, 0);
    for (int i = 0; i < 4; ++i) dw_blk(D, sv::S3_H2 + 32 * i, SEG_IDENT, 32 * i);
    if (!fea) dw_blk(D, sv::S3_VD, SEG_VIEW3, 0);
    dw_add(D, b.grows3, sv::K3G_ROWS, sv::K3G_DF, 1, 27, 0, a.sp.act3, sv::S3_ROWS, 72, 72, G->basis,
           nullptr, cnt, 0);
    for (int i = 0; i < 3; ++i) dw_blk(D, sv::S3_G + 32 * i, SEG_IDENT, 32 * i);
    dw_add(D, b.grows3, sv::K3G_ROWS, sv::K3G_DZ2, 4, 128, 0, a.sp.act3, sv::S3_ROWS, 128, 128, G->w2,
           G->b2, cnt, 0);
    for (int i = 0; i < 4; ++i) dw_blk(D, sv::S3_H1 + 32 * i, SEG_IDENT, 32 * i);
    dw_add(D, b.grows3, sv::K3G_ROWS, sv::K3G_DZ1, 4, 128, 0, a.sp.act3, sv::S3_ROWS, in1, in1, G->w1,
           G->b1, cnt, 0);
    dw_blk(D, sv::S3_F, fea ? SEG_STAT1_F_FEA : SEG_STAT1_F_TE, 0);
    for (int i = 0; i < 4; ++i) dw_blk(D, sv::S3_P + 32 * i, fea ? SEG_STAT1_P_FEA : SEG_STAT1_P_TE, 32 * i);
    rc = dw_launch(D, stream, "dw_static");
    if (rc) return rc;
  }
  if (g_sigma != nullptr || g_weight != nullptr || ((g_rays != nullptr || g_z != nullptr) && g_dists != nullptr)) {
    RDRF_LAUNCH("static_density_bwd", k_static_density_bwd, dim3(N), dim3(64), stream, a, w, b.gf);
    if (g_sigma != nullptr || g_weight != nullptr) {
      ScatterArgs sa;
      fill_scatter_common(sa, a);
      sa.vm[0] = P->density; sa.gvm[0] = G->density; sa.nsets = 1;
      sa.rows = b.gf; sa.stride = 1; sa.row0[0] = 0; sa.bcast = 1;
      sa.g_xyz = g_xyz;
      const long t1 = (long)N * ((S + 31) / 32);
      { int rc_ = launch_scatter("scatter_static_density", k_scatter<4, 1, 3>, sa, t1, stream); if (rc_) return rc_; }
    }
  }
  return 0;
}

extern "C" int rdrf_dynamic_bwd(const RdrfDynamicParams* P, const RdrfFieldCfg* cfg,
                                const float* rays, const float* ts, const float* xyz,
                                const float* z, const uint8_t* valid, int N, int S,
                                const float* g_blending, const float* g_weight,
                                const float* g_xyz_prime, const float* g_rgb, const float* g_sigma,
                                const float* g_dists, const RdrfDynamicParams* G, float* g_xyz,
                                float* g_z, float* g_rays, void* saved, size_t saved_bytes, void* ws,
                                size_t ws_bytes, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(P && cfg && G && saved && N > 0 && S > 0 && S <= 4096, -1, "dynamic_bwd: bad arguments (S <= 4096)");
  RDRF_CHECK((size_t)N * S * 3 < (size_t)INT32_MAX, -1, "dynamic_bwd: N * S * 3 must stay below 2^31 (32-bit sample indices)");
  BwdArgs a;
  fill_bwd_common(a, cfg, rays, ts, xyz, z, valid, N, S);
  a.g_rgb = g_rgb; a.g_sigma = g_sigma; a.g_weight = g_weight; a.g_blending = g_blending;
  a.g_xyz_prime = g_xyz_prime; a.g_xyz = g_xyz; a.g_rays = g_rays; a.g_dists = g_dists; a.g_z = g_z;
  static const bool small_dw = !(RDRF_ENV("RDRF_DW_SMALL") && atoi(RDRF_ENV("RDRF_DW_SMALL")) == 0);   // 0: as k_dw2 products (tools build)
  a.small_dw = small_dw ? 1 : 0;
  RDRF_CHECK(carve_saved(a.sp, saved, saved_bytes, 1, N, S), -3, "dynamic_bwd: saved buffer too small");
  BwdWs b;
  int rc = carve_bwd(b, ws, ws_bytes, N, S, 1);
  if (rc) return rc;
  a.pk = b.pk; a.grows1 = b.grows1; a.grows3 = b.grows3; a.dxw_app = b.dxw; a.dxn_app = b.dxn;
  a.dtout = b.dtout;
  DynW w;
  fill_dyn_w(w, P);
  DynG gw;
  gw.density = G->density; gw.blending = G->blending; gw.app = G->app;
  gw.rbv = G->rbv; gw.rwv = G->rwv; gw.l5b = G->l5b; gw.db2 = G->db2; gw.bb2 = G->bb2;
  gw.l5w = G->l5w; gw.dw2 = G->dw2; gw.bw2 = G->bw2;
  if (P->packed_bwd != nullptr) a.pk = P->packed_bwd;   // caller-packed image (rdrf_dynamic_pack)
  else {
    PackJobs J;
    dyn_pack_jobs_bwd(J, P);
    rc = pack_launch(J, b.pk, stream);
    if (rc) return rc;
  }
  const size_t ns = (size_t)N * S, t3 = (ns + 31) / 32, t1 = (size_t)N * ((S + 31) / 32);
  RDRF_FILL(b.dxw, 0, ns * 3 * 4, stream);
  RDRF_FILL(b.dxn, 0, ns * 3 * 4, stream);
  const int* cnt = &a.sp.hdr->count;
  DwJobs D;
  D.n = 0;
  if (g_rgb != nullptr) {
    const Geo g = geo_for_units((long)t3);
    const int smode_app = scatter_mode(ns, stream);
    a.dfa = smode_app != 0 ? b.dfa : nullptr;   // sorted: k_dyn_app_bwd writes sample-major records instead of DA rows
    if (smode_app != 0) RDRF_LAUNCH("dyn_app_bwd", (k_dyn_app_bwd<false, true>), dim3(g.grid), dim3(g.block), stream, a, w, gw);
    else RDRF_LAUNCH("dyn_app_bwd", (k_dyn_app_bwd<false, false>), dim3(g.grid), dim3(g.block), stream, a, w, gw);
    if (smode_app != 0) {
      rdrf_prof_begin("scatter_dyn_app", stream);
      rc = scatter_dyn_app_sorted(a, b, P, G, stream);
      rdrf_prof_end("scatter_dyn_app", stream);
      if (rc) return rc;
    } else {
      ScatterArgs sa;
      fill_scatter_common(sa, a);
      sa.vm[0] = P->app; sa.gvm[0] = G->app; sa.nsets = 1;
      sa.rows = b.grows3; sa.stride = sv::K3G_ROWS; sa.row0[0] = sv::K3G_DA;
      sa.xw = a.sp.xw; sa.list = a.sp.list; sa.count = cnt;
      sa.dxw = b.dxw; sa.dxw_accumulate = 0;
      { int rc_ = launch_scatter("scatter_dyn_app", k_scatter<12, 3, 27>, sa, (long)t3, stream); if (rc_) return rc_; }
    }
    dw_add(D, b.grows3, sv::K3G_ROWS, sv::K3G_DZV, 1, 3, 0, a.sp.act3, sv::K3_ROWS, 128, 131, G->rwv,
           G->rbv, cnt, 0);
    for (int i = 0; i < 4; ++i) dw_blk(D, sv::K3_H2 + 32 * i, SEG_IDENT, 32 * i);
    dw_blk(D, sv::K3_VD, SEG_VIEW3, 0);
    dw_add(D, b.grows3, sv::K3G_ROWS, sv::K3G_DF, 1, 27, 0, a.sp.act3, sv::K3_ROWS, 216, 216, G->basis,
           nullptr, cnt, 0);
    for (int i = 0; i < 7; ++i) dw_blk(D, sv::K3_A + 32 * i, SEG_IDENT, 32 * i);
    dw_add(D, b.grows3, sv::K3G_ROWS, sv::K3G_DZ2, 4, 128, 0, a.sp.act3, sv::K3_ROWS, 128, 128, G->rw2,
           G->rb2, cnt, 0);
    for (int i = 0; i < 4; ++i) dw_blk(D, sv::K3_H1 + 32 * i, SEG_IDENT, 32 * i);
    dw_add(D, b.grows3, sv::K3G_ROWS, sv::K3G_DZ1, 4, 128, 0, a.sp.act3, sv::K3_ROWS, 107, 107, G->rw1,
           G->rb1, cnt, 0);
    dw_blk(D, sv::K3_F, SEG_RGB1_F, 0);
    dw_blk(D, sv::K3_X0, SEG_RGB1_X0, 0);
    dw_blk(D, sv::K3_X0 + 32, SEG_RGB1_X0, 32);
    dw_blk(D, sv::K3_X1, SEG_RGB1_X1, 0);
  }
  {
    const Geo g = geo_for_units(N);
    const int smode = scatter_mode(ns, stream);
    a.dfs = smode != 0 ? b.dfs : nullptr;
    RDRF_LAUNCH("dyn_heads_bwd", (k_dyn_density_bwd<0, false>), dim3(g.grid), dim3(g.block), stream, a, w, gw);
    if (smode != 0) {
      const int set_mask = ((g_sigma != nullptr || g_weight != nullptr) ? 1 : 0) | (g_blending != nullptr ? 2 : 0);
      if (set_mask != 0) {
        rdrf_prof_begin("scatter_dyn_density", stream);
        rc = scatter_dyn_density_sorted(a, b, P, G, set_mask, stream);
        rdrf_prof_end("scatter_dyn_density", stream);
        if (rc) return rc;
      }
    } else {
      // a head without an upstream gradient (passes B-D of the trainer carry none for the blending head before the late
      // mask terms) has d(features) == 0: its factor set is left out of the launch instead of being walked with zeros
      const bool has_d = g_sigma != nullptr || g_weight != nullptr, has_b = g_blending != nullptr;
      if (has_d || has_b) {
        ScatterArgs sa;
        fill_scatter_common(sa, a);
        sa.nsets = 0;
        if (has_d) { sa.vm[sa.nsets] = P->density; sa.gvm[sa.nsets] = G->density; sa.row0[sa.nsets] = sv::K1G_DFD; ++sa.nsets; }
        if (has_b) { sa.vm[sa.nsets] = P->blending; sa.gvm[sa.nsets] = G->blending; sa.row0[sa.nsets] = sv::K1G_DFB; ++sa.nsets; }
        sa.rows = b.grows1; sa.stride = sv::K1G_ROWS;
        sa.xw = a.sp.xw;
        sa.dxw = b.dxw; sa.dxw_accumulate = 1;
        { int rc_ = launch_scatter("scatter_dyn_density", k_scatter<4, 1, 9>, sa, (long)t1, stream); if (rc_) return rc_; }
      }
    }
    RDRF_LAUNCH("dyn_warp_bwd", (k_dyn_density_bwd<1, false>), dim3(g.grid), dim3(g.block), stream, a, w, gw);
    RDRF_LAUNCH("time_branch_bwd", k_time_branch_bwd, dim3((N + TB_RPB - 1) / TB_RPB), dim3(128), stream, ts, w,
                N, b.dtout, G->l1w, G->l1b, G->l2w, G->l2b);
    const bool small_in_kernel = a.small_dw != 0;
    add_density_phase_dw(D, b.grows1, a.sp.act1, G, (int)t1, g_sigma != nullptr || g_weight != nullptr, g_blending != nullptr,
                         small_in_kernel);
  }
  rc = dw_launch(D, stream, "dw_dyn");
  return rc;
}

// ------------------------------------------------------------------------------------------------
// feature mode backward (compute_* / warp_coordinate): gradients of the raw features wrt every
// parameter they depend on and wrt the input coordinates; pseudo-ray geometry (rdrf_fwd.hip)
// ------------------------------------------------------------------------------------------------
struct FeatWs {
  float *pk, *grows3, *grows1, *dxw, *dxn, *dtout, *gpad, *xpad;
  uint8_t* valid;
};
static int carve_feat_bwd(FeatWs& b, void* ws, size_t ws_bytes, int M, int dynamic) {
  WsCarver c(ws, ws_bytes);
  const size_t t = ((size_t)M + 31) / 32, mp = t * 32;
  b.pk = c.take<float>(PACK_AREA_FLOATS);
  b.grows3 = c.take<float>(t * sv::K3G_ROWS * 32);
  b.grows1 = dynamic ? c.take<float>(t * sv::K1G_ROWS * 32) : nullptr;
  b.dxw = dynamic ? c.take<float>(mp * 3) : nullptr;
  b.dxn = dynamic ? c.take<float>(mp * 3) : nullptr;
  b.dtout = dynamic ? c.take<float>(mp * 32) : nullptr;
  b.gpad = dynamic ? nullptr : c.take<float>(mp);
  b.xpad = dynamic ? nullptr : c.take<float>(mp * 3);
  b.valid = c.take<uint8_t>(mp);
  RDRF_CHECK(c.ok(), -3, "features backward: workspace too small: need %zu have %zu", c.off, ws_bytes);
  return 0;
}
extern "C" size_t rdrf_features_bwd_workspace_bytes(int M) {
  const size_t t = ((size_t)M + 31) / 32, mp = t * 32;
  return (size_t)PACK_AREA_FLOATS * 4 + t * (sv::K3G_ROWS + sv::K1G_ROWS) * 32 * 4 + mp * (6 + 32 + 4) * 4 + mp +
         (1 << 14);
}

extern "C" int rdrf_static_features_bwd(const RdrfStaticParams* P, const RdrfFieldCfg* cfg, const float* xn,
                                        int M, const float* g_density, const float* g_app,
                                        const RdrfStaticParams* G, float* g_xn, void* saved, size_t saved_bytes,
                                        void* ws, size_t ws_bytes, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(P && cfg && G && xn && M > 0 && (g_density || g_app), -1, "static_features_bwd: bad arguments");
  RDRF_CHECK(g_app == nullptr || saved != nullptr, -1, "static_features_bwd: the appearance features need the "
             "saved buffer of their forward call");
  const int Np = (M + 31) / 32;
  const size_t mp = (size_t)Np * 32;
  FeatWs b;
  int rc = carve_feat_bwd(b, ws, ws_bytes, M, 0);
  if (rc) return rc;
  BwdArgs a;
  fill_bwd_common(a, cfg, nullptr, nullptr, xn, nullptr, b.valid, Np, 32);
  a.M = M; a.in_norm = 1; a.g_feat = g_app; a.pk = b.pk; a.grows3 = b.grows3;
  if (saved != nullptr)
    RDRF_CHECK(carve_saved_feat(a.sp, saved, saved_bytes, 0, M), -3, "static_features_bwd: saved buffer too small");
  RDRF_FILL(b.valid, 0, mp, stream);
  RDRF_FILL(b.valid, 1, (size_t)M, stream);
  RDRF_FILL(b.xpad, 0, mp * 3 * 4, stream);
  RDRF_HIP(hipMemcpyAsync(b.xpad, xn, (size_t)M * 3 * 4, hipMemcpyDeviceToDevice, stream));
  ScatterArgs sa0;
  fill_scatter_common(sa0, a);
  sa0.xw = b.xpad;                       // already normalised
  sa0.box.inv[0] = sa0.box.inv[1] = sa0.box.inv[2] = 1.0f;   // g_xn is the gradient wrt the normalised input
  sa0.g_xyz = g_xn;
  if (g_density != nullptr) {            // the feature is the plain sum of the 24 products: broadcast rows
    RDRF_FILL(b.gpad, 0, mp * 4, stream);
    RDRF_HIP(hipMemcpyAsync(b.gpad, g_density, (size_t)M * 4, hipMemcpyDeviceToDevice, stream));
    ScatterArgs sa = sa0;
    sa.vm[0] = P->density; sa.gvm[0] = G->density; sa.nsets = 1;
    sa.rows = b.gpad; sa.stride = 1; sa.row0[0] = 0; sa.bcast = 1;
    { int rc_ = launch_scatter("feat_scatter_static_density", k_scatter<4, 1, 3>, sa, (long)Np, stream); if (rc_) return rc_; }
  }
  if (g_app != nullptr) {
    StaticW w;
    fill_static_w(w, P);
    StaticG gw;
    gw.density = G->density; gw.app = G->app; gw.b3 = G->b3; gw.w3 = G->w3;
    if (P->packed_bwd != nullptr) a.pk = P->packed_bwd;
    else {
      PackJobs J;
      static_pack_jobs_bwd(J, P, cfg->static_head);
      rc = pack_launch(J, b.pk, stream);
      if (rc) return rc;
    }
    const Geo g = geo_for_units(Np);
    RDRF_LAUNCH("feat_static_app_bwd", (k_static_app_bwd<RDRF_HEAD_MLP_FEA, true>), dim3(g.grid), dim3(g.block),
                stream, a, w, gw);
    ScatterArgs sa = sa0;
    sa.vm[0] = P->app; sa.gvm[0] = G->app; sa.nsets = 1;
    sa.rows = b.grows3; sa.stride = sv::K3G_ROWS; sa.row0[0] = sv::K3G_DA;
    { int rc_ = launch_scatter("feat_scatter_static_app", k_scatter<12, 3, 9>, sa, (long)Np, stream); if (rc_) return rc_; }
    DwJobs D;
    D.n = 0;
    dw_add(D, b.grows3, sv::K3G_ROWS, sv::K3G_DF, 1, 27, 0, a.sp.act3, sv::S3_ROWS, 72, 72, G->basis, nullptr,
           nullptr, Np);
    for (int i = 0; i < 3; ++i) dw_blk(D, sv::S3_G + 32 * i, SEG_IDENT, 32 * i);
    rc = dw_launch(D, stream, "feat_dw_static");
    if (rc) return rc;
  }
  return 0;
}

extern "C" int rdrf_dynamic_features_bwd(const RdrfDynamicParams* P, const RdrfFieldCfg* cfg, const float* x,
                                         const float* t, int M, int x_is_normalized, const float* g_density,
                                         const float* g_blending, const float* g_app, const float* g_xyz_prime,
                                         const RdrfDynamicParams* G, float* g_x, void* saved, size_t saved_bytes,
                                         void* ws, size_t ws_bytes, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(P && cfg && G && x && t && saved && M > 0 && (g_density || g_blending || g_app || g_xyz_prime), -1,
             "dynamic_features_bwd: bad arguments");
  const int Np = (M + 31) / 32;
  const size_t mp = (size_t)Np * 32;
  FeatWs b;
  int rc = carve_feat_bwd(b, ws, ws_bytes, M, 1);
  if (rc) return rc;
  BwdArgs a;
  fill_bwd_common(a, cfg, nullptr, t, x, nullptr, b.valid, Np, 32);
  a.M = M; a.in_norm = x_is_normalized ? 1 : 0;
  a.g_sigma = g_density; a.g_blending = g_blending; a.g_feat = g_app; a.g_xyz_prime = g_xyz_prime; a.g_xyz = g_x;
  RDRF_CHECK(carve_saved_feat(a.sp, saved, saved_bytes, 1, M), -3, "dynamic_features_bwd: saved buffer too small");
  a.pk = b.pk; a.grows1 = b.grows1; a.grows3 = b.grows3; a.dxw_app = b.dxw; a.dxn_app = b.dxn; a.dtout = b.dtout;
  DynW w;
  fill_dyn_w(w, P);
  DynG gw;
  gw.density = G->density; gw.blending = G->blending; gw.app = G->app;
  gw.rbv = G->rbv; gw.rwv = G->rwv; gw.l5b = G->l5b; gw.db2 = G->db2; gw.bb2 = G->bb2;
  gw.l5w = G->l5w; gw.dw2 = G->dw2; gw.bw2 = G->bw2;
  if (P->packed_bwd != nullptr) a.pk = P->packed_bwd;   // caller-packed image (rdrf_dynamic_pack)
  else {
    PackJobs J;
    dyn_pack_jobs_bwd(J, P);
    rc = pack_launch(J, b.pk, stream);
    if (rc) return rc;
  }
  RDRF_FILL(b.valid, 0, mp, stream);
  RDRF_FILL(b.valid, 1, (size_t)M, stream);
  RDRF_FILL(b.dxw, 0, mp * 3 * 4, stream);
  RDRF_FILL(b.dxn, 0, mp * 3 * 4, stream);
  const Geo g = geo_for_units(Np);
  DwJobs D;
  D.n = 0;
  if (g_app != nullptr) {
    RDRF_LAUNCH("feat_dyn_app_bwd", k_dyn_app_bwd<true>, dim3(g.grid), dim3(g.block), stream, a, w, gw);
    ScatterArgs sa;
    fill_scatter_common(sa, a);
    sa.vm[0] = P->app; sa.gvm[0] = G->app; sa.nsets = 1;
    sa.rows = b.grows3; sa.stride = sv::K3G_ROWS; sa.row0[0] = sv::K3G_DA;
    sa.xw = a.sp.xw;
    sa.dxw = b.dxw; sa.dxw_accumulate = 0;
    { int rc_ = launch_scatter("feat_scatter_dyn_app", k_scatter<12, 3, 27>, sa, (long)Np, stream); if (rc_) return rc_; }
    dw_add(D, b.grows3, sv::K3G_ROWS, sv::K3G_DF, 1, 27, 0, a.sp.act3, sv::K3_ROWS, 216, 216, G->basis, nullptr,
           nullptr, Np);
    for (int i = 0; i < 7; ++i) dw_blk(D, sv::K3_A + 32 * i, SEG_IDENT, 32 * i);
  }
  RDRF_LAUNCH("feat_dyn_heads_bwd", (k_dyn_density_bwd<0, true>), dim3(g.grid), dim3(g.block), stream, a, w, gw);
  if (g_density != nullptr || g_blending != nullptr) {
    ScatterArgs sa;
    fill_scatter_common(sa, a);
    sa.nsets = 0;   // live heads only (see k_dyn_density_bwd<0>)
    if (g_density != nullptr) { sa.vm[sa.nsets] = P->density; sa.gvm[sa.nsets] = G->density; sa.row0[sa.nsets] = sv::K1G_DFD; ++sa.nsets; }
    if (g_blending != nullptr) { sa.vm[sa.nsets] = P->blending; sa.gvm[sa.nsets] = G->blending; sa.row0[sa.nsets] = sv::K1G_DFB; ++sa.nsets; }
    sa.rows = b.grows1; sa.stride = sv::K1G_ROWS;
    sa.xw = a.sp.xw;
    sa.dxw = b.dxw; sa.dxw_accumulate = 1;
    { int rc_ = launch_scatter("feat_scatter_dyn_density", k_scatter<4, 1, 9>, sa, (long)Np, stream); if (rc_) return rc_; }
  }
  RDRF_LAUNCH("feat_dyn_warp_bwd", (k_dyn_density_bwd<1, true>), dim3(g.grid), dim3(g.block), stream, a, w, gw);
  RDRF_LAUNCH("time_branch_bwd", k_time_branch_bwd, dim3((M + TB_RPB - 1) / TB_RPB), dim3(128), stream, t, w, M,
              b.dtout, G->l1w, G->l1b, G->l2w, G->l2b);
  add_density_phase_dw(D, b.grows1, a.sp.act1, G, Np, g_density != nullptr, g_blending != nullptr);
  return dw_launch(D, stream, "feat_dw_dyn");
}

extern "C" int rdrf_scene_flow_bwd(const RdrfDynamicParams* P, const RdrfFieldCfg* cfg,
                                   const float* pts, const float* ts, int N, int S,
                                   const float* g_sf_f, const float* g_sf_b,
                                   const RdrfDynamicParams* G, float* g_pts, void* saved,
                                   size_t saved_bytes, void* ws, size_t ws_bytes,
                                   rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(P && cfg && G && saved && N > 0 && S > 0, -1, "scene_flow_bwd: bad arguments");
  RDRF_CHECK((size_t)N * S * 3 < (size_t)INT32_MAX, -1, "scene_flow_bwd: N * S * 3 must stay below 2^31 (32-bit sample indices)");
  const size_t tiles = ((size_t)N * S + 31) / 32;
  RDRF_CHECK(saved_bytes >= tiles * sv::SF_ROWS * 32 * 4, -3, "scene_flow_bwd: saved buffer too small");
  WsCarver c(ws, ws_bytes);
  float* pkbuf = c.take<float>(PACK_AREA_FLOATS);
  float* grows = c.take<float>(tiles * sv::SFG_ROWS * 32);
  RDRF_CHECK(c.ok(), -3, "scene_flow_bwd: workspace too small: need %zu have %zu", c.off, ws_bytes);
  const float* pkimg = pkbuf;
  int rc = 0;
  if (P->packed_bwd != nullptr) pkimg = P->packed_bwd;
  else {
    PackJobs J;
    dyn_pack_jobs_bwd(J, P);
    rc = pack_launch(J, pkbuf, stream);
    if (rc) return rc;
  }
  const Geo g = geo_for_units((long)tiles);
  RDRF_LAUNCH("scene_flow_bwd", k_scene_flow_bwd, dim3(g.grid), dim3(g.block), stream, N, S,
              make_box(cfg), pkimg, (const float*)saved, grows, g_sf_f, g_sf_b, G->sfb[3], g_pts);
  const float* act = (const float*)saved;
  const int T = (int)tiles;
  DwJobs D;
  D.n = 0;
  dw_add(D, grows, sv::SFG_ROWS, sv::SFG_DZ6, 1, 6, 0, act, sv::SF_ROWS, 64, 64, G->sfw[3], G->sfb[3],
         nullptr, T);
  dw_blk(D, sv::SF_H4, SEG_IDENT, 0);
  dw_blk(D, sv::SF_H4 + 32, SEG_IDENT, 32);
  dw_add(D, grows, sv::SFG_ROWS, sv::SFG_DZ4, 2, 64, 0, act, sv::SF_ROWS, 64, 64, G->sfw[2], G->sfb[2],
         nullptr, T);
  dw_blk(D, sv::SF_H2, SEG_IDENT, 0);
  dw_blk(D, sv::SF_H2 + 32, SEG_IDENT, 32);
  dw_add(D, grows, sv::SFG_ROWS, sv::SFG_DZ2, 2, 64, 0, act, sv::SF_ROWS, 64, 64, G->sfw[1], G->sfb[1],
         nullptr, T);
  dw_blk(D, sv::SF_H0, SEG_IDENT, 0);
  dw_blk(D, sv::SF_H0 + 32, SEG_IDENT, 32);
  dw_add(D, grows, sv::SFG_ROWS, sv::SFG_DZ0, 2, 64, 0, act, sv::SF_ROWS, 36, 36, G->sfw[0], G->sfb[0],
         nullptr, T);
  dw_blk(D, sv::SF_X, SEG_SF_X, 0);
  dw_blk(D, sv::SF_X + 32, SEG_SF_X, 32);
  return dw_launch(D, stream, "dw_sf");
}

// ------------------------------------------------------------------------------------------------
// ray generation backward: hand-written adjoint of k_generate_rays (rdrf_misc.hip)
// The gradients of a batch land on T x 9 pose entries and ONE focal length: one global atomic per ray and entry was
// 36 864 atomics on 108 addresses for a 4096-ray launch at T = 12 (109 us; five launches per iteration of the
// pose-optimising configs).  Each workgroup now accumulates its rays in LDS (ds_add_f32) and issues one global atomic
// per touched entry: GRB_LDS_POSES pose rows fit (any longer table falls back to global atomics).
#define GRB_LDS_POSES 448
__global__ __launch_bounds__(256) void k_generate_rays_bwd(const int64_t* __restrict__ ids, const float* __restrict__ uv, int view_shift,
                                    const float* __restrict__ poses9,
                                    const float* __restrict__ focal_p, int N, int T, int H, int W,
                                    int ndc, float near, const float* __restrict__ g_rays,
                                    float* __restrict__ g_poses, float* __restrict__ g_focal) {
  __shared__ float s_gp[GRB_LDS_POSES * 9];
  __shared__ float s_gf[4];
  const bool in_lds = T <= GRB_LDS_POSES;   // (uniform)
  if (in_lds) {
    for (int i = threadIdx.x; i < T * 9; i += blockDim.x) s_gp[i] = 0.f;
    __syncthreads();
  }
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  float gf = 0.f;
  if (n < N) {
    const long id = ids[n];
    const int col = (int)(id % W), row = (int)((id / W) % H);
    int view = (int)(id / ((long)W * H)) + view_shift;
    view = view < 0 ? 0 : (view >= T ? T - 1 : view);
    const float f = focal_p[0];
    const float pu = uv ? uv[2 * n] : (float)col + 0.5f, pv = uv ? uv[2 * n + 1] : (float)row + 0.5f;
    const float dir[3] = {(pu - 0.5f * W) / f, -(pv - 0.5f * H) / f, -1.0f};
    const float* p = poses9 + view * 9;
    float b1[3] = {p[0], p[1], p[2]};
    const float n1 = sqrtf(b1[0] * b1[0] + b1[1] * b1[1] + b1[2] * b1[2]);
    for (int k = 0; k < 3; ++k) b1[k] /= n1;
    const float dt = b1[0] * p[3] + b1[1] * p[4] + b1[2] * p[5];
    float u[3] = {p[3] - dt * b1[0], p[4] - dt * b1[1], p[5] - dt * b1[2]};
    const float n2 = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    const float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2],
                         b1[0] * b2[1] - b1[1] * b2[0]};
    float d[3], o[3] = {p[6], p[7], p[8]};
    for (int r = 0; r < 3; ++r) d[r] = dir[0] * b1[r] + dir[1] * b2[r] + dir[2] * b3[r];
    const float* g = g_rays + (size_t)n * 6;
    float go[3] = {g[0], g[1], g[2]}, gd[3] = {g[3], g[4], g[5]};
    if (ndc) {
      const float t = -(near + o[2]) / d[2];
      const float op[3] = {o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]};
      const float kw = -2.0f * f / (float)W, kh = -2.0f * f / (float)H;
      const float aa = op[0] / op[2], bb = op[1] / op[2], ra = d[0] / d[2], rb = d[1] / d[2];
      const float g_kw = go[0] * aa + gd[0] * (ra - aa), g_kh = go[1] * bb + gd[1] * (rb - bb);
      gf += g_kw * (-2.0f / (float)W) + g_kh * (-2.0f / (float)H);
      const float g_a = kw * (go[0] - gd[0]), g_b = kh * (go[1] - gd[1]);
      const float g_ra = kw * gd[0], g_rb = kh * gd[1];
      float gop[3];
      gop[0] = g_a / op[2];
      gop[1] = g_b / op[2];
      gop[2] = -(g_a * aa + g_b * bb) / op[2] - 2.0f * near * go[2] / (op[2] * op[2]) +
               2.0f * near * gd[2] / (op[2] * op[2]);
      float gdd[3] = {g_ra / d[2], g_rb / d[2], -(g_ra * ra + g_rb * rb) / d[2]};
      const float g_t = gop[0] * d[0] + gop[1] * d[1] + gop[2] * d[2];
      for (int k = 0; k < 3; ++k) { go[k] = gop[k]; gdd[k] += t * gop[k]; }
      go[2] += -g_t / d[2];
      gdd[2] += -g_t * t / d[2];
      for (int k = 0; k < 3; ++k) gd[k] = gdd[k];
    }
    // d = sum_c dir_c b_c
    float gb1[3], gb2[3], gb3[3], gdir[3];
    for (int k = 0; k < 3; ++k) { gb1[k] = dir[0] * gd[k]; gb2[k] = dir[1] * gd[k]; gb3[k] = dir[2] * gd[k]; }
    gdir[0] = gd[0] * b1[0] + gd[1] * b1[1] + gd[2] * b1[2];
    gdir[1] = gd[0] * b2[0] + gd[1] * b2[1] + gd[2] * b2[2];
    gf += -gdir[0] * dir[0] / f - gdir[1] * dir[1] / f;
    // b3 = b1 x b2:  g_b1 += b2 x g_b3,  g_b2 += g_b3 x b1
    gb1[0] += b2[1] * gb3[2] - b2[2] * gb3[1]; gb1[1] += b2[2] * gb3[0] - b2[0] * gb3[2];
    gb1[2] += b2[0] * gb3[1] - b2[1] * gb3[0];
    gb2[0] += gb3[1] * b1[2] - gb3[2] * b1[1]; gb2[1] += gb3[2] * b1[0] - gb3[0] * b1[2];
    gb2[2] += gb3[0] * b1[1] - gb3[1] * b1[0];
    // b2 = u/|u|
    const float dot2 = gb2[0] * b2[0] + gb2[1] * b2[1] + gb2[2] * b2[2];
    float gu[3];
    for (int k = 0; k < 3; ++k) gu[k] = (gb2[k] - dot2 * b2[k]) / n2;
    float gp1[3];
    float g_dt = 0.f;
    for (int k = 0; k < 3; ++k) { gp1[k] = gu[k]; g_dt -= gu[k] * b1[k]; gb1[k] -= dt * gu[k]; }
    for (int k = 0; k < 3; ++k) { gb1[k] += g_dt * p[3 + k]; gp1[k] += g_dt * b1[k]; }
    const float dot1 = gb1[0] * b1[0] + gb1[1] * b1[1] + gb1[2] * b1[2];
    float* gp = in_lds ? s_gp + view * 9 : g_poses + view * 9;
    for (int k = 0; k < 3; ++k) {
      atomicAdd(gp + k, (gb1[k] - dot1 * b1[k]) / n1);
      atomicAdd(gp + 3 + k, gp1[k]);
      atomicAdd(gp + 6 + k, go[k]);
    }
  }
  gf = wave_sum(gf);
  if ((threadIdx.x & 63) == 0) s_gf[threadIdx.x >> 6] = gf;
  __syncthreads();
  if (in_lds)
    for (int i = threadIdx.x; i < T * 9; i += blockDim.x)
      if (s_gp[i] != 0.f) atomicAdd(g_poses + i, s_gp[i]);
  if (threadIdx.x == 0) {
    const float t = (s_gf[0] + s_gf[1]) + (s_gf[2] + s_gf[3]);
    if (t != 0.f) atomicAdd(g_focal, t);
  }
}

extern "C" int rdrf_generate_rays_uv_bwd(const int64_t* ids, const float* uv, int view_shift, const float* poses9,
                                         const float* focal, int N, int T, int H, int W, int ndc, float near,
                                         const float* grad_rays, float* grad_poses9, float* grad_focal,
                                         rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return 0;   // empty batch: a no-op, like torch ops on empty tensors (their data pointers are null)
  RDRF_CHECK(N > 0 && T > 0 && grad_rays && grad_poses9 && grad_focal, -1, "generate_rays_bwd: bad arguments");
  RDRF_LAUNCH("generate_rays_bwd", k_generate_rays_bwd, dim3((N + 255) / 256), dim3(256), stream, ids, uv,
              view_shift, poses9, focal, N, T, H, W, ndc, near, grad_rays, grad_poses9, grad_focal);
  return 0;
}
extern "C" int rdrf_generate_rays_bwd(const int64_t* ids, const float* poses9, const float* focal,
                                      int N, int T, int H, int W, int ndc, float near,
                                      const float* grad_rays, float* grad_poses9, float* grad_focal,
                                      rdrf_stream_t stream_) {
  return rdrf_generate_rays_uv_bwd(ids, nullptr, 0, poses9, focal, N, T, H, W, ndc, near, grad_rays, grad_poses9,
                                   grad_focal, stream_);
}

// ------------------------------------------------------------------------------------------------
// caller-managed packed weight images (include/rodynrf.h)
// ------------------------------------------------------------------------------------------------
void dyn_pack_jobs_fwd(PackJobs& J, const RdrfDynamicParams* P);
void static_pack_jobs_fwd(PackJobs& J, const RdrfStaticParams* P, int head);
extern "C" size_t rdrf_pack_floats(void) { return PACK_AREA_FLOATS; }
extern "C" int rdrf_static_pack(const RdrfStaticParams* P, int static_head, int backward, float* image,
                                rdrf_stream_t stream_) {
  RDRF_CHECK(P && image && (((uintptr_t)image) & 15) == 0, -1, "static_pack: bad arguments");
  PackJobs J;
  if (backward) static_pack_jobs_bwd(J, P, static_head); else static_pack_jobs_fwd(J, P, static_head);
  return pack_launch(J, image, (hipStream_t)stream_);
}
extern "C" int rdrf_dynamic_pack(const RdrfDynamicParams* P, int backward, float* image, rdrf_stream_t stream_) {
  RDRF_CHECK(P && image && (((uintptr_t)image) & 15) == 0, -1, "dynamic_pack: bad arguments");
  PackJobs J;
  if (backward) dyn_pack_jobs_bwd(J, P); else dyn_pack_jobs_fwd(J, P);
  return pack_launch(J, image, (hipStream_t)stream_);
}


// ------------------------------------------------------------------------------------------------
// deterministic build: bind a field's flat gradient buffer to its fixed-point shadow, fold it back
// ------------------------------------------------------------------------------------------------
#ifdef RDRF_DETERMINISTIC
int det_bind_optim(int slot, const float* base, size_t n, unsigned long long* shadow, hipStream_t stream);   // rdrf_optim.hip
static DetMap g_det_host[2];
__global__ void k_det_finish(float* __restrict__ g, unsigned long long* __restrict__ shadow, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const long long v = (long long)shadow[i];
    if (v != 0) {
      g[i] += (float)((double)v * (1.0 / (double)RDRF_DET_SCALE));
      shadow[i] = 0ull;
    }
  }
}
#endif

extern "C" int rdrf_deterministic(void) {
#ifdef RDRF_DETERMINISTIC
  return 1;
#else
  return 0;
#endif
}

extern "C" int rdrf_det_bind(int slot, float* grad_base, size_t n, void* shadow_i64, rdrf_stream_t stream_) {
#ifdef RDRF_DETERMINISTIC
  hipStream_t stream = (hipStream_t)stream_;
  RDRF_CHECK(slot == 0 || slot == 1, -1, "det_bind: slot 0 (static field) or 1 (dynamic field)");
  g_det_host[slot].base = grad_base;
  g_det_host[slot].n = n;
  g_det_host[slot].shadow = (unsigned long long*)shadow_i64;
  RDRF_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_det), &g_det_host[slot], sizeof(DetMap), slot * sizeof(DetMap),
                                  hipMemcpyHostToDevice, stream));
  return det_bind_optim(slot, grad_base, n, (unsigned long long*)shadow_i64, stream);
#else
  (void)slot; (void)grad_base; (void)n; (void)shadow_i64; (void)stream_;
  rdrf_set_error("det_bind: this library is the product build (fp32 atomics); load librodynrf_det.so (RDRF_DETERMINISTIC=1)");
  return -1;
#endif
}

extern "C" int rdrf_det_finish(int slot, rdrf_stream_t stream_) {
#ifdef RDRF_DETERMINISTIC
  hipStream_t stream = (hipStream_t)stream_;
  RDRF_CHECK((slot == 0 || slot == 1) && g_det_host[slot].shadow != nullptr, -1, "det_finish: slot %d is not bound", slot);
  const size_t n = g_det_host[slot].n;
  RDRF_LAUNCH("det_finish", k_det_finish, dim3((unsigned)((n + 1023) / 1024 > 4096 ? 4096 : (n + 1023) / 1024)), dim3(256), stream,
              (float*)g_det_host[slot].base, g_det_host[slot].shadow, n);
  return 0;
#else
  (void)slot; (void)stream_;
  rdrf_set_error("det_finish: product build");
  return -1;
#endif
}

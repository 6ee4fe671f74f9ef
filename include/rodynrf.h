/* rodynrf.h -- C ABI of the MI355X-native RoDynRF ray-batch hot path (librodynrf.so).
 *
 * The reference (facebookresearch/robust-dynrf) is pure Python: it has no FFI.  Its boundary for
 * this path is the Python call surface of three objects:
 *
 *   TensorVMSplit.forward / TensorVMSplit_TimeEmbedding.forward   models/tensorBase.py:704-850
 *   renderer.sampleXYZ                                            renderer.py:147-170
 *   renderer.raw2outputs                                          renderer.py:173-315
 *   TensorVMSplit_TimeEmbedding.get_forward_backward_scene_flow   models/tensoRF.py:446-462
 *   ray generation (ids2pixel .. ndc_rays_blender2)               train.py:96-103, 1062-1077
 *
 * Each entry point below replaces one of those calls (cited per function).  The host-side mirror
 * (robust-dynrf_amd/ *.py) binds them with ctypes and re-exposes the reference signatures.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; all tensors fp32, row major;
 *     masks are uint8; indices int32/int64 as stated.
 *   - return 0 on success, a negative code on failure; rdrf_last_error() gives the message
 *     (thread local). No entry point throws, allocates device memory, or synchronises: outputs and
 *     the workspace are caller-allocated, work is enqueued on `stream`.
 *   - VM factors are CHANNEL-LAST: the reference's (1,C,H,W) plane tensors with the component
 *     axis contiguous and explicit h/w strides (RdrfVM); line i is [L_i][C_i].  plane 0/1/2 =
 *     XY/XZ/YZ, line 0/1/2 = Z/Y/X (matMode/vecMode, models/tensorBase.py:326-327).  The host
 *     mirror stores every plane as [H][W][C] (plane 0 [y][x][C], planes 1,2 [z][x|y][C]): the two
 *     bilinear columns of a tap are then adjacent in memory, which the scatter exploits (one L2
 *     atomic request for both); any other h/w strides are accepted.
 *   - gradients are ACCUMULATED (+=) into the caller's (zero-initialised) buffers.
 */
#ifndef RODYNRF_H
#define RODYNRF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: RdrfStaticParams / RdrfDynamicParams grew the trailing packed_fwd / packed_bwd pointers (round 2);
 * 3: sorted scatter workspace + fused render entry points (round 3);
 * 4: rdrf_set_scatter_mode replaces the RDRF_SCATTER / RDRF_RENDER environment switches -- no entry point reads the
 *    caller's environment any more; rdrf_render_chunks_fwd (round 4).  A binding built against another version must refuse to load: the structs
 *    are passed by pointer and read to their full length.
 * 5: rdrf_saved_row_bytes and RDRF_SCATTER_SORTED_PLAIN (added in round 5 under version 4: a stale version-4 library then failed
 *    on the missing symbol instead of on the version check); rdrf_render_chunks_fwd coalesces the chunks that fit the
 *    caller's workspace into one launch sequence (same results, round 6).
 * 6: iteration scalars may live on the DEVICE, so that a whole training iteration can be captured in one HIP graph and
 *    replayed while they change (train.py draws the white-background coin and ramps loss weights every iteration):
 *    rdrf_composite_fwd / _bwd take `white_dev` (a device float, 0 or 1, that overrides add_white_bg when non-NULL) and
 *    RdrfLossTerm grew the trailing `coef_dev` (a device float that multiplies `coef` when non-NULL);
 *    rdrf_rows_scatter_add. */
#define RDRF_ABI_VERSION 6

typedef void* rdrf_stream_t; /* hipStream_t */

enum { RDRF_RAY_NDC = 0, RDRF_RAY_CONTRACT = 1, RDRF_RAY_OTHER = 2 };
enum { RDRF_ACT_RELU = 0, RDRF_ACT_SOFTPLUS = 1 };
enum { RDRF_HEAD_MLP_FEA = 0, RDRF_HEAD_MLP_FEA_TIMEEMBEDDING = 1 };
enum { RDRF_SCATTER_RAY = 0, RDRF_SCATTER_SORTED = 1, RDRF_SCATTER_AUTO = 2, RDRF_SCATTER_SORTED_PLAIN = 3 };

/* One vector-matrix factor set (3 planes + 3 lines).  Components are always contiguous (channel
 * stride 1); the texel strides are explicit so that the planes containing the ray-marching axis can
 * be stored with that axis fastest: element (c,h,w) of plane i lives at h*sH[i] + w*sW[i] + c. */
typedef struct {
  float* plane[3]; /* logical (C,H,W) */
  float* line[3];  /* [L][C]    */
  int C[3];        /* components per plane/line pair: {16,4,4} or {48,12,12} */
  int H[3], W[3];  /* plane i: H = grid[matMode[i][1]], W = grid[matMode[i][0]] */
  int L[3];        /* line i:  L = grid[vecMode[i]] */
  int sH[3], sW[3]; /* float strides of one step in h / w */
} RdrfVM;

/* Scalars the path reads from the field object (models/tensorBase.py:282-339, get_kwargs). */
typedef struct {
  float aabb[6];        /* min xyz, max xyz */
  float distance_scale; /* 25 */
  float weight_thres;   /* rayMarch_weight_thres, 1e-4 */
  float density_shift;  /* softplus shift */
  int act;              /* RDRF_ACT_* (fea2denseAct) */
  int ray_type;         /* RDRF_RAY_* */
  int static_head;      /* RDRF_HEAD_* (static field only) */
} RdrfFieldCfg;

/* Static TensorVMSplit parameters (state_dict names in comments). Used for values AND, with the
 * same struct filled with gradient buffers, for gradients. */
typedef struct {
  RdrfVM density, app;  /* density_plane/line.{0,1,2}, app_plane/line.{0,1,2} */
  float* basis;         /* basis_mat.weight (27,72) */
  float *w1, *b1;       /* renderModule.mlp.0 (128,138) | (128,135) */
  float *w2, *b2;       /* renderModule.mlp.2 (128,128) */
  float *w3, *b3;       /* renderModule.mlp.4 (3,128) | renderModule.mlp_view.0 (3,131) */
  /* optional (NULL = the entry point packs into its workspace, as before): images written by rdrf_static_pack
   * for THESE weights.  The caller owns their validity: re-pack after every change of w1..b3 / basis. */
  const float* packed_fwd;
  const float* packed_bwd;
} RdrfStaticParams;

/* Dynamic TensorVMSplit_TimeEmbedding parameters. */
typedef struct {
  RdrfVM density, blending, app;
  float* basis;                 /* basis_mat.weight (27,216) */
  float *rw1, *rb1;             /* renderModule.mlp.0 (128,107) */
  float *rw2, *rb2;             /* renderModule.mlp.2 (128,128) */
  float *rwv, *rbv;             /* renderModule.mlp_view.0 (3,131) */
  float *l1w, *l1b;             /* layer1 (64,17) */
  float *l2w, *l2b;             /* layer2 (30,64) */
  float *l3w, *l3b;             /* layer3 (64,93) */
  float *l4w, *l4b;             /* layer4 (64,64) */
  float *l5w, *l5b;             /* layer5 (3,64) */
  float *dw1, *db1, *dw2, *db2; /* density_layer1 (64,152), density_layer2 (1,64) */
  float *bw1, *bb1, *bw2, *bb2; /* blending_layer1/2 */
  float *sfw[4], *sfb[4];       /* scene_flow_mlp.{0,2,4,6}: (64,36)(64,64)(64,64)(6,64) */
  const float* packed_fwd;      /* optional images of rdrf_dynamic_pack (see RdrfStaticParams) */
  const float* packed_bwd;
} RdrfDynamicParams;

int rdrf_abi_version(void);
const char* rdrf_last_error(void);

/* Bytes of workspace a forward/backward call of either field needs for N rays x S samples. */
size_t rdrf_workspace_bytes(int N, int S);
/* The forward calls alone (inference: rdrf_*_fwd with saved == NULL, rdrf_render_*) need only this much of it
 * (weight pack area, compaction list, warped coordinates: ~16 B per sample + 4 MB instead of ~6 KB per sample). */
size_t rdrf_forward_workspace_bytes(int N, int S);
/* Training mode: a forward call given a `saved` buffer of this size (kind 0 = static field,
 * 1 = dynamic field, 2 = scene flow) stores the activations its backward needs; the matching
 * *_bwd call must receive the same buffer untouched.  saved == NULL => inference, nothing kept. */
size_t rdrf_saved_bytes(int kind, int N, int S);
/* Bytes per sample of the activation rows inside that buffer (a measurement aid, bench.py `saved_bytes_per_sample`):
 * phase 0 = dynamic field, density phase (every sample); 1 = dynamic field, appearance phase (per sample that passes the
 * weight > 1e-4 mask, models/tensorBase.py:773-790); 2 = static field, appearance phase (per masked sample); 3 = scene flow. */
size_t rdrf_saved_row_bytes(int phase);

/* ---- ray generation: train.py:96-103 + dataLoader/ray_utils.py:53-140 + camera.py:8-15 -------
 * ids[N] int64 flat ray ids over (T,H,W); poses9[T][9] 6-D rotation + translation; focal scalar.
 * rays[N][6]; if ndc != 0 the NDC warp with near = `near` is applied (ndc_rays_blender2). */
int rdrf_generate_rays(const int64_t* ids, const float* poses9, const float* focal, int N, int T,
                       int H, int W, int ndc, float near, float* rays, rdrf_stream_t stream);
/* grad_rays[N][6] -> grad_poses9[T][9] (+=), grad_focal[1] (+=) */
int rdrf_generate_rays_bwd(const int64_t* ids, const float* poses9, const float* focal, int N,
                           int T, int H, int W, int ndc, float near, const float* grad_rays,
                           float* grad_poses9, float* grad_focal, rdrf_stream_t stream);

/* The flow-displaced rays of train.py:1433-1460, 1530-1557, 1968-1990, 2033-2050: uv[N][2] (may be NULL)
 * replaces the pixel centre of the ray id by (column + 0.5 + flow_x, row + 0.5 + flow_y); the camera is
 * frame id / (H W) + view_shift clamped to [0, T-1] (allposes_refine_f / allposes_refine_b). uv carries no
 * gradient (the flow is data). */
int rdrf_generate_rays_uv(const int64_t* ids, const float* uv, int view_shift, const float* poses9,
                          const float* focal, int N, int T, int H, int W, int ndc, float near, float* rays,
                          rdrf_stream_t stream);
int rdrf_generate_rays_uv_bwd(const int64_t* ids, const float* uv, int view_shift, const float* poses9,
                              const float* focal, int N, int T, int H, int W, int ndc, float near,
                              const float* grad_rays, float* grad_poses9, float* grad_focal,
                              rdrf_stream_t stream);

/* ---- renderer.sampleXYZ (renderer.py:147-170) ------------------------------------------------
 * NDC: models/tensorBase.py:487-499. jitter[S] (uniform [0,1), shared by all rays) or NULL. */
int rdrf_sample_ndc(const float* rays, int N, int S, float near, float far, const float* jitter,
                    const float aabb_host[6], float* xyz, float* z, uint8_t* valid,
                    rdrf_stream_t stream);
/* contract: models/tensorBase.py:524-559. jitter_inner[S-S/2+1], jitter_outer[S/2+1] or NULL. */
int rdrf_sample_contract(const float* rays, int N, int S, float near, float far,
                         const float* jitter_inner, const float* jitter_outer, float* xyz,
                         float* z, uint8_t* valid, rdrf_stream_t stream);
/* grad_xyz[N][S][3] -> grad_rays[N][6] (+=). z[N][S] are the sample depths the sampler returned. */
int rdrf_sample_bwd(const float* rays, const float* z, int N, int S, int ray_type,
                    const float* grad_xyz, float* grad_rays, rdrf_stream_t stream);

/* ---- TensorVMSplit.forward (models/tensorBase.py:704-850, models/tensoRF.py:118-196) ---------
 * outputs: rgb[N][S][3], sigma[N][S], weight[N][S], dists[N][S] (= dists*distance_scale).
 * rgb may be NULL (rdrf_static_fwd and rdrf_dynamic_fwd): the caller does not consume the colours -- passes B-D of the
 * trainer discard them (train.py:1166-1246, 1433-1625) -- and the appearance phase (gather, basis, RGB head) is not run;
 * every other output is unchanged.  A later backward of such a call must not carry a gradient for rgb. */
int rdrf_static_fwd(const RdrfStaticParams* P, const RdrfFieldCfg* cfg, const float* rays,
                    const float* ts, const float* xyz, const float* z, const uint8_t* valid, int N,
                    int S, float* rgb, float* sigma, float* weight, float* dists, void* saved,
                    size_t saved_bytes, void* ws, size_t ws_bytes, rdrf_stream_t stream);
/* g_* are gradients wrt the four outputs (any may be NULL = zero). G receives parameter
 * gradients (+=); g_xyz[N][S][3], g_z[N][S], g_rays[N][6] (+=, each may be NULL). */
int rdrf_static_bwd(const RdrfStaticParams* P, const RdrfFieldCfg* cfg, const float* rays,
                    const float* ts, const float* xyz, const float* z, const uint8_t* valid, int N,
                    int S, const float* g_rgb, const float* g_sigma, const float* g_weight,
                    const float* g_dists, const RdrfStaticParams* G, float* g_xyz, float* g_z,
                    float* g_rays, void* saved, size_t saved_bytes, void* ws, size_t ws_bytes,
                    rdrf_stream_t stream);

/* ---- TensorVMSplit_TimeEmbedding.forward (models/tensorBase.py:704-850,
 *      models/tensoRF.py:521-811) -- adds blending[N][S], xyz_prime[N][S][3]. */
int rdrf_dynamic_fwd(const RdrfDynamicParams* P, const RdrfFieldCfg* cfg, const float* rays,
                     const float* ts, const float* xyz, const float* z, const uint8_t* valid,
                     int N, int S, float* blending, float* weight, float* xyz_prime, float* rgb,
                     float* sigma, float* dists, void* saved, size_t saved_bytes, void* ws,
                     size_t ws_bytes, rdrf_stream_t stream);
int rdrf_dynamic_bwd(const RdrfDynamicParams* P, const RdrfFieldCfg* cfg, const float* rays,
                     const float* ts, const float* xyz, const float* z, const uint8_t* valid,
                     int N, int S, const float* g_blending, const float* g_weight,
                     const float* g_xyz_prime, const float* g_rgb, const float* g_sigma,
                     const float* g_dists, const RdrfDynamicParams* G, float* g_xyz, float* g_z,
                     float* g_rays, void* saved, size_t saved_bytes, void* ws, size_t ws_bytes,
                     rdrf_stream_t stream);

/* ---- compute_densityfeature / compute_appfeature / compute_blendingfeature / warp_coordinate -----
 * The per-point building blocks TensorBase.forward calls (models/tensoRF.py:118-196 static;
 * :521-541 warp_coordinate, :543-629 compute_blendingfeature, :646-732 compute_densityfeature,
 * :734-811 compute_appfeature dynamic), on a batch of M independent points.  They run the SAME
 * kernels as rdrf_*_fwd/bwd in a point geometry (32 points per wave tile, time per point).
 *   xn[M][3]     NORMALISED coordinates ([-1,1] inside the aabb; outside is zero padded, as grid_sample)
 *   density[M]   raw density feature (static: sum of the 24 VM products; dynamic: density_layer2 output)
 *   blending[M]  raw blending feature (before the sigmoid)
 *   app[M][27]   basis_mat output
 * Any output pointer may be NULL (not computed).  saved: NULL for inference, else a buffer of
 * rdrf_features_saved_bytes(dynamic, M) bytes that the matching *_bwd call needs.
 * dynamic: x is xn when x_is_normalized != 0 (compute_*), else UN-normalised coordinates
 * (warp_coordinate); t[M] is the time of every point; xyz_prime[M][3] = un-normalised warped point.
 * bwd: g_* gradients wrt the outputs (NULL = zero); parameter gradients accumulate (+=) into G;
 * g_x[M][3] (+=, may be NULL) is the gradient wrt the coordinates as they were passed in. */
size_t rdrf_features_saved_bytes(int dynamic, int M);
size_t rdrf_features_workspace_bytes(int M);
size_t rdrf_features_bwd_workspace_bytes(int M);
int rdrf_static_features_fwd(const RdrfStaticParams* P, const RdrfFieldCfg* cfg, const float* xn, int M,
                             float* density, float* app, void* saved, size_t saved_bytes, void* ws,
                             size_t ws_bytes, rdrf_stream_t stream);
int rdrf_static_features_bwd(const RdrfStaticParams* P, const RdrfFieldCfg* cfg, const float* xn, int M,
                             const float* g_density, const float* g_app, const RdrfStaticParams* G,
                             float* g_xn, void* saved, size_t saved_bytes, void* ws, size_t ws_bytes,
                             rdrf_stream_t stream);
int rdrf_dynamic_features_fwd(const RdrfDynamicParams* P, const RdrfFieldCfg* cfg, const float* x,
                              const float* t, int M, int x_is_normalized, float* density, float* blending,
                              float* app, float* xyz_prime, void* saved, size_t saved_bytes, void* ws,
                              size_t ws_bytes, rdrf_stream_t stream);
int rdrf_dynamic_features_bwd(const RdrfDynamicParams* P, const RdrfFieldCfg* cfg, const float* x,
                              const float* t, int M, int x_is_normalized, const float* g_density,
                              const float* g_blending, const float* g_app, const float* g_xyz_prime,
                              const RdrfDynamicParams* G, float* g_x, void* saved, size_t saved_bytes,
                              void* ws, size_t ws_bytes, rdrf_stream_t stream);

/* ---- get_forward_backward_scene_flow (models/tensoRF.py:446-462) -----------------------------
 * pts[N][S][3] un-normalised, ts[N] -> sf_f[N][S][3], sf_b[N][S][3]. */
int rdrf_scene_flow_fwd(const RdrfDynamicParams* P, const RdrfFieldCfg* cfg, const float* pts,
                        const float* ts, int N, int S, float* sf_f, float* sf_b, void* saved,
                        size_t saved_bytes, void* ws, size_t ws_bytes, rdrf_stream_t stream);
int rdrf_scene_flow_bwd(const RdrfDynamicParams* P, const RdrfFieldCfg* cfg, const float* pts,
                        const float* ts, int N, int S, const float* g_sf_f, const float* g_sf_b,
                        const RdrfDynamicParams* G, float* g_pts, void* saved, size_t saved_bytes,
                        void* ws, size_t ws_bytes, rdrf_stream_t stream);

/* ---- renderer.raw2outputs (renderer.py:173-315) ----------------------------------------------
 * add_white_bg: the caller draws the train-time coin (renderer.py:269).  white_dev (nullable): the same coin as a DEVICE
 * float (0.0f or 1.0f) read when the kernel runs -- it overrides add_white_bg, so a launch captured in a HIP graph follows
 * the coin of each replay.  out13: pointers to the 13
 * outputs in the reference's order: rgb_map_full[N][3], depth_map_full[N], acc_map_full[N],
 * weights_full[N][S], rgb_map_s, depth_map_s, acc_map_s, weights_s, rgb_map_d, depth_map_d,
 * acc_map_d, weights_d, dynamicness_map[N]. */
int rdrf_composite_fwd(const float* rgb_s, const float* sigma_s, const float* rgb_d,
                       const float* sigma_d, const float* dists, const float* blending,
                       const float* z, const float* rays, int N, int S, int ray_type,
                       int add_white_bg, const float* white_dev, float* const out13[13], rdrf_stream_t stream);
/* g_out13: gradients wrt the 13 outputs (entries may be NULL). g_in8: gradient buffers (+=) for
 * rgb_s, sigma_s, rgb_d, sigma_d, dists, blending, z, rays (entries may be NULL). */
int rdrf_composite_bwd(const float* rgb_s, const float* sigma_s, const float* rgb_d,
                       const float* sigma_d, const float* dists, const float* blending,
                       const float* z, const float* rays, int N, int S, int ray_type,
                       int add_white_bg, const float* white_dev, const float* const g_out13[13],
                       float* const g_in8[8], rdrf_stream_t stream);

/* ---- induced optical flow / disparity of the rendered 3-D point (renderer.py:1334-1392
 * render_3d_point + induce_flow; NDC2world :1266, world2NDC :1276, contract2world :1286).
 * P = sum_s w p + (1 - sum_s w) * far(rays); world = NDC2world(P) | contract2world(P); projected
 * with the neighbour pose c2w[N][3][4] and focal; flow = pixel - pts_2d, disp = 1 + 2/z_cam.
 * focal: DEVICE pointer to one float (it is a trained scalar). H, W: image size.
 * render_single_3d_point / induce_flow_single (renderer.py:1301, :1381) are the S = 1, w = 1 case.
 * bwd: gradient buffers accumulate (+=); g_rays, g_c2w[N][12], g_focal[1] may be NULL. */
int rdrf_induce_flow_fwd(int H, int W, const float* focal, const float* c2w, const float* weights,
                         const float* pts, const float* pts_2d, const float* rays, int N, int S,
                         int ray_type, float* flow, float* disp, rdrf_stream_t stream);
int rdrf_induce_flow_bwd(int H, int W, const float* focal, const float* c2w, const float* weights,
                         const float* pts, const float* rays, int N, int S, int ray_type,
                         const float* g_flow, const float* g_disp, float* g_weights, float* g_pts,
                         float* g_rays, float* g_c2w, float* g_focal, rdrf_stream_t stream);

/* ---- adjoint of a row gather rows[n] = table[idx[n]] (the camera matrices of the neighbour frames,
 * `allposes_refine[view +- 1]`, train.py:1895-1948, live when the poses are optimised): g_table[idx[n]][c] += g_rows[n][c].
 * idx [N] int64 in [0, R); g_rows [N][C]; g_table [R][C] accumulates (+=). */
int rdrf_rows_scatter_add(const int64_t* idx, const float* g_rows, int N, int R, int C, float* g_table,
                          rdrf_stream_t stream);

/* ---- distortion loss (train.py:1299-1312, 1685-1716, 1840-1856 call flatten_eff_distloss of the
 * un-vendored torch_efficient_distloss with ray_id = tile(arange(N), S), i.e. N rays x S points).
 * Published formulation (DVGO v2 / mip-NeRF 360), m ascending along a ray:
 *   loss_ray[n] = sum_i 2 w_i (m_i sum_{j<i} w_j - sum_{j<i} w_j m_j) + (1/3) sum_i interval w_i^2
 * (= sum_ij w_i w_j |m_i - m_j| + ...); the caller sums loss_ray and divides by N.
 * bwd: g_w[n][i] += g_ray[n] * (2 (m_i (P_i - Q_i) + (QM_i - PM_i)) + (2/3) interval w_i) with
 * P/PM exclusive prefix and Q/QM exclusive suffix sums of w and w m.  m gets no gradient (the
 * reference passes it detached / it is not trained).  interval_pt (per point, may be NULL) overrides
 * the scalar interval.  PARITY UNPINNED against the library itself (absent here): checked against an
 * O(S^2) evaluation of the double sum. */
int rdrf_distloss_fwd(const float* w, const float* m, float interval, const float* interval_pt,
                      int N, int S, float* loss_ray, rdrf_stream_t stream);
int rdrf_distloss_bwd(const float* w, const float* m, float interval, const float* interval_pt,
                      int N, int S, const float* g_ray, float* g_w, rdrf_stream_t stream);

/* ---- total-variation regulariser of the factor tensors (utils.py:157-181 TVLoss, applied to every
 * plane / line by models/tensoRF.py:100-116, 418-444 each iteration of configs/Nvidia.txt).
 * One launch handles up to RDRF_TV_MAX logical (1,C,H,W) views with arbitrary element strides
 * (channel-last storage included).  sums[t] = { sum (x[h+1]-x[h])^2 , sum (x[w+1]-x[w])^2 }; the
 * caller forms TVLoss_weight * 2 * (h_tv/count_h + w_tv/count_w) / batch exactly as the reference
 * does (a line has count_w = 0, so the reference's VALUE is 0/0 = NaN while its gradient is finite;
 * that is reproduced by doing this scalar arithmetic on the host side of the ABI, in torch).
 * bwd: g_x[t] (same strides as x[t]) += g_sums[t][0] * d h_tv/dx + g_sums[t][1] * d w_tv/dx, the
 * second term skipped when W == 1 (its coefficient is 2/0 there). g_sums is a DEVICE array. */
#define RDRF_TV_MAX 16
typedef struct {
  const float* x;
  float* g;            /* gradient buffer (bwd only) */
  int C, H, W;
  long long sC, sH, sW;  /* element strides of the (1,C,H,W) view */
} RdrfTensor4;
int rdrf_tv_fwd(const RdrfTensor4* t, int n, float* sums /* [n][2], overwritten */, rdrf_stream_t stream);
int rdrf_tv_bwd(const RdrfTensor4* t, int n, const float* g_sums /* [n][2] */, rdrf_stream_t stream);
/* The gradient of the TV terms in one pass, for any number of tensors: t[i].g (+=) receives
 * coef_host[i][0] * d(h_tv)/dx + coef_host[i][1] * d(w_tv)/dx (the W term is skipped when W == 1).  The caller
 * folds TVLoss_weight * 2 / (count * batch), the 1e-2 / 1e-3 plane / line factors of TV_loss_* and the
 * config's TV weight into the HOST coefficients: no forward sums, no scalar arithmetic on the device. */
int rdrf_tv_grad(const RdrfTensor4* t, int n, const float* coef_host /* [n][2] */, rdrf_stream_t stream);

/* ---- Adam over a flat parameter range (torch.optim.Adam as train.py:924-934 builds it: betas
 * (0.9, 0.99), eps 1e-8, no weight decay / amsgrad; the learning-rate decay of train.py:2608-2612 is the
 * caller's: it passes the current rates).  p, g, m, v: n floats each, 16-byte aligned, n % 4 == 0.
 * Elements [0, split) step with lr0 (the VM factors), [split, n) with lr1 (the networks).  step >= 1 is
 * the 1-based iteration for the bias corrections; grad_scale multiplies g first (1 / world size when the
 * data-parallel exchange summed the per-rank gradients). */
int rdrf_adam_step(float* p, const float* g, float* m, float* v, size_t n, size_t split, float lr0,
                   float lr1, float beta1, float beta2, float eps, int step, float grad_scale,
                   rdrf_stream_t stream);

/* ---- upsample_volume_grid (models/tensoRF.py:199-232, 814-850): bilinear, align_corners=True, of up to
 * RDRF_TV_MAX (1,C,H,W) views with arbitrary element strides (src[i] -> dst[i], C % 4 == 0) in one launch;
 * lines are the W == 1 case. */
int rdrf_upsample_bilinear(const RdrfTensor4* src, const RdrfTensor4* dst, int n, rdrf_stream_t stream);

/* ---- density_L1 / blending_L1 (models/tensoRF.py:80-98, 378-416): sum over the X x Y x Z grid of
 * |feature2density(sum_c plane x line)| without materialising any volume (the reference builds a
 * [1,24,X,Y,Z] tensor).  fwd: sum_out[0] = the sum (the caller divides by X*Y*Z).  bwd: g_mean[0] (device)
 * = d loss / d mean; plane / line gradients accumulate (+=) into gvm. {16,4,4}-component sets only. */
int rdrf_dense_l1_fwd(const RdrfVM* vm, int act, float density_shift, float* sum_out, rdrf_stream_t stream);
int rdrf_dense_l1_bwd(const RdrfVM* vm, const RdrfVM* gvm, int act, float density_shift, const float* g_mean,
                      rdrf_stream_t stream);

/* ---- packed weight images.  Every field entry point first re-lays its MLP weights out for the MFMA kernels
 * (a ~6 us launch, ~40 of them per training iteration).  The weights of a field do not change between the passes
 * of one iteration or the chunks of a render, so a caller may pack once and hand the image in through
 * packed_fwd / packed_bwd of the parameter struct.  image: rdrf_pack_floats() floats, 16-byte aligned.
 * backward = 0: the image of *_fwd / *_features_fwd / scene_flow_fwd / render_fwd; 1: of the *_bwd entry points.
 * The factor tensors (planes / lines) are NOT part of the image. */
size_t rdrf_pack_floats(void);
int rdrf_static_pack(const RdrfStaticParams* P, int static_head, int backward, float* image, rdrf_stream_t stream);
int rdrf_dynamic_pack(const RdrfDynamicParams* P, int backward, float* image, rdrf_stream_t stream);

/* ---- the per-ray / per-sample loss terms of one iteration, reduced in one launch (+ a finishing launch)
 * and differentiated in one launch.  Replaces the elementwise chains of train.py:1323-1331 (photometric),
 * :1341-1365 (dynamicness mask), :1392-1410 (induced flow, masked means), :1421, 1627 (scene flow, weighted by
 * the sample weights in this path), :1522-1524, 1619-1621 (induced disparity), :1828-1832 (static photometric,
 * background-masked), :2293-2299 (disparity smoothness).
 *   term = coef [* *coef_dev] * sum_rows w[row] * sum_cols rho(x + ysign * y) / Z
 *   rho = r^2 | |r| | r ;  Z = rows * cols (NORM_MEAN)  or  sum_rows w + 1e-8 (NORM_WEIGHT, a masked mean)
 * x, y: [rows][cols] contiguous (y nullable); w: [rows] (nullable = 1).  gx / gy (backward only, nullable):
 * d loss / d x, d loss / d y, WRITTEN (not accumulated).
 * fwd: partial = rdrf_loss_terms_workspace_floats(n) floats of scratch; out = 1 + 2 n floats:
 *      out[0] = sum of the terms, out[1 + k] = coef_k / Z_k, out[1 + n + k] = value of term k.
 * bwd: out as written by fwd; g_loss[0] (device) = d L / d out[0].  Deterministic (no float atomics). */
enum { RDRF_LOSS_SQUARE = 0, RDRF_LOSS_ABS = 1, RDRF_LOSS_IDENTITY = 2 };
enum { RDRF_LOSS_NORM_MEAN = 0, RDRF_LOSS_NORM_WEIGHT = 1 };
#define RDRF_MAX_LOSS_TERMS 32
typedef struct {
  const float* x;
  const float* y;
  const float* w;
  float* gx;
  float* gy;
  long long rows;
  int cols;
  int kind;   /* RDRF_LOSS_* */
  int norm;   /* RDRF_LOSS_NORM_* */
  float ysign;
  float coef;
  const float* coef_dev; /* nullable: DEVICE float multiplied into coef when the finishing kernel runs (loss weights that
                            change every iteration -- train.py:1299-1312 distortion ramp, Temp_static :1034-1036 -- inside
                            a captured HIP graph) */
} RdrfLossTerm;
size_t rdrf_loss_terms_workspace_floats(int n);
int rdrf_loss_terms_fwd(const RdrfLossTerm* terms, int n, float* partial, float* out, rdrf_stream_t stream);
/* The same in two stages, for data-parallel runs that want the single-process normalisers (the reference's
 * masked means divide by the mask sum of the WHOLE batch, train.py:1391-1394): stats[2k], stats[2k+1] = this rank's
 * (sum_rows w rho, sum_rows w) of term k; the caller sums `stats` over the ranks (all-reduce) and passes both to finish,
 * which normalises a masked mean by (sum over ranks of sum w) / world + 1e-8, so that the MEAN over ranks of the per-rank
 * losses / gradients is exactly the single-process value.  rdrf_loss_terms_fwd == stats + finish(local, local, 1). */
int rdrf_loss_terms_stats(const RdrfLossTerm* terms, int n, float* partial, float* stats, rdrf_stream_t stream);
int rdrf_loss_terms_finish(const RdrfLossTerm* terms, int n, const float* stats_local, const float* stats_global, int world,
                           float* out, rdrf_stream_t stream);
int rdrf_loss_terms_bwd(const RdrfLossTerm* terms, int n, const float* out, const float* g_loss,
                        rdrf_stream_t stream);

/* ---- per-frame median-normalised monocular depth loss: train.py:797-807 compute_depth_loss summed over the
 * frames of the batch and divided by the number of rays used (train.py:1636-1664 dynamic, 2097-2121 static):
 *   for each frame k with more than one (unmasked) ray:  t = median(p), s = mean|p - t|, u = (p - t) / (s + 1e-10),
 *   likewise v from gt;  L_k = sum (u - v)^2;      loss = coef * sum_k L_k / sum_k n_k.
 * The reference loops over the frames on the host (one sync each); here one workgroup per frame selects its
 * rays (in ray order), bitonic-sorts them in LDS for the median (torch.median: the lower middle element) and
 * writes the loss AND its gradient wrt pred in the same pass: g_raw[j] = d(sum_k L_k)/d pred[j], with the
 * median's gradient spread evenly over the elements equal to it (ATen's evenly_distribute_backward).
 * pred, gt [N]; frame [N] int64 in [0, T); mask [N] uint8 or NULL (NULL: every ray is used).
 * out[0] = loss, out[1] = coef / sum_k n_k (the factor g_raw is to be multiplied with), out[2] = sum_k n_k.
 * ws: rdrf_frame_depth_loss_workspace_bytes(N, T).  N <= 32768 (the gathered batch of a data-parallel run included). */
size_t rdrf_frame_depth_loss_workspace_bytes(int N, int T);
int rdrf_frame_depth_loss_fwd(const float* pred, const float* gt, const int64_t* frame, const uint8_t* mask,
                              int N, int T, float coef, float* out, float* g_raw, void* ws, size_t ws_bytes,
                              rdrf_stream_t stream);
/* g_pred[j] = g_loss[0] * out[1] * gscale * g_raw[j]   (gscale: world size when the batch was gathered over a
 * data-parallel group whose exchange averages the per-rank gradients; 1 otherwise) */
int rdrf_frame_depth_loss_bwd(const float* g_raw, const float* out, const float* g_loss, int N, float gscale, float* g_pred,
                              rdrf_stream_t stream);

/* ---- no-grad render of a ray chunk (renderer.py:740-812 loop body): sample -> static -> dynamic -> composite;
 * writes rgb_map_full[N][3], depth_map_full[N]; scratch for the per-sample tensors comes out of ws
 * (rdrf_render_workspace_bytes).
 *   rdrf_render_fused_fwd     ONE cooperative launch: persistent workgroups walk the sampler, both density phases, both
 *                             appearance phases and the compositor, separated by grid-wide barriers (the LDS is re-filled
 *                             with each phase's weight image).  Same device code as the per-phase kernels: identical bits.
 *   rdrf_render_sequence_fwd  the per-phase kernels as a launch sequence (11 stream operations).
 *   rdrf_render_fwd           the launch sequence (measured faster at every size on MI355X: 274 vs 378 us per 512-ray
 *                             chunk, equal on whole frames; csrc/rdrf_render.hip). */
size_t rdrf_render_workspace_bytes(int N, int S);
int rdrf_render_fwd(const RdrfStaticParams* PS, const RdrfFieldCfg* cfg_s,
                    const RdrfDynamicParams* PD, const RdrfFieldCfg* cfg_d, const float* rays,
                    const float* ts, int N, int S, float near, float far, float* rgb_map,
                    float* depth_map, void* ws, size_t ws_bytes, rdrf_stream_t stream);
int rdrf_render_fused_fwd(const RdrfStaticParams* PS, const RdrfFieldCfg* cfg_s,
                          const RdrfDynamicParams* PD, const RdrfFieldCfg* cfg_d, const float* rays,
                          const float* ts, int N, int S, float near, float far, float* rgb_map,
                          float* depth_map, void* ws, size_t ws_bytes, rdrf_stream_t stream);
int rdrf_render_sequence_fwd(const RdrfStaticParams* PS, const RdrfFieldCfg* cfg_s,
                             const RdrfDynamicParams* PD, const RdrfFieldCfg* cfg_d, const float* rays,
                             const float* ts, int N, int S, float near, float far, float* rgb_map,
                             float* depth_map, void* ws, size_t ws_bytes, rdrf_stream_t stream);

/* ---- the eval chunk loop of renderer.py:740-812 (chunk = 512 rays, renderer.py:732) as one native call: the launch
 * sequences of the chunks are issued round-robin on `nstreams` caller streams (0: all on main_stream), which first wait for
 * main_stream and which main_stream finally waits for.  Both parameter structs must carry packed_fwd; ws: nstreams slices of
 * rdrf_render_workspace_bytes(chunk, S) (rdrf_render_chunks_workspace_bytes).  Same bits as rdrf_render_fwd per chunk. */
size_t rdrf_render_chunks_workspace_bytes(int chunk, int S, int nstreams);
int rdrf_render_chunks_fwd(const RdrfStaticParams* PS, const RdrfFieldCfg* cfg_s, const RdrfDynamicParams* PD,
                           const RdrfFieldCfg* cfg_d, const float* rays, const float* ts, int N, int S, int chunk,
                           float near, float far, float* rgb_map, float* depth_map, void* ws, size_t ws_bytes,
                           rdrf_stream_t main_stream, const rdrf_stream_t* streams, int nstreams);

/* ---- process-wide choice of the density / blending scatter of the dynamic field's backward (models/tensoRF.py:646-811,
 * grid_sampler_2d_backward semantics either way): RDRF_SCATTER_RAY = ray tiles, RDRF_SCATTER_SORTED = samples grouped by
 * plane cell first (about 10x fewer memory-side atomic requests, a fixed grouping cost per launch), RDRF_SCATTER_AUTO
 * (default) = sorted from 300 k samples per launch.  The sorted density / blending passes form their plane sums in LDS
 * windows (k_scatter_tiled) wherever those fit beside two workgroups per CU; RDRF_SCATTER_SORTED_PLAIN forces the sorted
 * path WITHOUT the windows (every run tail goes to global memory: the form large grids fall back to), so that both forms
 * can be held to the same parity tests at any size.  Returns 0, or -1 for an unknown mode. */
int rdrf_set_scatter_mode(int mode);

/* ---- deterministic debugging build (librodynrf_det.so = the same sources with -DRDRF_DETERMINISTIC) ----------------
 * Every addition into a BOUND flat gradient buffer (scatter of the VM factors, line flushes, dW / bias sums, time
 * branch) goes to a 64-bit fixed-point shadow of that buffer (value * 2^40, integer atomics: order-independent) and
 * rdrf_det_finish adds the shadow into the fp32 gradients and clears it; the compaction lists of the appearance
 * phases are sorted, LDS line accumulators are bypassed.  Result: bit-identical parameter gradients run after run,
 * within fp32 rounding of the product build's.  rdrf_deterministic() = 1 in that build, 0 in the product build
 * (where bind / finish fail).  slot 0 = static field, 1 = dynamic field; shadow: n 64-bit words, zero-initialised. */
int rdrf_deterministic(void);
int rdrf_det_bind(int slot, float* grad_base, size_t n, void* shadow_i64, rdrf_stream_t stream);
int rdrf_det_finish(int slot, rdrf_stream_t stream);

/* ---- kernel self-tests (used by tests/ only): run the MFMA layer chain on a synthetic input
 * and return it for comparison with a numpy matmul. x[M][K] -> y[M][OUT] = relu(x W^T + b). */
int rdrf_selftest_mlp(const float* x, const float* w, const float* b, int M, int K, int OUT,
                      float* y, void* ws, size_t ws_bytes, rdrf_stream_t stream);

/* timing hook: average device time (ms) of the dominant kernel launches recorded with HIP events
 * since the last reset; used by bench.py for the roofline figure. */
void rdrf_prof_reset(void);
int rdrf_prof_enable(int on);
int rdrf_prof_get(const char* kernel, double* total_ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif

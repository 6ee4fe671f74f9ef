// rdrf_loss.hip -- the per-ray / per-sample loss terms of one training iteration as ONE reduction launch
// (+ a finishing launch) forward and ONE launch backward.  The reference writes every term as a chain of
// elementwise torch ops ( ((a - b) ** 2).mean(), masked means, (|sf| * w).mean(), ... : train.py:1262-1296,
// 1330-1371, 1391-1413, 1476-1528, 1786-1839 ); with the ray path in kernels those chains were ~250 of the
// ~550 launches of an iteration, each a few microseconds of launch latency on a few KB of data.
//
// A term is   coef [* *coef_dev] * sum_rows w[row] * sum_cols rho(x + ysign * y) / Z
//   rho  : square | abs | identity            Z : rows * cols (a mean)  |  sum_rows w + 1e-8 (a masked mean)
// which covers every photometric / mask / flow / disparity / scene-flow term of the step.  The per-frame
// median depth loss (a sort), the distortion loss and TV have their own kernels.
#include "rdrf_host.hpp"

#define LOSS_BLOCKS 128   // partial sums per term (deterministic two-stage reduction, no float atomics)
struct LossTermsK {
  RdrfLossTerm t[RDRF_MAX_LOSS_TERMS];
  int n;
};

RDRF_D float loss_rho(int kind, float r) { return kind == RDRF_LOSS_SQUARE ? r * r : (kind == RDRF_LOSS_ABS ? fabsf(r) : r); }
RDRF_D float loss_drho(int kind, float r) {
  return kind == RDRF_LOSS_SQUARE ? 2.0f * r : (kind == RDRF_LOSS_ABS ? (r > 0.f ? 1.0f : (r < 0.f ? -1.0f : 0.f)) : 1.0f);
}

// partial[(k * LOSS_BLOCKS + b) * 2 + {0, 1}] = block b's share of term k's weighted sum / weight sum
__global__ __launch_bounds__(256) void k_loss_fwd(LossTermsK T, float* __restrict__ partial) {
  const RdrfLossTerm& t = T.t[blockIdx.y];
  const long long total = t.rows * t.cols;
  float s = 0.f, ws = 0.f;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long row = t.cols == 1 ? e : e / t.cols;
    const float w = t.w ? t.w[row] : 1.0f;
    const float r = t.x[e] + (t.y ? t.ysign * t.y[e] : 0.f);
    s += w * loss_rho(t.kind, r);
    if (t.norm == RDRF_LOSS_NORM_WEIGHT && e - row * t.cols == 0) ws += w;
  }
  __shared__ float red[2][4];
  s = wave_sum(s); ws = wave_sum(ws);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = ws; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* p = partial + ((size_t)blockIdx.y * LOSS_BLOCKS + blockIdx.x) * 2;
    p[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    p[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// stats[k] = (sum_rows w rho, sum_rows w) of term k on this rank: the numerators / denominators a data-parallel run
// all-reduces when exact single-GPU loss normalisation is wanted (train.py:1391-1394)
__global__ __launch_bounds__(64) void k_loss_stats(LossTermsK T, const float* __restrict__ partial, float* __restrict__ stats) {
  const int k = threadIdx.x;
  if (k >= T.n) return;
  float s = 0.f, ws = 0.f;
  for (int b = 0; b < LOSS_BLOCKS; ++b) { s += partial[((size_t)k * LOSS_BLOCKS + b) * 2]; ws += partial[((size_t)k * LOSS_BLOCKS + b) * 2 + 1]; }
  stats[2 * k] = s;
  stats[2 * k + 1] = ws;
}

// out[0] = total loss; out[1 + k] = coef_k / Z_k (the backward's per-term factor); out[1 + n + k] = term k's value.
// local: this rank's (sum, weight sum) per term;  global: the same summed over `world` ranks (== local at world 1).
// A masked mean is normalised by the GLOBAL weight sum: Z = (sum_ranks ws) / world + 1e-8, so that the mean over ranks of
// the per-rank losses (the exchange's convention, optim.FlatAdam) is exactly the single-process masked mean.
__global__ __launch_bounds__(64) void k_loss_finish(LossTermsK T, const float* __restrict__ local, const float* __restrict__ global,
                                                    float world, float* __restrict__ out) {
  const int k = threadIdx.x;
  float val = 0.f;
  if (k < T.n) {
    const RdrfLossTerm& t = T.t[k];
    const float z = t.norm == RDRF_LOSS_NORM_WEIGHT ? global[2 * k + 1] / world + 1e-8f : (float)(t.rows * t.cols);
    const float scale = (t.coef_dev ? t.coef * t.coef_dev[0] : t.coef) / z;
    val = local[2 * k] * scale;
    out[1 + k] = scale;
    out[1 + T.n + k] = val;
  }
  __shared__ float vals[64];
  vals[k] = val;
  __syncthreads();
  if (k == 0) {
    float tot = 0.f;
    for (int i = 0; i < T.n; ++i) tot += vals[i];   // term order, like the reference's running `loss +=`
    out[0] = tot;
  }
}

__global__ __launch_bounds__(256) void k_loss_bwd(LossTermsK T, const float* __restrict__ out, const float* __restrict__ g_loss) {
  const RdrfLossTerm& t = T.t[blockIdx.y];
  if (!t.gx && !t.gy) return;
  const long long total = t.rows * t.cols;
  const float gs = g_loss[0] * out[1 + blockIdx.y];
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long row = t.cols == 1 ? e : e / t.cols;
    const float w = t.w ? t.w[row] : 1.0f;
    const float r = t.x[e] + (t.y ? t.ysign * t.y[e] : 0.f);
    const float g = gs * w * loss_drho(t.kind, r);
    if (t.gx) t.gx[e] = g;
    if (t.gy) t.gy[e] = t.ysign * g;
  }
}

static int loss_pack(LossTermsK& T, const RdrfLossTerm* terms, int n) {
  RDRF_CHECK(terms && n >= 1 && n <= RDRF_MAX_LOSS_TERMS, -1, "loss_terms: 1..%d terms", RDRF_MAX_LOSS_TERMS);
  memset(&T, 0, sizeof(T));
  T.n = n;
  for (int i = 0; i < n; ++i) {
    const RdrfLossTerm& t = terms[i];
    RDRF_CHECK(t.x && t.rows > 0 && t.cols > 0, -1, "loss_terms: term %d has no data", i);
    RDRF_CHECK(t.kind >= RDRF_LOSS_SQUARE && t.kind <= RDRF_LOSS_IDENTITY, -1, "loss_terms: term %d kind %d", i, t.kind);
    RDRF_CHECK(t.norm == RDRF_LOSS_NORM_MEAN || (t.norm == RDRF_LOSS_NORM_WEIGHT && t.w), -1,
               "loss_terms: term %d: the weight normaliser needs row weights", i);
    RDRF_CHECK(!t.gy || t.y, -1, "loss_terms: term %d: gy without y", i);
    T.t[i] = t;
  }
  return 0;
}

extern "C" size_t rdrf_loss_terms_workspace_floats(int n) { return (size_t)n * LOSS_BLOCKS * 2 + 2 * RDRF_MAX_LOSS_TERMS; }

extern "C" int rdrf_loss_terms_stats(const RdrfLossTerm* terms, int n, float* partial, float* stats, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  LossTermsK T;
  int rc = loss_pack(T, terms, n);
  if (rc) return rc;
  RDRF_CHECK(partial && stats, -1, "loss_terms_stats: null workspace / output");
  RDRF_LAUNCH("loss_terms", k_loss_fwd, dim3(LOSS_BLOCKS, n), dim3(256), stream, T, partial);
  RDRF_LAUNCH("loss_terms", k_loss_stats, dim3(1), dim3(64), stream, T, (const float*)partial, stats);
  return 0;
}

extern "C" int rdrf_loss_terms_finish(const RdrfLossTerm* terms, int n, const float* stats_local, const float* stats_global,
                                      int world, float* out, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  LossTermsK T;
  int rc = loss_pack(T, terms, n);
  if (rc) return rc;
  RDRF_CHECK(stats_local && stats_global && out && world >= 1, -1, "loss_terms_finish: bad arguments");
  RDRF_LAUNCH("loss_terms", k_loss_finish, dim3(1), dim3(64), stream, T, stats_local, stats_global, (float)world, out);
  return 0;
}

extern "C" int rdrf_loss_terms_fwd(const RdrfLossTerm* terms, int n, float* partial, float* out, rdrf_stream_t stream_) {
  RDRF_CHECK(partial && out, -1, "loss_terms_fwd: null workspace / output");
  float* stats = partial + (size_t)n * LOSS_BLOCKS * 2;   // tail of the workspace
  int rc = rdrf_loss_terms_stats(terms, n, partial, stats, stream_);
  if (rc) return rc;
  return rdrf_loss_terms_finish(terms, n, stats, stats, 1, out, stream_);
}

extern "C" int rdrf_loss_terms_bwd(const RdrfLossTerm* terms, int n, const float* out, const float* g_loss,
                                   rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  LossTermsK T;
  int rc = loss_pack(T, terms, n);
  if (rc) return rc;
  RDRF_CHECK(out && g_loss, -1, "loss_terms_bwd: null arguments");
  RDRF_LAUNCH("loss_terms_bwd", k_loss_bwd, dim3(LOSS_BLOCKS, n), dim3(256), stream, T, out, g_loss);
  return 0;
}


// ------------------------------------------------------------------------------------------------------------------
// per-frame median-normalised depth loss (train.py:797-807, 1636-1664, 2097-2121): one workgroup per frame
// ------------------------------------------------------------------------------------------------------------------
#define FDL_THREADS 1024
#define FDL_MAXN 32768

RDRF_D float block_sum(float v, float* red /*[16]*/) {   // deterministic tree; every thread gets the total
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < FDL_THREADS / 64; ++i) t += red[i];
  return t;
}

// bitonic sort of buf[0..np2) (np2 a power of two, padding = +inf) by the whole block
RDRF_D void bitonic_sort_lds(float* buf, int np2) {
  for (int k = 2; k <= np2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = threadIdx.x; i < np2; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const float a = buf[i], b = buf[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { buf[i] = b; buf[l] = a; }
        }
      }
    }
  }
  __syncthreads();
}

struct FdlStats { float med, inv; };   // median, 1 / (mean|x - med| + 1e-10)

RDRF_D bool fdl_sel(const int64_t* __restrict__ frame, const uint8_t* __restrict__ mask, int i, int N, int k) {
  return i < N && frame[i] == (int64_t)k && (mask == nullptr || mask[i] != 0);
}

// median (lower middle element of the sorted values) and mean absolute deviation of the n values x[i], i selected,
// which the caller has compacted into sortbuf[0..n) (any order)
RDRF_D FdlStats fdl_stats(const float* __restrict__ x, const int64_t* __restrict__ frame, const uint8_t* __restrict__ mask,
                          int N, int k, int n, float* sortbuf, float* red) {
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  for (int i = n + threadIdx.x; i < np2; i += blockDim.x) sortbuf[i] = INFINITY;
  bitonic_sort_lds(sortbuf, np2);
  const float med = sortbuf[(n - 1) >> 1];
  __syncthreads();
  float dev = 0.f;
  for (int i = threadIdx.x; i < N; i += blockDim.x)
    if (fdl_sel(frame, mask, i, N, k)) dev += fabsf(x[i] - med);
  const float s = block_sum(dev, red) / (float)n;
  FdlStats r;
  r.med = med;
  r.inv = 1.0f / (s + 1e-10f);
  return r;
}

// compaction of the selected values of x into sortbuf, in ray order (ballot ranks: deterministic); returns the count
RDRF_D int fdl_compact(const float* __restrict__ x, const int64_t* __restrict__ frame, const uint8_t* __restrict__ mask, int N,
                       int k, float* sortbuf, int* wcount, int* base_s) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) *base_s = 0;
  __syncthreads();
  for (int i0 = 0; i0 < N; i0 += FDL_THREADS) {
    const int i = i0 + threadIdx.x;
    const bool sel = fdl_sel(frame, mask, i, N, k);
    const unsigned long long b = __ballot(sel);
    if (lane == 0) wcount[wave] = __popcll(b);
    __syncthreads();
    int off = *base_s;
    for (int w = 0; w < wave; ++w) off += wcount[w];
    if (sel) sortbuf[off + __popcll(b & ((1ull << lane) - 1ull))] = x[i];
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < FDL_THREADS / 64; ++w) t += wcount[w];
      *base_s += t;
    }
    __syncthreads();
  }
  return *base_s;
}

__global__ __launch_bounds__(FDL_THREADS) void k_frame_depth_loss(const float* __restrict__ pred, const float* __restrict__ gt,
                                                                   const int64_t* __restrict__ frame,
                                                                   const uint8_t* __restrict__ mask, int N,
                                                                   float* __restrict__ g_raw, float* __restrict__ part /*[T][2]*/) {
  extern __shared__ float fdl_lds[];
  float* sortbuf = fdl_lds;                       // [pow2(n)]
  __shared__ float red[FDL_THREADS / 64];
  __shared__ int wcount[FDL_THREADS / 64];
  __shared__ int base_s;
  const int k = blockIdx.x;
  const int n = fdl_compact(pred, frame, mask, N, k, sortbuf, wcount, &base_s);
  if (n <= 1) {   // train.py:1641 / 2103: frames with fewer than two rays are skipped
    for (int i = threadIdx.x; i < N; i += blockDim.x)
      if (fdl_sel(frame, mask, i, N, k)) g_raw[i] = 0.f;
    if (threadIdx.x == 0) { part[2 * k] = 0.f; part[2 * k + 1] = 0.f; }
    return;
  }
  const FdlStats sp = fdl_stats(pred, frame, mask, N, k, n, sortbuf, red);
  __syncthreads();
  fdl_compact(gt, frame, mask, N, k, sortbuf, wcount, &base_s);
  const FdlStats sg = fdl_stats(gt, frame, mask, N, k, n, sortbuf, red);
  // ---- loss and the three sums of the gradient:  A = sum a, B = sum a (p - m), Sg = sum sgn(p - m), c = #(p == m)
  float Ls = 0.f, A = 0.f, B = 0.f, Sg = 0.f, cnt = 0.f;
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    if (!fdl_sel(frame, mask, j, N, k)) continue;
    const float dp = pred[j] - sp.med;
    const float r = dp * sp.inv - (gt[j] - sg.med) * sg.inv;
    const float a = 2.0f * r;
    Ls += r * r;
    A += a;
    B += a * dp;
    Sg += dp > 0.f ? 1.0f : (dp < 0.f ? -1.0f : 0.f);
    cnt += dp == 0.f ? 1.0f : 0.f;
  }
  Ls = block_sum(Ls, red); A = block_sum(A, red); B = block_sum(B, red); Sg = block_sum(Sg, red); cnt = block_sum(cnt, red);
  // d u_i / d p_j = (delta_ij - e_j) inv - (p_i - m) inv^2 ds/dp_j,   ds/dp_j = (sgn(p_j - m) - e_j Sg) / n,   e_j = [p_j == m] / c
  const float kB = B * sp.inv * sp.inv / (float)n;
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    if (!fdl_sel(frame, mask, j, N, k)) continue;
    const float dp = pred[j] - sp.med;
    const float r = dp * sp.inv - (gt[j] - sg.med) * sg.inv;
    const float e = dp == 0.f ? 1.0f / cnt : 0.f;
    const float sgn = dp > 0.f ? 1.0f : (dp < 0.f ? -1.0f : 0.f);
    g_raw[j] = 2.0f * r * sp.inv - e * A * sp.inv - kB * (sgn - e * Sg);
  }
  if (threadIdx.x == 0) { part[2 * k] = Ls; part[2 * k + 1] = (float)n; }
}

// rays no frame selected (masked out) get a zero gradient; loss = coef * sum L_k / sum n_k
__global__ __launch_bounds__(256) void k_frame_depth_finish(const float* __restrict__ part, int T, float coef,
                                                            const uint8_t* __restrict__ mask, int N,
                                                            float* __restrict__ g_raw, float* __restrict__ out) {
  if (mask != nullptr)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
      if (mask[i] == 0) g_raw[i] = 0.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float L = 0.f, c = 0.f;
    for (int k = 0; k < T; ++k) { L += part[2 * k]; c += part[2 * k + 1]; }   // frame order, like the reference's loop
    out[0] = coef * L / c;
    out[1] = coef / c;
    out[2] = c;
  }
}

__global__ __launch_bounds__(256) void k_frame_depth_bwd(const float* __restrict__ g_raw, const float* __restrict__ out,
                                                         const float* __restrict__ g_loss, int N, float gscale,
                                                         float* __restrict__ g_pred) {
  const float s = g_loss[0] * out[1] * gscale;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) g_pred[i] = s * g_raw[i];
}

extern "C" size_t rdrf_frame_depth_loss_workspace_bytes(int N, int T) { return (size_t)(2 * (T > 0 ? T : 1)) * sizeof(float) + 256; }

extern "C" int rdrf_frame_depth_loss_fwd(const float* pred, const float* gt, const int64_t* frame, const uint8_t* mask,
                                         int N, int T, float coef, float* out, float* g_raw, void* ws, size_t ws_bytes,
                                         rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RDRF_CHECK(pred && gt && frame && out && g_raw && ws, -1, "frame_depth_loss: null argument");
  RDRF_CHECK(N >= 1 && N <= FDL_MAXN && T >= 1, -1, "frame_depth_loss: N must be in 1..%d (got %d), T >= 1 (got %d)", FDL_MAXN, N, T);
  RDRF_CHECK(ws_bytes >= rdrf_frame_depth_loss_workspace_bytes(N, T), -2, "frame_depth_loss: workspace too small");
  float* part = (float*)ws;
  size_t np2 = 1;
  while (np2 < (size_t)N) np2 <<= 1;
  const size_t lds = np2 * 4;
  if (lds > 48 * 1024) {   // beyond the default dynamic-LDS limit: raise it for this launch's need, on the current device
    int dev = 0, lds_max = 0;
    RDRF_HIP(hipGetDevice(&dev));
    RDRF_HIP(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    RDRF_CHECK(lds <= (size_t)lds_max, -1, "frame_depth_loss: %d rays need %zu bytes of LDS, the device has %d", N, lds, lds_max);
    RDRF_HIP(hipFuncSetAttribute((const void*)k_frame_depth_loss, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  // rays whose frame id lies outside [0, T) are selected by no workgroup: their gradient is 0, not uninitialised memory
  RDRF_FILL(g_raw, 0, (size_t)N * sizeof(float), stream);
  rdrf_prof_begin("frame_depth_loss", stream);
  hipLaunchKernelGGL(k_frame_depth_loss, dim3(T), dim3(FDL_THREADS), lds, stream, pred, gt, frame, mask, N, g_raw, part);
  hipLaunchKernelGGL(k_frame_depth_finish, dim3(mask ? (N + 255) / 256 : 1), dim3(256), 0, stream, (const float*)part, T, coef,
                     mask, N, g_raw, out);
  rdrf_prof_end("frame_depth_loss", stream);
  RDRF_HIP(hipGetLastError());
  return 0;
}

extern "C" int rdrf_frame_depth_loss_bwd(const float* g_raw, const float* out, const float* g_loss, int N, float gscale,
                                         float* g_pred, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RDRF_CHECK(g_raw && out && g_loss && g_pred && N >= 1, -1, "frame_depth_loss_bwd: bad arguments");
  RDRF_LAUNCH("frame_depth_loss_bwd", k_frame_depth_bwd, dim3((N + 255) / 256), dim3(256), stream, g_raw, out, g_loss, N, gscale, g_pred);
  return 0;
}

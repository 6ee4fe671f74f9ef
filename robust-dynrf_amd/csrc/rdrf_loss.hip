// rdrf_loss.hip -- the per-ray / per-sample loss terms of one training iteration as ONE reduction launch
// (+ a finishing launch) forward and ONE launch backward.  The reference writes every term as a chain of
// elementwise torch ops ( ((a - b) ** 2).mean(), masked means, (|sf| * w).mean(), ... : train.py:1262-1296,
// 1330-1371, 1391-1413, 1476-1528, 1786-1839 ); with the ray path in kernels those chains were ~250 of the
// ~550 launches of an iteration, each a few microseconds of launch latency on a few KB of data.
//
// A term is   coef * sum_rows w[row] * sum_cols rho(x + ysign * y) / Z
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

// out[0] = total loss; out[1 + k] = coef_k / Z_k (the backward's per-term factor); out[1 + n + k] = term k's value
__global__ __launch_bounds__(64) void k_loss_finish(LossTermsK T, const float* __restrict__ partial, float* __restrict__ out) {
  const int k = threadIdx.x;
  float val = 0.f;
  if (k < T.n) {
    float s = 0.f, ws = 0.f;
    for (int b = 0; b < LOSS_BLOCKS; ++b) { s += partial[((size_t)k * LOSS_BLOCKS + b) * 2]; ws += partial[((size_t)k * LOSS_BLOCKS + b) * 2 + 1]; }
    const RdrfLossTerm& t = T.t[k];
    const float z = t.norm == RDRF_LOSS_NORM_WEIGHT ? ws + 1e-8f : (float)(t.rows * t.cols);
    const float scale = t.coef / z;
    val = s * scale;
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

extern "C" size_t rdrf_loss_terms_workspace_floats(int n) { return (size_t)n * LOSS_BLOCKS * 2; }

extern "C" int rdrf_loss_terms_fwd(const RdrfLossTerm* terms, int n, float* partial, float* out, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  LossTermsK T;
  int rc = loss_pack(T, terms, n);
  if (rc) return rc;
  RDRF_CHECK(partial && out, -1, "loss_terms_fwd: null workspace / output");
  RDRF_LAUNCH("loss_terms", k_loss_fwd, dim3(LOSS_BLOCKS, n), dim3(256), stream, T, partial);
  RDRF_LAUNCH("loss_terms", k_loss_finish, dim3(1), dim3(64), stream, T, (const float*)partial, out);
  return 0;
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

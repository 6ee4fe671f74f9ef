// rdrf_optim.hip -- factor-space kernels that run once per training iteration next to the ray path
// (SURVEY.md section 8f rank 3): the Adam update over the flat parameter buffers, the bilinear
// upsampling of the VM factors, and the dense L1 regulariser of the density / blending factors.
// All three are HBM-streaming kernels (a few bytes of arithmetic per byte moved).
#include "rdrf_host.hpp"

// ------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam as train.py:924-934 builds it: betas (0.9, 0.99), eps 1e-8, no weight decay,
// no amsgrad) over ONE flat fp32 range.  p, g, m, v are views of flat buffers (fields.TensorBase keeps
// every parameter and every gradient of a field as views of one flat buffer each), so one launch
// replaces torch's multi-tensor launches, and a rank of the sharded data-parallel step updates just the
// slice it owns.  The learning rate is piecewise constant: elements [0, split) are the VM factors
// (lr_init), [split, n) the networks (lr_basis) -- get_optparam_groups, models/tensoRF.py:49-61, 352-376.
// 28 bytes of HBM traffic per element (read p, g, m, v; write p, m, v).
// ------------------------------------------------------------------------------------------------
struct AdamCfg {
  float lr0, lr1, beta1, beta2, eps, grad_scale;
  float inv_bc1, inv_sqrt_bc2;   // 1 / (1 - beta1^t), 1 / sqrt(1 - beta2^t)
};

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v, size_t n4,
                                              size_t split4, AdamCfg c) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float lr = i < split4 ? c.lr0 : c.lr1;
    const float step_size = lr * c.inv_bc1;
    f32x4 pp = ((const f32x4*)p)[i], gg = ((const f32x4*)g)[i], mm = ((const f32x4*)m)[i], vv = ((const f32x4*)v)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = gg[k] * c.grad_scale;
      const float mk = mm[k] + (gr - mm[k]) * (1.0f - c.beta1);          // exp_avg.lerp_(grad, 1 - beta1)
      const float vk = vv[k] * c.beta2 + (1.0f - c.beta2) * gr * gr;      // exp_avg_sq.mul_().addcmul_()
      const float denom = sqrtf(vk) * c.inv_sqrt_bc2 + c.eps;
      pp[k] = pp[k] - step_size * (mk / denom);
      mm[k] = mk;
      vv[k] = vk;
    }
    ((f32x4*)p)[i] = pp;
    ((f32x4*)m)[i] = mm;
    ((f32x4*)v)[i] = vv;
  }
}

extern "C" int rdrf_adam_step(float* p, const float* g, float* m, float* v, size_t n, size_t split, float lr0,
                              float lr1, float beta1, float beta2, float eps, int step, float grad_scale,
                              rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RDRF_CHECK(p && g && m && v && n > 0 && step >= 1, -1, "adam_step: bad arguments");
  RDRF_CHECK(n % 4 == 0 && split % 4 == 0 && split <= n, -1, "adam_step: n and split must be multiples of 4");
  RDRF_CHECK((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, -1,
             "adam_step: buffers must be 16-byte aligned");
  AdamCfg c;
  c.lr0 = lr0; c.lr1 = lr1; c.beta1 = beta1; c.beta2 = beta2; c.eps = eps; c.grad_scale = grad_scale;
  c.inv_bc1 = (float)(1.0 / (1.0 - pow((double)beta1, (double)step)));
  c.inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
  const size_t n4 = n / 4;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;   // 8 workgroups per CU, grid-stride for the rest
  RDRF_LAUNCH("adam", k_adam, dim3((unsigned)blocks), dim3(256), stream, p, g, m, v, n4, split / 4, c);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// upsample_volume_grid (models/tensoRF.py:199-232, 814-850): F.interpolate(mode="bilinear",
// align_corners=True) of every plane (1,C,H,W) -> (1,C,H',W') and line (1,C,L,1) -> (1,C,L',1), on the
// channel-last storage, all tensors of a field in one launch.  Follows ATen's upsample_bilinear2d:
// scale = (in - 1) / (out - 1) (0 when out == 1), src = scale * dst, i0 = min(floor(src), in - 1),
// lambda = clamp(src - i0, 0, 1), i1 = i0 + (i0 < in - 1).
// ------------------------------------------------------------------------------------------------
struct UpJobs {
  RdrfTensor4 src[RDRF_TV_MAX], dst[RDRF_TV_MAX];
  int n;
};

__global__ __launch_bounds__(256) void k_upsample(UpJobs J) {
  const RdrfTensor4 s = J.src[blockIdx.y], d = J.dst[blockIdx.y];
  const int cq = d.C >> 2;   // quads of 4 components (C is 16, 4, 48 or 12)
  const long total = (long)d.H * d.W * cq;
  const float sh = d.H > 1 ? (float)(s.H - 1) / (float)(d.H - 1) : 0.f;
  const float sw = d.W > 1 ? (float)(s.W - 1) / (float)(d.W - 1) : 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % cq);
    const long r = i / cq;
    const int x = (int)(r % d.W), y = (int)(r / d.W);
    const float fy = sh * (float)y, fx = sw * (float)x;
    const int y0 = min((int)floorf(fy), s.H - 1), x0 = min((int)floorf(fx), s.W - 1);
    const float ly = fminf(fmaxf(fy - (float)y0, 0.f), 1.f), lx = fminf(fmaxf(fx - (float)x0, 0.f), 1.f);
    const int y1 = y0 + (y0 < s.H - 1 ? 1 : 0), x1 = x0 + (x0 < s.W - 1 ? 1 : 0);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long c = (long)(4 * q + k) * s.sC;
      const float a = s.x[c + y0 * s.sH + x0 * s.sW], b = s.x[c + y0 * s.sH + x1 * s.sW];
      const float e = s.x[c + y1 * s.sH + x0 * s.sW], f = s.x[c + y1 * s.sH + x1 * s.sW];
      o[k] = (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * e + lx * f);
    }
    float* out = const_cast<float*>(d.x);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[(long)(4 * q + k) * d.sC + (long)y * d.sH + (long)x * d.sW] = o[k];
  }
}

extern "C" int rdrf_upsample_bilinear(const RdrfTensor4* src, const RdrfTensor4* dst, int n,
                                      rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RDRF_CHECK(src && dst && n > 0 && n <= RDRF_TV_MAX, -1, "upsample_bilinear: 1..%d tensors per call", RDRF_TV_MAX);
  UpJobs J;
  J.n = n;
  long mx = 0;
  for (int i = 0; i < n; ++i) {
    RDRF_CHECK(src[i].x && dst[i].x && src[i].C == dst[i].C && src[i].C % 4 == 0 && src[i].H > 0 && src[i].W > 0 &&
                   dst[i].H > 0 && dst[i].W > 0, -1, "upsample_bilinear: bad tensor %d", i);
    J.src[i] = src[i];
    J.dst[i] = dst[i];
    const long t = (long)dst[i].H * dst[i].W * (dst[i].C / 4);
    mx = t > mx ? t : mx;
  }
  long bx = (mx + 255) / 256;
  bx = bx < 1 ? 1 : (bx > 1024 ? 1024 : bx);
  RDRF_LAUNCH("upsample", k_upsample, dim3((unsigned)bx, n), dim3(256), stream, J);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// density_L1 / blending_L1 (models/tensoRF.py:80-98, 378-416): mean over the X x Y x Z grid of
// |feature2density(f)|,  f[x,y,z] = sum_c P0[c,y,x] L0[c,z] + sum_c P1[c,z,x] L1[c,y] + sum_c P2[c,z,y] L2[c,x].
// The reference materialises the 24-component products ([1,24,X,Y,Z]: 1.6 GB at 256^3); here no volume
// exists at all: a thread owns one texel of a plane and walks the third axis, the 24-term dot product
// is formed in registers, and in the backward the plane gradient of that texel accumulates in registers
// (written once, no atomics).  Three sweeps (one per plane orientation) recompute f: 3 x 48 FLOP per
// voxel, the factors stay in L2.
//   sweep A: thread (x,y), loop z  -> value sum;  bwd: dP0[y,x,:], dL1[y,:], dL2[x,:]
//   sweep B: thread (x,z), loop y  ->             bwd: dP1[z,x,:], dL0[z,:]
//   sweep C: thread (y,z), loop x  ->             bwd: dP2[z,y,:]
// Line gradients: per-thread partial sums over the loop axis, one atomic per (thread, component).
// ------------------------------------------------------------------------------------------------
struct L1Args {
  RdrfVM vm, gvm;
  int act;
  float shift;
  const float* g;   // device scalar: d(loss) / d(mean)   (bwd)
  float* out;       // device scalar: sum of |act(f)|      (fwd)
  float inv_nv;
};

RDRF_D float l1_act(float f, int act, float shift) { return act == RDRF_ACT_RELU ? fmaxf(f, 0.f) : softplusf_(f + shift); }
RDRF_D float l1_dact(float f, int act, float shift) {   // d |act(f)| / df
  return act == RDRF_ACT_RELU ? (f > 0.f ? 1.f : 0.f) : sigmoidf_(f + shift);
}
RDRF_D float dot16(const float* a, const float* b) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) s = fmaf(a[c], b[c], s);
  return s;
}
RDRF_D float dot4p(const float* a, const float* b) { return fmaf(a[3], b[3], fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]))); }
RDRF_D void ldn(float* dst, const float* src, int n) {
  for (int c = 0; c < n; c += 4) {
    const f32x4 v = ld4(src + c);
    dst[c] = v.x; dst[c + 1] = v.y; dst[c + 2] = v.z; dst[c + 3] = v.w;
  }
}

// SWEEP 0: thread (x, y) loops z;  1: thread (x, z) loops y;  2: thread (y, z) loops x.  BWD = 0: value only.
template <int SWEEP, int BWD>
__global__ __launch_bounds__(256) void k_dense_l1(L1Args a) {
  const RdrfVM& vm = a.vm;
  const int X = vm.W[0], Y = vm.H[0], Z = vm.L[0];
  // thread -> (u fastest, v): sweep 0 (x, y), sweep 1 (x, z), sweep 2 (y, z)
  const int U = SWEEP == 2 ? Y : X, V = SWEEP == 0 ? Y : Z, Wn = SWEEP == 0 ? Z : (SWEEP == 1 ? Y : X);
  const int u = blockIdx.x * 32 + (threadIdx.x & 31), v = blockIdx.y * 8 + (threadIdx.x >> 5);
  const bool live = u < U && v < V;
  const int uc = live ? u : 0, vc = live ? v : 0;
  const float gscale = BWD ? a.g[0] * a.inv_nv : 0.f;
  float acc_val = 0.f;
  // loop-invariant factors of this thread
  float pf[16], l_a[4], l_b[4];        // sweep 0: P0[y,x,:16], L1[y,:4], L2[x,:4]
  float gp[16];                        // gradient of the owned plane texel (16 or 4 components used)
  float gl_a[16], gl_b[4];             // line-gradient partials
#pragma unroll
  for (int c = 0; c < 16; ++c) { gp[c] = 0.f; gl_a[c] = 0.f; }
#pragma unroll
  for (int c = 0; c < 4; ++c) gl_b[c] = 0.f;
  if (SWEEP == 0) {
    ldn(pf, vm.plane[0] + (size_t)vc * vm.sH[0] + (size_t)uc * vm.sW[0], 16);
    ldn(l_a, vm.line[1] + (size_t)vc * 4, 4);
    ldn(l_b, vm.line[2] + (size_t)uc * 4, 4);
  } else if (SWEEP == 1) {   // owns P1[z,x,:4]; invariant: P1 texel, L0[z,:16], L2[x,:4]
    ldn(l_a, vm.plane[1] + (size_t)vc * vm.sH[1] + (size_t)uc * vm.sW[1], 4);   // P1[z,x]
    ldn(pf, vm.line[0] + (size_t)vc * 16, 16);                                  // L0[z]
    ldn(l_b, vm.line[2] + (size_t)uc * 4, 4);                                   // L2[x]
  } else {                   // owns P2[z,y,:4]; invariant: P2 texel, L0[z,:16], L1[y,:4]
    ldn(l_a, vm.plane[2] + (size_t)vc * vm.sH[2] + (size_t)uc * vm.sW[2], 4);   // P2[z,y]
    ldn(pf, vm.line[0] + (size_t)vc * 16, 16);                                  // L0[z]
    ldn(l_b, vm.line[1] + (size_t)uc * 4, 4);                                   // L1[y]
  }
  for (int w = 0; w < Wn; ++w) {
    float f, q16[16], q4a[4], q4b[4];
    if (SWEEP == 0) {          // w = z: L0[z], P1[z,x], P2[z,y]
      ldn(q16, vm.line[0] + (size_t)w * 16, 16);
      ldn(q4a, vm.plane[1] + (size_t)w * vm.sH[1] + (size_t)uc * vm.sW[1], 4);
      ldn(q4b, vm.plane[2] + (size_t)w * vm.sH[2] + (size_t)vc * vm.sW[2], 4);
      f = dot16(pf, q16) + dot4p(q4a, l_a) + dot4p(q4b, l_b);
    } else if (SWEEP == 1) {   // w = y: P0[y,x], L1[y], P2[z,y]
      ldn(q16, vm.plane[0] + (size_t)w * vm.sH[0] + (size_t)uc * vm.sW[0], 16);
      ldn(q4a, vm.line[1] + (size_t)w * 4, 4);
      ldn(q4b, vm.plane[2] + (size_t)vc * vm.sH[2] + (size_t)w * vm.sW[2], 4);
      f = dot16(q16, pf) + dot4p(l_a, q4a) + dot4p(q4b, l_b);
    } else {                   // w = x: P0[y,x], P1[z,x], L2[x]
      ldn(q16, vm.plane[0] + (size_t)uc * vm.sH[0] + (size_t)w * vm.sW[0], 16);
      ldn(q4a, vm.plane[1] + (size_t)vc * vm.sH[1] + (size_t)w * vm.sW[1], 4);
      ldn(q4b, vm.line[2] + (size_t)w * 4, 4);
      f = dot16(q16, pf) + dot4p(q4a, l_b) + dot4p(l_a, q4b);
    }
    if (!BWD) {
      acc_val += fabsf(l1_act(f, a.act, a.shift));
    } else {
      const float s = live ? gscale * l1_dact(f, a.act, a.shift) : 0.f;
      if (SWEEP == 0) {        // dP0[y,x,c] += s L0[z,c]; dL1[y,c] += s P1[z,x,c]; dL2[x,c] += s P2[z,y,c]
#pragma unroll
        for (int c = 0; c < 16; ++c) gp[c] = fmaf(s, q16[c], gp[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) { gl_a[c] = fmaf(s, q4a[c], gl_a[c]); gl_b[c] = fmaf(s, q4b[c], gl_b[c]); }
      } else if (SWEEP == 1) { // dP1[z,x,c] += s L1[y,c]; dL0[z,c] += s P0[y,x,c]
#pragma unroll
        for (int c = 0; c < 4; ++c) gp[c] = fmaf(s, q4a[c], gp[c]);
#pragma unroll
        for (int c = 0; c < 16; ++c) gl_a[c] = fmaf(s, q16[c], gl_a[c]);
      } else {                 // dP2[z,y,c] += s L2[x,c]
#pragma unroll
        for (int c = 0; c < 4; ++c) gp[c] = fmaf(s, q4b[c], gp[c]);
      }
    }
  }
  if (!BWD) {
    if (!live) acc_val = 0.f;
    acc_val = wave_sum(acc_val);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc_val;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(a.out, part[0] + part[1] + part[2] + part[3]);
    return;
  }
  // ---- backward write-out: the owned plane texel (+=, no atomics: one thread per texel and sweep)
  if (live) {
    if (SWEEP == 0) {
      float* d = a.gvm.plane[0] + (size_t)v * a.gvm.sH[0] + (size_t)u * a.gvm.sW[0];
#pragma unroll
      for (int c = 0; c < 16; ++c) d[c] += gp[c];
    } else {
      const int pi = SWEEP;
      float* d = a.gvm.plane[pi] + (size_t)v * a.gvm.sH[pi] + (size_t)u * a.gvm.sW[pi];
#pragma unroll
      for (int c = 0; c < 4; ++c) d[c] += gp[c];
    }
  }
  // ---- line gradients: reduce over the 32 lanes that share v (same y for sweep 0 / same z for sweep
  // 1), then one atomic per (row of the block, component); the other line of sweep 0 (L2[x], indexed
  // by u) is reduced over the 8 rows of the block through LDS
  if (SWEEP == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float r = gl_a[c];
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) r += __shfl_xor(r, d, 32);
      if ((threadIdx.x & 31) == 0 && v < V) grad_add(a.gvm.line[1] + (size_t)v * 4 + c, r);
    }
    __shared__ float cols[8][32][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) cols[threadIdx.x >> 5][threadIdx.x & 31][c] = gl_b[c];
    __syncthreads();
    if (threadIdx.x < 128) {
      const int uu = threadIdx.x >> 2, c = threadIdx.x & 3;
      float r = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) r += cols[k][uu][c];
      const int ug = blockIdx.x * 32 + uu;
      if (ug < U) grad_add(a.gvm.line[2] + (size_t)ug * 4 + c, r);
    }
  } else if (SWEEP == 1) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      float r = gl_a[c];
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) r += __shfl_xor(r, d, 32);
      if ((threadIdx.x & 31) == 0 && v < V) grad_add(a.gvm.line[0] + (size_t)v * 16 + c, r);
    }
  }
}

static bool l1_vm_ok(const RdrfVM* vm) {
  return vm && vm_ok(*vm, 16, 4) && vm->H[0] == vm->L[1] && vm->W[0] == vm->L[2] && vm->H[1] == vm->L[0] &&
         vm->H[2] == vm->L[0] && vm->W[1] == vm->W[0] && vm->W[2] == vm->H[0];
}

extern "C" int rdrf_dense_l1_fwd(const RdrfVM* vm, int act, float density_shift, float* sum_out,
                                 rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RDRF_CHECK(l1_vm_ok(vm) && sum_out, -1, "dense_l1_fwd: expects a {16,4,4}-component factor set of one grid");
  L1Args a;
  memset(&a, 0, sizeof(a));
  a.vm = *vm; a.act = act; a.shift = density_shift; a.out = sum_out;
  RDRF_FILL(sum_out, 0, sizeof(float), stream);
  const int X = vm->W[0], Y = vm->H[0];
  RDRF_LAUNCH("dense_l1", (k_dense_l1<0, 0>), dim3((X + 31) / 32, (Y + 7) / 8), dim3(256), stream, a);
  return 0;
}

extern "C" int rdrf_dense_l1_bwd(const RdrfVM* vm, const RdrfVM* gvm, int act, float density_shift,
                                 const float* g_mean, rdrf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RDRF_CHECK(l1_vm_ok(vm) && gvm && g_mean, -1, "dense_l1_bwd: bad arguments");
  L1Args a;
  memset(&a, 0, sizeof(a));
  a.vm = *vm; a.gvm = *gvm; a.act = act; a.shift = density_shift; a.g = g_mean;
  const int X = vm->W[0], Y = vm->H[0], Z = vm->L[0];
  a.inv_nv = (float)(1.0 / ((double)X * Y * Z));
  RDRF_LAUNCH("dense_l1_bwd", (k_dense_l1<0, 1>), dim3((X + 31) / 32, (Y + 7) / 8), dim3(256), stream, a);
  RDRF_LAUNCH("dense_l1_bwd", (k_dense_l1<1, 1>), dim3((X + 31) / 32, (Z + 7) / 8), dim3(256), stream, a);
  RDRF_LAUNCH("dense_l1_bwd", (k_dense_l1<2, 1>), dim3((Y + 31) / 32, (Z + 7) / 8), dim3(256), stream, a);
  return 0;
}


#ifdef RDRF_DETERMINISTIC
int det_bind_optim(int slot, const float* base, size_t n, unsigned long long* shadow, hipStream_t stream) {
  static DetMap host[2];
  host[slot].base = base; host[slot].n = n; host[slot].shadow = shadow;
  RDRF_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_det), &host[slot], sizeof(DetMap), slot * sizeof(DetMap), hipMemcpyHostToDevice, stream));
  return 0;
}
#endif

// microbenchmark: fp32 global atomics, same total count, different lane->address patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// A: every lane owns a random texel, 4 instructions (one per component)
__global__ void kA(float* buf, unsigned ntex, int iters) {
  unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    unsigned tex = hash(t * 977u + i) % ntex;
    float* p = buf + (size_t)tex * 4;
    atomicAdd(p + 0, 1.f); atomicAdd(p + 1, 1.f); atomicAdd(p + 2, 1.f); atomicAdd(p + 3, 1.f);
  }
}
// B: 4 adjacent lanes share a texel (lane&3 = component), 4 instructions cover 4x more texels each
__global__ void kB(float* buf, unsigned ntex, int iters) {
  unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    for (int k = 0; k < 4; ++k) {
      unsigned tex = hash(((t >> 2) * 4 + k) * 977u + i) % ntex;
      atomicAdd(buf + (size_t)tex * 4 + (t & 3), 1.f);
    }
  }
}
// C: like A but 16 components per texel row (64B), lanes 0..15 of a 16-group cover one texel
__global__ void kC(float* buf, unsigned ntex, int iters) {
  unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    for (int k = 0; k < 4; ++k) {
      unsigned tex = hash(((t >> 4) * 4 + k) * 977u + i) % (ntex / 4);
      atomicAdd(buf + (size_t)tex * 16 + (t & 15), 1.f);
    }
  }
}
// E: like B but only every 4th quad is live (16 of 64 lanes active): same instruction count as B,
// 4x fewer requests.  E ~ B  => bound by atomic INSTRUCTION issue; E ~ B/4 => bound by requests.
__global__ void kE(float* buf, unsigned ntex, int iters) {
  unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    for (int k = 0; k < 4; ++k) {
      unsigned tex = hash(((t >> 2) * 4 + k) * 977u + i) % ntex;
      if (((t >> 2) & 3) == 0) atomicAdd(buf + (size_t)tex * 4 + (t & 3), 1.f);
    }
  }
}
// F: like B with only ONE quad live per instruction (4 lanes)
__global__ void kF(float* buf, unsigned ntex, int iters) {
  unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    for (int k = 0; k < 4; ++k) {
      unsigned tex = hash(((t >> 2) * 4 + k) * 977u + i) % ntex;
      if (((t >> 2) & 15) == 0) atomicAdd(buf + (size_t)tex * 4 + (t & 3), 1.f);
    }
  }
}
// G: like B, but the 4-lane group at lanes l (half 0) and l+32 (half 1) hit two ADJACENT quads of the
// same 64-byte texel.  G ~ B/2 => the coalescer merges same-line lanes across the whole wave.
__global__ void kG(float* buf, unsigned ntex, int iters) {
  unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned lane = threadIdx.x & 63, hh = lane >> 5;
  unsigned tq = (t & ~63u) | (lane & 31);   // same id for lane and lane+32
  for (int i = 0; i < iters; ++i) {
    for (int k = 0; k < 4; ++k) {
      unsigned tex = hash(((tq >> 2) * 4 + k) * 977u + i) % (ntex / 4);
      atomicAdd(buf + (size_t)tex * 16 + hh * 4 + (t & 3), 1.f);
    }
  }
}
// H: like G but the two halves hit quads of DIFFERENT 64-byte texels (control: same address math)
__global__ void kH(float* buf, unsigned ntex, int iters) {
  unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned lane = threadIdx.x & 63, hh = lane >> 5;
  unsigned tq = (t & ~63u) | (lane & 31);
  for (int i = 0; i < iters; ++i) {
    for (int k = 0; k < 4; ++k) {
      unsigned tex = hash(((tq >> 2) * 4 + k) * 977u + i + hh * 7919u) % (ntex / 4);
      atomicAdd(buf + (size_t)tex * 16 + hh * 4 + (t & 3), 1.f);
    }
  }
}
// I: like B, but every XCD (workgroups are dealt round-robin: XCD = blockIdx.x % 8) updates its own
// eighth of the buffer.  I >> B  => cross-XCD sharing of lines (L2 coherence) limits the atomic rate.
__global__ void kI(float* buf, unsigned ntex, int iters) {
  unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned part = ntex / 8, base = (blockIdx.x & 7) * part;
  for (int i = 0; i < iters; ++i) {
    for (int k = 0; k < 4; ++k) {
      unsigned tex = base + hash(((t >> 2) * 4 + k) * 977u + i) % part;
      atomicAdd(buf + (size_t)tex * 4 + (t & 3), 1.f);
    }
  }
}
// D: A without atomics (plain read-modify-write, racy) as a bandwidth reference
__global__ void kD(float* buf, unsigned ntex, int iters) {
  unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    unsigned tex = hash(t * 977u + i) % ntex;
    float4* p = (float4*)(buf + (size_t)tex * 4);
    float4 v = *p; v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f; *p = v;
  }
}
int main() {
  const unsigned sizes[3] = {13254u, 200000u, 4000000u};  // texels (x4 floats): 212 KB, 3.2 MB, 64 MB
  for (int si = 0; si < 3; ++si) {
    unsigned ntex = sizes[si];
    float* buf; CHECK(hipMalloc(&buf, (size_t)ntex * 16)); CHECK(hipMemset(buf, 0, (size_t)ntex * 16));
    const int blocks = 2048, threads = 256, iters = 16;
    const double natom = (double)blocks * threads * iters * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 9; ++kind) {
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) kA<<<blocks, threads>>>(buf, ntex, iters);
        if (kind == 1) kB<<<blocks, threads>>>(buf, ntex, iters);
        if (kind == 2) kC<<<blocks, threads>>>(buf, ntex, iters);
        if (kind == 3) kD<<<blocks, threads>>>(buf, ntex, iters);
        if (kind == 4) kE<<<blocks, threads>>>(buf, ntex, iters);
        if (kind == 5) kF<<<blocks, threads>>>(buf, ntex, iters);
        if (kind == 6) kG<<<blocks, threads>>>(buf, ntex, iters);
        if (kind == 7) kH<<<blocks, threads>>>(buf, ntex, iters);
        if (kind == 8) kI<<<blocks, threads>>>(buf, ntex, iters);
        hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("ntex %8u  kernel %c  %.3f ms  %.1f G scalar-updates/s (if all lanes live)  %.2f G wave-instr/s\n", ntex, "ABCDEFGHI"[kind], best, natom / best / 1e6, natom / 64.0 / best / 1e6);
    }
    CHECK(hipFree(buf));
  }
  return 0;
}

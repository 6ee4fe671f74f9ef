// Which lane/address patterns of ONE fp32 atomic instruction does gfx950 merge into one request?
// Every wave issues `iters` atomic instructions; in each, lane l adds to base + pattern(l) where base is
// a random 256-byte-aligned block.  Time x 20.5 G requests/s (tools/micro/atomic_scope.hip) / instructions
// = requests per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// float offset of lane l inside a 1 KB block (256 floats)
__device__ inline int pattern(int P, int l) {
  const int q = l >> 2, c = l & 3;
  switch (P) {
    case 0: return l;                                   // 64 contiguous floats = 4 lines
    case 1: return q * 8 + c;                           // quads at 32-byte stride (8 lines of 64 B)
    case 2: return (15 - q) * 4 + c;                    // quads contiguous but DESCENDING
    case 3: return ((l + 4) & 63) ;                     // contiguous, rotated by one quad (wraps inside block)
    case 4: return q * 16 + c;                          // one quad per 64-byte line: 16 lines
    case 5: return (q ^ 1) * 4 + c;                     // contiguous lines, quads pair-swapped
    case 6: return ((q * 5) & 15) * 4 + c;              // same 4 lines, quads in scrambled order
    case 7: return (q >> 1) * 4 + c;                    // pairs of quads on the SAME texel (duplicates)
    case 8: return c * 16 + q;                          // lane-strided: 4 B at 64-byte... transposed
    case 9: return (l & 15) + 64 * (l >> 4);            // 4 rows of 64 B, each contiguous, 256 B apart
    default: return l;
  }
}

__global__ __launch_bounds__(256) void k_pat(float* __restrict__ buf, long blocks1k, int iters, int P, int active_mask_mode) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  unsigned long long st = wave * 0x9E3779B97F4A7C15ull + 4242;
  const int off = pattern(P, lane);
  const bool act = active_mask_mode == 0 ? true : (((lane >> 2) * 7 + 3) % 5 != 0);   // mode 1: ~20% of quads idle
  for (int i = 0; i < iters; ++i) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    const long b = (long)((st >> 20) % (unsigned long long)blocks1k);
    if (act) __hip_atomic_fetch_add(buf + b * 256 + off, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main() {
  const long blocks1k = 200000;   // 200 MB footprint / 1 KB
  float* buf;
  CK(hipMalloc(&buf, blocks1k * 1024));
  CK(hipMemset(buf, 0, blocks1k * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 1536, iters = 256;
  const double instr = (double)blocks * 4 * iters;
  const char* names[] = {"64 contiguous floats (4 lines)", "quads at 32 B stride (8 lines)", "quads contiguous, descending",
                         "contiguous rotated by a quad", "one quad per line (16 lines)", "quads pair-swapped",
                         "quads scrambled in 4 lines", "duplicate texels (pairs)", "lane-transposed (4B @64B x16, 4 cols)",
                         "4 rows x 64 B, 256 B apart"};
  for (int mode = 0; mode < 2; ++mode)
    for (int P = 0; P < 10; ++P) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        k_pat<<<blocks, 256>>>(buf, blocks1k, iters, P, mode);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("mask %d  P%d %-40s %.3f ms  %.2f ns/instr/chip  ~%.1f requests/instr (at 20.5 G/s)\n", mode, P, names[P], best,
             best * 1e6 / instr, best * 1e-3 * 20.5e9 / instr);
    }
  return 0;
}

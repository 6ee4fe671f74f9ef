// rdrf_sort.hip -- device-wide stable key sort for the sorted scatter (rdrf_bwd.hip), hand-written for gfx950.
//
// The keys are (plane | level-0 cell) codes of every live sample: 17-19 significant bits, a few million entries, the value
// of an entry is its position in the key array.  An LSD radix sort over digits of <= 9 bits (two passes for up to 18
// bits); one pass =
//   k_radix_hist     a workgroup owns a tile of 2048 consecutive entries; digit histogram of the tile in LDS, written
//                    digit-major ([digit][tile]) so that a row-wise exclusive scan orders equal digits by tile
//   k_radix_scan     one workgroup per digit: exclusive scan of its row + the digit's total
//   k_radix_scatter  the same tiles again: exclusive scan of the 512 digit totals in LDS, then a STABLE rank of every
//                    entry among the equal digits of its tile -- waves own contiguous quarters of the tile, a wave walks
//                    its quarter 64 entries at a time, equal digits inside a 64-entry round are ranked with nine ballots
//                    (the peers of a lane = the lanes whose digit agrees in every bit), per-(wave, digit) running offsets
//                    live in LDS -- and the scatter of key and value to offset + rank.
// No atomics with return values, no data-dependent loops: a pass moves 16 B per entry through HBM twice; at 1.4 M entries
// the sort is launch-latency bound (6 launches).  Equal keys keep ascending position order (the deterministic build and
// the run reduction of the scatter rely on it).
#include <cstring>
#include <hip/hip_runtime.h>

#include "rdrf_host.hpp"

namespace {
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
#ifndef RDRF_RS_TILE
#define RDRF_RS_TILE 2048   // 8192 -> 2048: sort -11 % at stage 0, -12 % at the final stage (profiles/r06_ab_sort_tile.txt)
#endif
constexpr int RS_TILE = RDRF_RS_TILE;          // entries per workgroup
constexpr int RS_PER_WAVE = RS_TILE / RS_WAVES;
constexpr int RS_ROUNDS = RS_PER_WAVE / 64;    // 64-entry rounds per wave
constexpr int RS_MAX_DIGIT_BITS = 9;
constexpr int RS_BINS = 1 << RS_MAX_DIGIT_BITS;

struct RadixArgs {
  const unsigned* keys_in;
  const unsigned* vals_in;   // nullptr: the value of an entry is its position
  unsigned* keys_out;
  unsigned* vals_out;        // nullptr: values are not wanted (key-only sort)
  unsigned* hist;            // [nbins][ntiles] digit-major
  unsigned* totals;          // [nbins]
  unsigned n;
  int ntiles, shift, nbins;
  const int* n_dev;          // device-side entry count: the sort covers min(n, n_mul * *n_dev) entries (nullptr: n).  The
  unsigned n_mul;            // launches are sized for n; tiles beyond the device count find nothing to do
};
__device__ __forceinline__ unsigned radix_n(const RadixArgs& a) {
  if (a.n_dev == nullptr) return a.n;
  const unsigned m = (unsigned)*a.n_dev * a.n_mul;
  return m < a.n ? m : a.n;
}

__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(RadixArgs a) {
  __shared__ unsigned h[RS_BINS];
  for (int i = threadIdx.x; i < a.nbins; i += RS_THREADS) h[i] = 0u;
  __syncthreads();
  const unsigned mask = (unsigned)a.nbins - 1u;
  const size_t base = (size_t)blockIdx.x * RS_TILE;
  const unsigned n = radix_n(a);
  if (base < n) {
    for (int i = threadIdx.x; i < RS_TILE; i += RS_THREADS) {
      const size_t p = base + i;
      if (p < n) atomicAdd(&h[(a.keys_in[p] >> a.shift) & mask], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < a.nbins; i += RS_THREADS) a.hist[(size_t)i * a.ntiles + blockIdx.x] = h[i];
}

// workgroup-wide exclusive scan of one value per thread (256 threads); returns the exclusive prefix, *total = the sum
__device__ __forceinline__ unsigned block_exscan(unsigned v, unsigned* wsum, unsigned* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  unsigned off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < RS_WAVES; ++w) {
    const unsigned s = wsum[w];
    if (w < wave) off += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return off + inc - v;
}

__global__ __launch_bounds__(RS_THREADS) void k_radix_scan(RadixArgs a) {
  __shared__ unsigned wsum[RS_WAVES];
  unsigned* row = a.hist + (size_t)blockIdx.x * a.ntiles;
  unsigned carry = 0;
  for (int i0 = 0; i0 < a.ntiles; i0 += RS_THREADS) {
    const int i = i0 + threadIdx.x;
    const unsigned v = i < a.ntiles ? row[i] : 0u;
    unsigned tot;
    const unsigned ex = block_exscan(v, wsum, &tot);
    if (i < a.ntiles) row[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) a.totals[blockIdx.x] = carry;
}

__global__ __launch_bounds__(RS_THREADS) void k_radix_scatter(RadixArgs a) {
  __shared__ unsigned offs[RS_WAVES][RS_BINS];   // running output offset of (wave, digit)
  __shared__ unsigned gbase[RS_BINS];
  __shared__ unsigned wsum[RS_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned mask = (unsigned)a.nbins - 1u;
  const unsigned n = radix_n(a);
  if ((size_t)blockIdx.x * RS_TILE >= n) return;   // (uniform: the whole tile lies beyond the device-side count)
  // ---- global base of every digit = exclusive scan of the totals + this tile's entry of the digit's row scan
  {
    unsigned carry = 0;
    for (int d0 = 0; d0 < a.nbins; d0 += RS_THREADS) {
      const int d = d0 + threadIdx.x;
      const unsigned v = d < a.nbins ? a.totals[d] : 0u;
      unsigned tot;
      const unsigned ex = block_exscan(v, wsum, &tot);
      if (d < a.nbins) gbase[d] = carry + ex + a.hist[(size_t)d * a.ntiles + blockIdx.x];
      carry += tot;
    }
  }
  for (int i = threadIdx.x; i < RS_WAVES * RS_BINS; i += RS_THREADS) (&offs[0][0])[i] = 0u;
  __syncthreads();
  // ---- per-wave digit counts of the wave's quarter (keys stay in registers for the second sweep)
  const size_t wbase = (size_t)blockIdx.x * RS_TILE + (size_t)wave * RS_PER_WAVE;
  unsigned key[RS_ROUNDS];
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const size_t p = wbase + (size_t)r * 64 + lane;
    key[r] = p < n ? a.keys_in[p] : 0xffffffffu;
    if (p < n) atomicAdd(&offs[wave][(key[r] >> a.shift) & mask], 1u);
  }
  __syncthreads();
  // ---- counts -> starting offsets: digit base + the counts of the lower waves
  for (int d = threadIdx.x; d < a.nbins; d += RS_THREADS) {
    unsigned run = gbase[d];
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) {
      const unsigned c = offs[w][d];
      offs[w][d] = run;
      run += c;
    }
  }
  __syncthreads();
  // ---- stable rank inside each 64-entry round, scatter
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll   // (key[] is a register array: a rolled loop would index it dynamically, i.e. through scratch)
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const size_t p = wbase + (size_t)r * 64 + lane;
    const bool act = p < n;
    const unsigned d = (key[r] >> a.shift) & mask;
    unsigned long long peers = __ballot(act);
#pragma unroll
    for (int b = 0; b < RS_MAX_DIGIT_BITS; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    if (act) {
      const unsigned base = offs[wave][d];
      const unsigned rank = (unsigned)__popcll(peers & lt);
      const unsigned pos = base + rank;
      a.keys_out[pos] = key[r];
      if (a.vals_out) a.vals_out[pos] = a.vals_in ? a.vals_in[p] : (unsigned)p;
      if ((peers >> lane) >> 1 == 0ull) offs[wave][d] = base + (unsigned)__popcll(peers);   // highest peer: next round's base
    }
  }
}

int radix_plan(int bits, int& passes, int& digit_bits) {
  if (bits < 1) bits = 1;
  passes = (bits + RS_MAX_DIGIT_BITS - 1) / RS_MAX_DIGIT_BITS;
  digit_bits = (bits + passes - 1) / passes;
  return passes;
}
}  // namespace

size_t rdrf_sort_temp_bytes(unsigned n, int bits) {
  int passes, db;
  radix_plan(bits, passes, db);
  const size_t ntiles = ((size_t)n + RS_TILE - 1) / RS_TILE;
  return 2 * (((size_t)n * 4 + 255) & ~(size_t)255) + (((size_t)RS_BINS * (ntiles ? ntiles : 1) * 4 + 255) & ~(size_t)255) + RS_BINS * 4 +
         1024;
}

static int radix_sort(const unsigned* keys_in, unsigned* keys_out, unsigned* vals_out, unsigned n, int bits, void* temp,
                      size_t temp_bytes, hipStream_t stream, const int* n_dev = nullptr, unsigned n_mul = 0) {
  if (n == 0) return 0;
  RDRF_CHECK(temp != nullptr && temp_bytes >= rdrf_sort_temp_bytes(n, bits), -3, "sort: temporary storage too small (%zu < %zu)",
             temp_bytes, rdrf_sort_temp_bytes(n, bits));
  int passes, db;
  radix_plan(bits, passes, db);
  WsCarver c(temp, temp_bytes);
  unsigned* tk = c.take<unsigned>(n);
  unsigned* tv = c.take<unsigned>(n);
  const int ntiles = (int)(((size_t)n + RS_TILE - 1) / RS_TILE);
  unsigned* hist = c.take<unsigned>((size_t)RS_BINS * ntiles);
  unsigned* totals = c.take<unsigned>(RS_BINS);
  RDRF_CHECK(c.ok(), -3, "sort: temporary storage too small");
  const unsigned* kin = keys_in;
  const unsigned* vin = nullptr;
  for (int p = 0; p < passes; ++p) {
    const bool to_out = ((passes - 1 - p) & 1) == 0;   // the last pass lands in the caller's arrays
    RadixArgs a;
    a.keys_in = kin; a.vals_in = vin;
    a.keys_out = to_out ? keys_out : tk;
    a.vals_out = vals_out ? (to_out ? vals_out : tv) : nullptr;
    a.hist = hist; a.totals = totals; a.n = n; a.ntiles = ntiles; a.n_dev = n_dev; a.n_mul = n_mul;
    a.shift = p * db;
    a.nbins = 1 << db;
    rdrf_prof_begin("sort", stream);
    hipLaunchKernelGGL(k_radix_hist, dim3(ntiles), dim3(RS_THREADS), 0, stream, a);
    hipLaunchKernelGGL(k_radix_scan, dim3(a.nbins), dim3(RS_THREADS), 0, stream, a);
    hipLaunchKernelGGL(k_radix_scatter, dim3(ntiles), dim3(RS_THREADS), 0, stream, a);
    rdrf_prof_end("sort", stream);
    RDRF_HIP(hipGetLastError());
    kin = a.keys_out;
    vin = a.vals_out;
  }
  return 0;
}

// keys_in [n] -> keys_out [n] ascending (stable), vals_out[i] = original position of keys_out[i]
// n_dev (optional): the first n_mul * *n_dev entries only (a count that lives on the device: the compacted appearance list)
int rdrf_sort_positions(const unsigned* keys_in, unsigned* keys_out, unsigned* vals_out, unsigned n, int bits, void* temp,
                        size_t temp_bytes, hipStream_t stream, const int* n_dev, unsigned n_mul) {
  RDRF_CHECK(keys_in && keys_out && vals_out, -1, "sort: null argument");
  return radix_sort(keys_in, keys_out, vals_out, n, bits, temp, temp_bytes, stream, n_dev, n_mul);
}

// deterministic build: ascending in-place sort of an int list (the app-mask compaction lists, whose append order depends on
// wave timing).  Scratch is a lazily grown device allocation owned by this debugging build.
int rdrf_sort_ints_inplace(int* data, unsigned n, hipStream_t stream) {
  static void* scratch = nullptr;
  static size_t scratch_bytes = 0;
  if (n == 0) return 0;
  const size_t need = rdrf_sort_temp_bytes(n, 32);
  const size_t total = need + (((size_t)n * 4 + 255) & ~(size_t)255) + 512;
  if (total > scratch_bytes) {
    if (scratch) RDRF_HIP(hipFree(scratch));
    RDRF_HIP(hipMalloc(&scratch, total));
    scratch_bytes = total;
  }
  unsigned* out = (unsigned*)scratch;
  void* tmp = (char*)scratch + (((size_t)n * 4 + 255) & ~(size_t)255);
  int rc = radix_sort((const unsigned*)data, out, nullptr, n, 32, tmp, need, stream);
  if (rc) return rc;
  RDRF_HIP(hipMemcpyAsync(data, out, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
  return 0;
}

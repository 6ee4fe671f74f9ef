// rdrf_sort.hip -- device-wide key sort for the sorted scatter (rdrf_bwd.hip): one stable LSD radix sort (rocPRIM's
// device primitive, the plain library routine for a plain library problem) of the (plane | cell) keys of every live
// sample; the values are the positions in the key array (a counting iterator: nothing is materialised), so equal
// keys keep ascending sample order -- the deterministic mode relies on that.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "rdrf_host.hpp"

size_t rdrf_sort_temp_bytes(unsigned n, int bits) {
  static thread_local unsigned last_n = 0;   // queried by every workspace-size call: remember the last answer
  static thread_local int last_bits = 0;
  static thread_local size_t last_need = 0;
  if (n == last_n && bits == last_bits && last_need != 0) return last_need;
  size_t need = 0;
  rocprim::counting_iterator<unsigned> vin(0);
  (void)rocprim::radix_sort_pairs(nullptr, need, (const unsigned*)nullptr, (unsigned*)nullptr, vin, (unsigned*)nullptr, n, 0,
                                  (unsigned)bits, (hipStream_t)0);
  last_n = n; last_bits = bits; last_need = need + 256;
  return need + 256;
}

// keys_in [n] -> keys_out [n] ascending (stable), vals_out[i] = original position of keys_out[i]
int rdrf_sort_positions(const unsigned* keys_in, unsigned* keys_out, unsigned* vals_out, unsigned n, int bits, void* temp,
                        size_t temp_bytes, hipStream_t stream) {
  size_t need = 0;
  rocprim::counting_iterator<unsigned> vin(0);
  RDRF_HIP(rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, vin, vals_out, n, 0, (unsigned)bits, stream));
  RDRF_CHECK(temp != nullptr && temp_bytes >= need, -3, "sort: temporary storage too small (%zu < %zu)", temp_bytes, need);
  RDRF_HIP(rocprim::radix_sort_pairs(temp, need, keys_in, keys_out, vin, vals_out, n, 0, (unsigned)bits, stream));
  return 0;
}


// deterministic build: ascending in-place sort of an int list (the app-mask compaction lists, whose append order depends on
// wave timing).  Scratch is a lazily grown device allocation owned by this debugging build.
int rdrf_sort_ints_inplace(int* data, unsigned n, hipStream_t stream) {
  static void* scratch = nullptr;
  static size_t scratch_bytes = 0;
  size_t need = 0;
  RDRF_HIP(rocprim::radix_sort_keys(nullptr, need, (const unsigned*)nullptr, (unsigned*)nullptr, n, 0, 32, stream));
  const size_t total = need + (size_t)n * 4 + 512;
  if (total > scratch_bytes) {
    if (scratch) RDRF_HIP(hipFree(scratch));
    RDRF_HIP(hipMalloc(&scratch, total));
    scratch_bytes = total;
  }
  unsigned* out = (unsigned*)scratch;
  void* tmp = (char*)scratch + (((size_t)n * 4 + 255) & ~(size_t)255);
  RDRF_HIP(rocprim::radix_sort_keys(tmp, need, (const unsigned*)data, out, n, 0, 32, stream));
  RDRF_HIP(hipMemcpyAsync(data, out, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
  return 0;
}

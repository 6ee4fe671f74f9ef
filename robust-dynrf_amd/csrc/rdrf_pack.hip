// rdrf_pack.hip -- weight packing into MFMA-fragment order, error string, profiling hooks.
//
// The parameters stay in the reference's state_dict layout (nn.Linear weight [out][in]); before a
// forward/backward launch sequence one kernel permutes every MLP weight into the order the MFMA
// A operand is read in: [out block][k-step/4][lane][4] with lane = (h<<5)|neuron and the k-step ->
// input-column map of rdrf_common.hpp (seg_imap o elem_of).  Each MFMA then needs ONE coalesced
// 16-byte load per lane for four k-steps.
#include <map>
#include <string>
#include <vector>

#include "rdrf_host.hpp"

// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void rdrf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* rdrf_last_error(void) { return g_err; }
extern "C" int rdrf_abi_version(void) { return RDRF_ABI_VERSION; }

// ------------------------------------------------------------------------------------------------
// byte fill as a KERNEL.  hipMemsetAsync is a memset node in a captured HIP graph, and replays of such nodes were observed
// not to clear the static field's 256-byte list counter (tools/graph/probe.py: rdrf_static_fwd replayed -> the list
// overruns its buffer on the second replay); a kernel node replays like every other launch of the sequence.
__global__ __launch_bounds__(256) void k_fill_bytes(unsigned char* __restrict__ p, unsigned value, size_t bytes) {
  const unsigned w = value * 0x01010101u;
  const size_t head = ((16 - ((size_t)p & 15)) & 15) < bytes ? ((16 - ((size_t)p & 15)) & 15) : bytes;   // to 16-byte alignment
  const size_t nvec = (bytes - head) / 16;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  uint4* v = (uint4*)(p + head);
  for (size_t i = tid; i < nvec; i += nth) v[i] = make_uint4(w, w, w, w);
  if (tid < head) p[tid] = (unsigned char)value;
  const size_t tail0 = head + nvec * 16;
  if (tail0 + tid < bytes && tid < 16) p[tail0 + tid] = (unsigned char)value;
}
int rdrf_fill_async(void* p, int byte_value, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return 0;
  RDRF_CHECK(p != nullptr, -1, "fill: null pointer");
  size_t blocks = (bytes / 16 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(k_fill_bytes, dim3((unsigned)blocks), dim3(256), 0, stream, (unsigned char*)p, (unsigned)(byte_value & 0xff), bytes);
  RDRF_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
struct ProfRec {
  std::string name;
  hipEvent_t a, b;
  bool closed;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::map<std::string, std::pair<double, int>> g_prof_acc;

void rdrf_prof_begin(const char* name, hipStream_t s) {
  if (!g_prof_on) return;
  ProfRec r;
  r.name = name;
  (void)hipEventCreate(&r.a);
  (void)hipEventCreate(&r.b);
  r.closed = false;
  (void)hipEventRecord(r.a, s);
  g_prof.push_back(r);
}
void rdrf_prof_end(const char* name, hipStream_t s) {
  if (!g_prof_on) return;
  // the innermost open record of this name (ranges may nest: a launch sequence and the kernels inside it)
  for (size_t i = g_prof.size(); i-- > 0;)
    if (!g_prof[i].closed && g_prof[i].name == name) {
      (void)hipEventRecord(g_prof[i].b, s);
      g_prof[i].closed = true;
      return;
    }
}
static void prof_drain() {
  for (auto& r : g_prof) {
    if (!r.closed) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); continue; }
    (void)hipEventSynchronize(r.b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    auto& acc = g_prof_acc[r.name];
    acc.first += ms;
    acc.second += 1;
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_prof.clear();
}
extern "C" void rdrf_prof_reset(void) {
  prof_drain();
  g_prof_acc.clear();
}
extern "C" int rdrf_prof_enable(int on) {
  prof_drain();
  g_prof_on = on != 0;
  return 0;
}
extern "C" int rdrf_prof_get(const char* kernel, double* total_ms, int* launches) {
  prof_drain();
  auto it = g_prof_acc.find(kernel);
  if (it == g_prof_acc.end()) {
    *total_ms = 0;
    *launches = 0;
    return -1;
  }
  *total_ms = it->second.first;
  *launches = it->second.second;
  return 0;
}

// ------------------------------------------------------------------------------------------------
__global__ void k_pack(PackJobs jobs, float* __restrict__ dst) {
  const PackJob J = jobs.j[blockIdx.y];
  const int stride = gridDim.x * blockDim.x;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (J.mode == 0 || J.mode == 2) {
    const int total = J.nb * J.kk * 64;
    const int k4n = J.kk >> 2;
    for (int i = tid; i < total; i += stride) {
      const int q = i & 3, lane = (i >> 2) & 63, rest = i >> 8;
      const int k4 = rest % k4n, nb = rest / k4n;
      const int kk = k4 * 4 + q, h = lane >> 5, li = lane & 31;
      float v = 0.f;
      if (J.mode == 0) {
        const int o = nb * 32 + li;
        const int col = seg_imap(J.seg, elem_of(kk + J.seg_kk0, h), J.in_dim);
        if (o < J.out_dim && col >= 0) v = J.src[(size_t)o * J.ld + col];
      } else {
        const int o = elem_of(kk, h);
        const int col = seg_imap(J.seg, nb * 32 + li, J.in_dim);
        if (o < J.out_dim && col >= 0) v = J.src[(size_t)o * J.ld + col];
      }
      dst[J.dst + i] = v;
    }
  } else if (J.mode == 7) {  // bf16 x 3 fragments (mfma_seg_b3): [nb][kk_tot/8][3 pieces][64 lanes][4 dwords]; this job fills the
    // dwords of its slots kk_off .. kk_off + kk - 1 (two slots per dword: kk_off and kk are even)
    const int pairs = J.kk >> 1, k8n = J.kk_tot >> 3;
    const int total = J.nb * pairs * 3 * 64;
    unsigned* out = reinterpret_cast<unsigned*>(dst) + J.dst;
    for (int i = tid; i < total; i += stride) {
      const int lane = i & 63;
      int r = i >> 6;
      const int p = r % 3; r /= 3;
      const int pr = r % pairs, nb = r / pairs;
      const int h = lane >> 5, o = nb * 32 + (lane & 31);
      unsigned d = 0;
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {
        const int col = seg_imap(J.seg, elem_of(J.seg_kk0 + 2 * pr + e2, h), J.in_dim);
        const float v = (o < J.out_dim && col >= 0) ? J.src[(size_t)o * J.ld + col] : 0.f;
        unsigned ph, pm, pl;
        split3(v, ph, pm, pl);
        const unsigned piece = (p == 0 ? ph : (p == 1 ? pm : pl)) >> 16;
        d |= piece << (16 * e2);
      }
      const int kk = J.kk_off + 2 * pr;
      out[((size_t)((nb * k8n + (kk >> 3)) * 3 + p) * 64 + lane) * 4 + ((kk & 7) >> 1)] = d;
    }
  } else if (J.mode == 9) {  // bf16 x 3 fragments, split storage (mfma_seg_b3s): hi and mid pieces [nb][kk_tot/8][2][64 lanes][4 dwords]
    // at dst (the LDS image: the fp32 footprint), lo pieces [nb][kk_tot/8][64 lanes][4 dwords] at dst2 (streamed from L2)
    const int pairs = J.kk >> 1, k8n = J.kk_tot >> 3;
    const int total = J.nb * pairs * 3 * 64;
    unsigned* out = reinterpret_cast<unsigned*>(dst) + J.dst;
    unsigned* out_lo = reinterpret_cast<unsigned*>(dst) + J.dst2;
    for (int i = tid; i < total; i += stride) {
      const int lane = i & 63;
      int r = i >> 6;
      const int p = r % 3; r /= 3;
      const int pr = r % pairs, nb = r / pairs;
      const int h = lane >> 5, o = nb * 32 + (lane & 31);
      unsigned d = 0;
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {
        const int col = seg_imap(J.seg, elem_of(J.seg_kk0 + 2 * pr + e2, h), J.in_dim);
        const float v = (o < J.out_dim && col >= 0) ? J.src[(size_t)o * J.ld + col] : 0.f;
        unsigned ph, pm, pl;
        split3(v, ph, pm, pl);
        const unsigned piece = (p == 0 ? ph : (p == 1 ? pm : pl)) >> 16;
        d |= piece << (16 * e2);
      }
      const int kk = J.kk_off + 2 * pr;
      if (p < 2) out[((size_t)((nb * k8n + (kk >> 3)) * 2 + p) * 64 + lane) * 4 + ((kk & 7) >> 1)] = d;
      else out_lo[((size_t)(nb * k8n + (kk >> 3)) * 64 + lane) * 4 + ((kk & 7) >> 1)] = d;
    }
  } else if (J.mode == 8 || J.mode == 10) {  // bf16 x 3 fragments of a TRANSPOSED layer (backward data, as mode 2): [nb][kk/8][3][64 lanes][4 dwords]
    // (mode 10: split storage as mode 9 -- hi + mid at dst, lo at dst2);
    // K = the forward layer's outputs (slot kk of lane half h = neuron elem_of(kk, h)), rows = the segment's input elements
    const int pairs = J.kk >> 1, k8n = J.kk >> 3;
    const int total = J.nb * pairs * 3 * 64;
    unsigned* out = reinterpret_cast<unsigned*>(dst) + J.dst;
    for (int i = tid; i < total; i += stride) {
      const int lane = i & 63;
      int r = i >> 6;
      const int p = r % 3; r /= 3;
      const int pr = r % pairs, nb = r / pairs;
      const int h = lane >> 5;
      const int col = seg_imap(J.seg, nb * 32 + (lane & 31), J.in_dim);
      unsigned d = 0;
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {
        const int o = elem_of(2 * pr + e2, h);
        const float v = (o < J.out_dim && col >= 0) ? J.src[(size_t)o * J.ld + col] : 0.f;
        unsigned ph, pm, pl;
        split3(v, ph, pm, pl);
        const unsigned piece = (p == 0 ? ph : (p == 1 ? pm : pl)) >> 16;
        d |= piece << (16 * e2);
      }
      const int kk = 2 * pr;
      if (J.mode == 8) out[((size_t)((nb * k8n + (kk >> 3)) * 3 + p) * 64 + lane) * 4 + ((kk & 7) >> 1)] = d;
      else if (p < 2) out[((size_t)((nb * k8n + (kk >> 3)) * 2 + p) * 64 + lane) * 4 + ((kk & 7) >> 1)] = d;   // split storage
      else (reinterpret_cast<unsigned*>(dst) + J.dst2)[((size_t)(nb * k8n + (kk >> 3)) * 64 + lane) * 4 + ((kk & 7) >> 1)] = d;
    }
  } else if (J.mode == 4) {  // 16x16x4 fragments: [nb][kk/2][64 lanes][2], lane = (g << 4) | neuron
    const int total = J.nb * J.kk * 64;
    const int k2n = J.kk >> 1;
    for (int i = tid; i < total; i += stride) {
      const int q = i & 1, lane = (i >> 1) & 63, rest = i >> 7;
      const int k2 = rest % k2n, nb = rest / k2n;
      const int kk = k2 * 2 + q, g = lane >> 4, o = nb * 16 + (lane & 15);
      const int col = seg_imap16(J.seg, kk, g, J.in_dim);
      dst[J.dst + i] = (o < J.out_dim && col >= 0) ? J.src[(size_t)o * J.ld + col] : 0.f;
    }
  } else if (J.mode == 5) {  // small layer, 16-sample layout: [OUT][4 groups][kk]
    const int total = J.nb * 4 * J.kk;
    for (int i = tid; i < total; i += stride) {
      const int kk = i % J.kk, g = (i / J.kk) & 3, o = i / (4 * J.kk);
      const int col = seg_imap16(J.seg, kk, g, J.in_dim);
      dst[J.dst + i] = (o < J.out_dim && col >= 0) ? J.src[(size_t)o * J.ld + col] : 0.f;
    }
  } else if (J.mode == 6) {  // bias, 16-sample layout: [4 groups][kk]
    const int total = 4 * J.kk;
    for (int i = tid; i < total; i += stride) {
      const int kk = i % J.kk, g = i / J.kk;
      const int o = elem16_of(kk, g);
      dst[J.dst + i] = (J.src != nullptr && o < J.out_dim) ? J.src[o] : 0.f;
    }
  } else if (J.mode == 3) {  // bias: [2][kk] canonical
    const int total = 2 * J.kk;
    for (int i = tid; i < total; i += stride) {
      const int kk = i % J.kk, h = i / J.kk;
      const int o = elem_of(kk, h);
      dst[J.dst + i] = (J.src != nullptr && o < J.out_dim) ? J.src[o] : 0.f;
    }
  } else {
    const int total = J.nb * 2 * J.kk;
    for (int i = tid; i < total; i += stride) {
      const int kk = i % J.kk, h = (i / J.kk) & 1, o = i / (2 * J.kk);
      const int col = seg_imap(J.seg, elem_of(kk, h), J.in_dim);
      dst[J.dst + i] = (o < J.out_dim && col >= 0) ? J.src[(size_t)o * J.ld + col] : 0.f;
    }
  }
}

void pack_add(PackJobs& J, const float* src, int ld, int out_dim, int in_dim, int seg, int mode,
              int nb, int kk, int dst) {
  PackJob& j = J.j[J.n++];
  j.src = src;
  j.ld = ld;
  j.out_dim = out_dim;
  j.in_dim = in_dim;
  j.seg = seg;
  j.mode = mode;
  j.nb = nb;
  j.kk = kk;
  j.dst = dst;
  j.seg_kk0 = 0;
  j.kk_off = 0;
  j.kk_tot = 0;
  j.dst2 = 0;
}
// as pack_add_b3 with split storage: hi + mid pieces at dst (LDS image), lo pieces at dst_lo (streamed)
void pack_add_b3s(PackJobs& J, const float* src, int ld, int out_dim, int in_dim, int seg, int nb, int kk, int seg_kk0, int kk_off,
                  int kk_tot, int dst, int dst_lo) {
  pack_add(J, src, ld, out_dim, in_dim, seg, 9, nb, kk, dst);
  PackJob& j = J.j[J.n - 1];
  j.seg_kk0 = seg_kk0;
  j.kk_off = kk_off;
  j.kk_tot = kk_tot;
  j.dst2 = dst_lo;
}
// one segment (slots seg_kk0 .. seg_kk0 + kk - 1 of `seg`) of a bf16 x 3 image of kk_tot slots, at slot kk_off of the image
void pack_add_b3(PackJobs& J, const float* src, int ld, int out_dim, int in_dim, int seg, int nb, int kk, int seg_kk0, int kk_off,
                 int kk_tot, int dst) {
  pack_add(J, src, ld, out_dim, in_dim, seg, 7, nb, kk, dst);
  PackJob& j = J.j[J.n - 1];
  j.seg_kk0 = seg_kk0;
  j.kk_off = kk_off;
  j.kk_tot = kk_tot;
}
// a transposed layer (backward data) as bf16 x 3 fragments with split storage: hi + mid at dst (LDS image), lo at dst_lo
void pack_add_b3s_t(PackJobs& J, const float* src, int ld, int out_dim, int in_dim, int seg, int nbi, int kk, int dst, int dst_lo) {
  pack_add(J, src, ld, out_dim, in_dim, seg, 10, nbi, kk, dst);
  J.j[J.n - 1].dst2 = dst_lo;
}
// an fp32 MFMA segment that starts at slot seg_kk0 of `seg`
void pack_add_from(PackJobs& J, const float* src, int ld, int out_dim, int in_dim, int seg, int nb, int kk, int seg_kk0, int dst) {
  pack_add(J, src, ld, out_dim, in_dim, seg, 0, nb, kk, dst);
  J.j[J.n - 1].seg_kk0 = seg_kk0;
}

int pack_launch(const PackJobs& J, float* dst, hipStream_t stream) {
  RDRF_CHECK(J.n > 0 && J.n <= RDRF_MAX_PACK_JOBS, -2, "pack: bad job count %d", J.n);
  RDRF_LAUNCH("pack", k_pack, dim3(16, J.n), dim3(256), stream, J, dst);
  return 0;
}

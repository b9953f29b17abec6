// VQ index entropy coder (vq_vae_gan/model.py:337-576) on gfx950: codebook-usage histogram, the float64 cumulative
// table, and the 32-bit binary arithmetic coder.  Integer/byte work, HBM- and latency-bound: no MFMA here.
//
// An arithmetic-coded stream is one serial dependency chain (every symbol's interval depends on the previous
// one), so the parallel unit is the STREAM: one 64-lane wave per stream, all lanes carrying the same coder state
// (uniform control flow), with the wave's width used for what is parallel inside a stream:
//   encode: the 64 next symbols and their two table entries are fetched by 64 lanes at once (they do not depend on
//           the coder state) and handed to the serial update by v_readlane;
//   decode: the symbol search is a 64-ary search (64 interval bounds per round, one ballot), and the bitstream is
//           read through a 256-byte window held one big-endian word per lane.
// The renormalisation loop of model.py:398-423 is done in closed form: the common leading bits of low and high
// (the E1/E2 steps) leave at once (clz of low^high), and the run of E3 "underflow" steps that can only follow them
// is the run of leading 1s of low / 0s of high below the top bit.  The reference does its interval arithmetic in
// unbounded Python integers, and a table whose last entry rounds above 1.0 pushes `high` past 2^32; the state here
// is 64-bit and such symbols take the literal step-by-step loop, so the bytes stay identical.
#include "common.h"

namespace {

constexpr uint64_t AC_MAX = 0xFFFFFFFFull, AC_HALF = 0x80000000ull, AC_QUARTER = 0x40000000ull;
enum { AC_OK = 0, AC_ZERO_PROB = 1, AC_OVERFLOW = 2, AC_BAD_SYMBOL = 3, AC_STATE = 4 };

__device__ inline double readlane_d(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// ---- histogram -----------------------------------------------------------------------------------------------
__device__ inline int hist_bin(int64_t s, int k, int exact_bins) {
  if (s < 0 || s > k - 1) return -1;                         // histc ignores values outside [min, max]
  if (exact_bins) return (int)s;
  const int b = (int)((float)s / (float)(k - 1) * (float)k);  // torch.histc: (int64)((x - min) / (max - min) * bins) in f32
  return b >= k ? k - 1 : b;
}
// A trained codebook's usage is skewed (a few codes take most hits), so same-address global atomics serialise:
// blocks count in LDS and flush one atomic per used bin (13.3 ms -> tens of us for 4 M indices at K = 512).
template <bool LDS_BINS>
__global__ __launch_bounds__(256) void hist_kernel(const int64_t* __restrict__ idx, int64_t n, int k, int exact_bins,
                                                   int* __restrict__ counts) {
  extern __shared__ int s_bins[];
  if (LDS_BINS) {
    for (int i = threadIdx.x; i < k; i += 256) s_bins[i] = 0;
    __syncthreads();
  }
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int b = hist_bin(idx[i], k, exact_bins);
    if (b >= 0) atomicAdd(LDS_BINS ? s_bins + b : counts + b, 1);
  }
  if (LDS_BINS) {
    __syncthreads();
    for (int i = threadIdx.x; i < k; i += 256)
      if (s_bins[i]) atomicAdd(counts + i, s_bins[i]);
  }
}
__global__ void hist_total_kernel(const int* __restrict__ counts, int k, int* __restrict__ total) {
  int s = 0;
  for (int i = threadIdx.x; i < k; i += blockDim.x) s += counts[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(total, s);
}
__global__ void hist_probs_kernel(const int* __restrict__ counts, int k, const int* __restrict__ total, float* __restrict__ probs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < k) probs[i] = (float)counts[i] / (float)total[0];   // histogram / histogram.sum(), both exact below 2^24
}

// ---- cumulative table ------------------------------------------------------------------------------------------
// probs == nullptr: uniform, p = 1.0f / K (model.py:371, not renormalised).  Else p = probs / f32(sum probs)
// (model.py:373); then cum[i+1] = cum[i] + (double)p[i] sequentially in float64 (model.py:378-380).  One block; the
// chain is serial by definition (one lane walks it through LDS, 1024 entries per round).
constexpr int TBL_CHUNK = 1024;
__global__ __launch_bounds__(256) void table_kernel(const float* __restrict__ probs, int k, double* __restrict__ cum) {
  __shared__ float sp[TBL_CHUNK];
  __shared__ double sc[TBL_CHUNK];
  __shared__ double carry;
  __shared__ float s_norm;
  const int tid = threadIdx.x;
  if (probs) {
    if (tid == 0) carry = 0.0;
    for (int base = 0; base < k; base += TBL_CHUNK) {
      __syncthreads();
      for (int i = tid; i < TBL_CHUNK && base + i < k; i += 256) sp[i] = probs[base + i];
      __syncthreads();
      if (tid == 0) {
        double s = carry;
        const int m = min(TBL_CHUNK, k - base);
        for (int i = 0; i < m; ++i) s += (double)sp[i];
        carry = s;
      }
    }
    __syncthreads();
    if (tid == 0) s_norm = (float)carry;
  }
  __syncthreads();
  const float uni = 1.0f / (float)k;
  const float nrm = probs ? s_norm : 1.f;
  __syncthreads();
  if (tid == 0) { carry = 0.0; cum[0] = 0.0; }
  for (int base = 0; base < k; base += TBL_CHUNK) {
    __syncthreads();
    for (int i = tid; i < TBL_CHUNK && base + i < k; i += 256) sp[i] = probs ? probs[base + i] / nrm : uni;
    __syncthreads();
    if (tid == 0) {
      double s = carry;
      const int m = min(TBL_CHUNK, k - base);
      for (int i = 0; i < m; ++i) { s += (double)sp[i]; sc[i] = s; }
      carry = s;
    }
    __syncthreads();
    for (int i = tid; i < TBL_CHUNK && base + i < k; i += 256) cum[base + i + 1] = sc[i];
  }
}

// ---- bit output ------------------------------------------------------------------------------------------------
struct BitWriter {
  uint8_t* out;
  int64_t cap, nb;
  uint64_t acc;
  int nacc;
  bool lane0;
  __device__ inline void put(uint32_t v, int c) {             // append the low c (<= 32) bits of v, MSB first
    if (c == 0) return;
    acc = (acc << c) | (uint64_t)(c == 32 ? v : (v & ((1u << c) - 1u)));
    nacc += c;
    while (nacc >= 8) {
      nacc -= 8;
      if (lane0 && nb < cap) out[nb] = (uint8_t)(acc >> nacc);
      ++nb;
    }
  }
  __device__ inline void run(int bit, uint64_t count) {
    while (count > 0) {
      const int c = count > 32 ? 32 : (int)count;
      put(bit ? 0xFFFFFFFFu : 0u, c);
      count -= c;
    }
  }
  __device__ inline void finish() {                           // zero-pad the last byte (model.py:437-443)
    if (nacc > 0) {
      if (lane0 && nb < cap) out[nb] = (uint8_t)(acc << (8 - nacc));
      ++nb;
      nacc = 0;
    }
  }
};

// One symbol's renormalisation, encoder side.  Returns false if the state left the supported range.
__device__ inline bool enc_renorm(uint64_t& low, uint64_t& high, uint64_t& pending, BitWriter& w) {
  if ((high >> 32) == 0) {                                    // closed form (see the header comment)
    const uint32_t lo = (uint32_t)low, hi = (uint32_t)high;
    uint32_t nl = lo, nh = hi;
    const uint32_t x = lo ^ hi;
    const int c = x ? __clz(x) : 32;
    if (c) {
      const uint32_t top = lo >> (32 - c);                    // c >= 1: the common prefix, MSB first
      const int b0 = (top >> (c - 1)) & 1;
      w.put(b0, 1);
      w.run(!b0, pending);
      pending = 0;
      w.put(top, c - 1);
      nl = c == 32 ? 0u : lo << c;
      nh = c == 32 ? 0xFFFFFFFFu : (hi << c) | ((1u << c) - 1u);
    }
    const uint32_t a = ~(nl << 1), b = nh << 1;
    int e = min(a ? __clz(a) : 32, b ? __clz(b) : 32);
    e = min(e, 31);
    if (e) {
      pending += e;
      nl = (nl << e) & 0x7FFFFFFFu;
      nh = ((nh << e) | ((1u << e) - 1u)) | 0x80000000u;
    }
    low = nl;
    high = nh;
    return true;
  }
  for (int guard = 0; guard < 4096; ++guard) {                // literal loop of model.py:398-423 on the 64-bit state
    if (high < AC_HALF) { w.put(0, 1); w.run(1, pending); pending = 0; }
    else if (low >= AC_HALF) { w.put(1, 1); w.run(0, pending); pending = 0; low -= AC_HALF; high -= AC_HALF; }
    else if (low >= AC_QUARTER && high < 3 * AC_QUARTER) { ++pending; low -= AC_QUARTER; high -= AC_QUARTER; }
    else return (high >> 52) == 0;
    low = low * 2;
    high = high * 2 + 1;
    if (high >> 60) return false;
  }
  return false;
}

// grid = streams, block = one wave.
__global__ __launch_bounds__(64) void ac_encode_kernel(const int64_t* __restrict__ idx, int64_t n, const double* __restrict__ cum,
                                                       int k, uint8_t* __restrict__ out, int64_t cap, int* __restrict__ nbytes,
                                                       int* __restrict__ status) {
  const int lane = threadIdx.x;
  const int64_t* my = idx + (int64_t)blockIdx.x * n;
  BitWriter w{out + (int64_t)blockIdx.x * cap, cap, 0, 0, 0, lane == 0};
  uint64_t low = 0, high = AC_MAX, pending = 0;
  int st = AC_OK;
  for (int64_t t0 = 0; t0 < n && st == AC_OK; t0 += 64) {
    const int cnt = (int)min((int64_t)64, n - t0);
    int64_t s = lane < cnt ? my[t0 + lane] : 0;
    const bool bad = s < 0 || s >= k;
    if (__ballot(bad)) { st = AC_BAD_SYMBOL; break; }
    const double clo = cum[s], chi = cum[s + 1];
    for (int j = 0; j < cnt; ++j) {
      const double a = readlane_d(clo, j), b = readlane_d(chi, j);
      const double r = (double)(high - low + 1);
      const uint64_t nh = low + (uint64_t)(int64_t)(r * b) - 1;       // int(range * cum[s+1]): truncation (model.py:394)
      const uint64_t nl = low + (uint64_t)(int64_t)(r * a);           // model.py:395
      if ((int64_t)nh < (int64_t)nl) { st = AC_ZERO_PROB; break; }    // the reference's loop never ends on such a symbol
      low = nl;
      high = nh;
      if (!enc_renorm(low, high, pending, w)) { st = AC_STATE; break; }
    }
  }
  if (st == AC_OK) {                                                  // model.py:425-434
    ++pending;
    if (low < AC_QUARTER) { w.put(0, 1); w.run(1, pending); }
    else { w.put(1, 1); w.run(0, pending); }
    w.finish();
    if (w.nb > cap) st = AC_OVERFLOW;
  }
  if (lane == 0) {
    nbytes[blockIdx.x] = st == AC_OK ? (int)w.nb : 0;
    status[blockIdx.x] = st;
  }
}

// ---- decode ------------------------------------------------------------------------------------------------------
struct BitReader {
  const uint8_t* data;
  int64_t nbytes;
  int64_t wbase;        // first 32-bit word of the window
  uint32_t win;         // this lane's big-endian word
  int lane;
  __device__ inline void load(int64_t word0) {
    wbase = word0;
    const int64_t o = (word0 + lane) * 4;
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) v = (v << 8) | (o + i < nbytes ? (uint32_t)data[o + i] : 0u);
    win = v;
  }
  __device__ inline uint32_t get(int64_t pos, int c) {        // c (1..32) bits at bit position pos; past the end: zeros
    int64_t wd = (pos >> 5) - wbase;
    if (wd + 1 >= 64) { load(pos >> 5); wd = 0; }
    const int wi = __builtin_amdgcn_readfirstlane((int)wd);
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)win, wi);
    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)win, wi + 1);
    const uint64_t comb = ((uint64_t)a << 32) | b;
    return (uint32_t)((comb << (pos & 31)) >> (64 - c));
  }
};

template <bool LDS_TABLE>
__global__ __launch_bounds__(64) void ac_decode_kernel(const uint8_t* __restrict__ data, const int* __restrict__ nbytes, int64_t cap,
                                                       int64_t n, const double* __restrict__ cum_g, int k,
                                                       int64_t* __restrict__ idx, int* __restrict__ status) {
  extern __shared__ double s_cum[];
  const int lane = threadIdx.x;
  const double* cum = cum_g;
  if (LDS_TABLE) {
    for (int i = lane; i <= k; i += 64) s_cum[i] = cum_g[i];
    __syncthreads();
    cum = s_cum;
  }
  BitReader rd{data + (int64_t)blockIdx.x * cap, (int64_t)nbytes[blockIdx.x], 0, 0, lane};
  rd.load(0);
  int64_t* my = idx + (int64_t)blockIdx.x * n;
  uint64_t low = 0, high = AC_MAX;
  uint64_t value = rd.get(0, 32);
  int64_t pos = 32;
  int st = AC_OK;
  int64_t keep = 0;       // lane j keeps the symbol of step t0 + j; written 64 at a time
  for (int64_t t = 0; t < n; ++t) {
    const double r = (double)(high - low + 1);
    // largest s in [0, k) with low + int(r * cum[s]) <= value: 64 bounds per round
    int lo_s = 0, hi_s = k;
    while (hi_s - lo_s > 1) {
      const int step = (hi_s - lo_s + 63) >> 6;
      const int p = lo_s + lane * step;
      bool ok = false;
      if (p < hi_s) ok = low + (uint64_t)(int64_t)(r * cum[p]) <= value;
      const uint64_t m = __ballot(ok) | 1ull;               // lane 0 probes lo_s itself: true by the invariant
      const int nt = __popcll(m);
      lo_s += (nt - 1) * step;
      hi_s = min(hi_s, lo_s + step);
    }
    const int s = lo_s;
    if ((int)(t & 63) == lane) keep = s;
    if ((t & 63) == 63 || t == n - 1) {
      const int64_t t0 = t & ~(int64_t)63;
      if (t0 + lane <= t) my[t0 + lane] = keep;
    }
    const uint64_t nh = low + (uint64_t)(int64_t)(r * cum[s + 1]) - 1;
    const uint64_t nl = low + (uint64_t)(int64_t)(r * cum[s]);
    if ((int64_t)nh < (int64_t)nl) { st = AC_ZERO_PROB; break; }
    low = nl;
    high = nh;
    if ((high >> 32) == 0) {
      uint32_t lo = (uint32_t)low, hi = (uint32_t)high, v = (uint32_t)value;
      const uint32_t x = lo ^ hi;
      const int c = x ? __clz(x) : 32;
      if (c) {
        const uint32_t in = rd.get(pos, c);
        pos += c;
        lo = c == 32 ? 0u : lo << c;
        hi = c == 32 ? 0xFFFFFFFFu : (hi << c) | ((1u << c) - 1u);
        v = c == 32 ? in : (v << c) | in;
      }
      const uint32_t a = ~(lo << 1), b = hi << 1;
      int e = min(a ? __clz(a) : 32, b ? __clz(b) : 32);
      e = min(e, 31);
      if (e) {
        const uint32_t in = rd.get(pos, e);
        pos += e;
        lo = (lo << e) & 0x7FFFFFFFu;
        hi = ((hi << e) | ((1u << e) - 1u)) | 0x80000000u;
        v = ((v << e) | in) ^ 0x80000000u;
      }
      low = lo;
      high = hi;
      value = v;
    } else {
      bool fine = false;
      for (int guard = 0; guard < 4096; ++guard) {
        if (high < AC_HALF) {}
        else if (low >= AC_HALF) { low -= AC_HALF; high -= AC_HALF; value -= AC_HALF; }
        else if (low >= AC_QUARTER && high < 3 * AC_QUARTER) { low -= AC_QUARTER; high -= AC_QUARTER; value -= AC_QUARTER; }
        else { fine = (high >> 52) == 0; break; }
        low = low * 2;
        high = high * 2 + 1;
        value = value * 2 + rd.get(pos, 1);
        ++pos;
        if (high >> 60) break;
      }
      if (!fine) { st = AC_STATE; break; }
    }
  }
  if (lane == 0) status[blockIdx.x] = st;
}

}  // namespace

extern "C" {

int vg_codebook_usage(const int64_t* indices, int64_t n, int k, int exact_bins, int* counts_ws, float* probs, void* stream) {
  VG_REQUIRE(indices && counts_ws && probs, "codebook_usage: null pointer");
  VG_REQUIRE(k >= 2, "codebook_usage: num_embeddings=%d must be >= 2", k);
  VG_REQUIRE(n > 0 && n < (1 << 24), "codebook_usage: %lld indices (supported: 1 .. 2^24-1, where float32 counts are exact)", (long long)n);
  hipStream_t s = (hipStream_t)stream;
  VG_HIP(hipMemsetAsync(counts_ws, 0, sizeof(int) * ((size_t)k + 1), s));
  const int64_t want = vg_ceil_div(n, 256 * 16);             // >= 16 indices per thread before a block pays its flush
  const int blocks = (int)(want < 1 ? 1 : (want < 1024 ? want : 1024));
  if (k <= 16384) hist_kernel<true><<<blocks, 256, sizeof(int) * (size_t)k, s>>>(indices, n, k, exact_bins, counts_ws);
  else hist_kernel<false><<<blocks, 256, 0, s>>>(indices, n, k, exact_bins, counts_ws);
  VG_LAUNCH_CHECK();
  hist_total_kernel<<<1, 256, 0, s>>>(counts_ws, k, counts_ws + k);   // histogram.sum(): out-of-range indices do not count
  VG_LAUNCH_CHECK();
  hist_probs_kernel<<<(k + 255) / 256, 256, 0, s>>>(counts_ws, k, counts_ws + k, probs);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

int vg_ac_table(const float* probs, int k, double* cum, void* stream) {
  VG_REQUIRE(cum, "ac_table: null pointer");
  VG_REQUIRE(k >= 1 && k <= (1 << 20), "ac_table: %d symbols unsupported (1 .. 2^20)", k);
  table_kernel<<<1, 256, 0, (hipStream_t)stream>>>(probs, k, cum);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

int64_t vg_ac_stream_capacity(int64_t symbols_per_stream) { return 4 * symbols_per_stream + 8; }

int vg_ac_encode(const int64_t* indices, int n_streams, int64_t symbols_per_stream, const double* cum, int k, uint8_t* out,
                 int64_t cap_per_stream, int* nbytes, int* status, void* stream) {
  VG_REQUIRE(indices || symbols_per_stream == 0, "ac_encode: null indices");
  VG_REQUIRE(cum && out && nbytes && status, "ac_encode: null pointer");
  VG_REQUIRE(n_streams > 0 && symbols_per_stream >= 0 && k >= 1, "ac_encode: bad sizes (%d streams, %lld symbols, %d codes)",
             n_streams, (long long)symbols_per_stream, k);
  VG_REQUIRE(cap_per_stream >= 1 && cap_per_stream < (int64_t)1 << 31, "ac_encode: capacity %lld per stream unsupported",
             (long long)cap_per_stream);
  ac_encode_kernel<<<n_streams, 64, 0, (hipStream_t)stream>>>(indices, symbols_per_stream, cum, k, out, cap_per_stream, nbytes, status);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

int vg_ac_decode(const uint8_t* data, const int* nbytes, int n_streams, int64_t cap_per_stream, int64_t symbols_per_stream,
                 const double* cum, int k, int64_t* indices, int* status, void* stream) {
  VG_REQUIRE(data && nbytes && cum && status, "ac_decode: null pointer");
  VG_REQUIRE(indices || symbols_per_stream == 0, "ac_decode: null indices");
  VG_REQUIRE(n_streams > 0 && symbols_per_stream >= 0 && k >= 1 && cap_per_stream >= 1,
             "ac_decode: bad sizes (%d streams, %lld symbols, %d codes)", n_streams, (long long)symbols_per_stream, k);
  const size_t tbl = sizeof(double) * ((size_t)k + 1);
  hipStream_t s = (hipStream_t)stream;
  if (tbl <= 96 * 1024) {      // table in LDS (K <= 12287); two waves' worth of blocks still share a CU at K = 8192
    VG_HIP(vg_lds_attr((const void*)ac_decode_kernel<true>, 96 * 1024));
    ac_decode_kernel<true><<<n_streams, 64, tbl, s>>>(data, nbytes, cap_per_stream, symbols_per_stream, cum, k, indices, status);
  } else {
    ac_decode_kernel<false><<<n_streams, 64, 0, s>>>(data, nbytes, cap_per_stream, symbols_per_stream, cum, k, indices, status);
  }
  VG_LAUNCH_CHECK();
  return VG_OK;
}

}  // extern "C"

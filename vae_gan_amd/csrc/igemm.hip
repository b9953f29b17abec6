// Implicit-GEMM convolution kernels for gfx950 (MI355X).
//
// One gather-GEMM kernel serves nn.Conv2d forward, its input gradient and nn.ConvTranspose2d
// forward (stride-s transposed convs are split into s*s output phases, each a stride-1 gather
// with its own taps), and one pixel-reduction kernel serves the weight gradient.  Activations
// are NHWC; the GEMM K dimension walks (tap, channel) in 64-byte chunks (32 bf16 / 16 f32), so
// the bf16 path (v_mfma_f32_32x32x16_bf16) and the exact-f32 path (v_mfma_f32_32x32x2_f32) share
// all staging code: both consume 16-byte fragments.
//
// LDS tiles are [row][64 B] with the 16-byte slot XOR-swizzled by (row>>2)&3, which makes both
// the ds_write_b128 staging stores and the ds_read_b128 fragment loads bank-conflict free.
#include "common.h"
#include <type_traits>

#ifdef VG_ABLATE   // timing ablations only (results are wrong): see tests/gpu_conv_ablate.py
#define VG_DBG(bit) (p.dbg & (bit))
#define VG_ABLATE_LEVEL VG_ABLATE
#else
#define VG_DBG(bit) false
#define VG_ABLATE_LEVEL 0
#endif
#define VG_MAX_PHASES 4
#define VG_MAX_TAPS 16

struct IgemmParams {
  const void* x;
  const void* w;       // [phase][Npad][ntaps*C]
  const float* bias;   // [Nout] or null
  void* y;
  double* stats;       // [VG_STATS_REPLICAS][2][Npad] or null
  int N, H, W, C, Cs;  // gathered tensor: dims, reduction channels (multiple of KC), channel stride
  int OHl, OWl;        // logical output grid of one phase
  int OH, OW, Ks;      // physical output tensor
  int Nout, Npad;      // valid / padded GEMM-N (output channels)
  int in_s, out_s;     // gather stride, scatter stride
  int ntaps, nphases;
  int act, accumulate;
  int ksplit;          // >1: grid.z slices of the K loop; raw f32 partials go to splitws by atomics
  float* splitws;      // [pixels][Npad] f32, zeroed by the launcher
  int M;               // N*OHl*OWl
  int mtiles, ntiles;
  signed char dy[VG_MAX_PHASES][VG_MAX_TAPS];
  signed char dx[VG_MAX_PHASES][VG_MAX_TAPS];
  // spatial-tile ("halo") kernel only
  int tilesX, tilesY, HH, HWd, nsh;
  int flip;               // igemm3: tap offsets descend (input gradients, transposed gathers)
  int lgc;                // igemm3 stride-2 forward: log2 of the real 64-byte channel chunks per pixel
  int dbg;             // timing ablations only (wrong results): 1 = no in-loop DMA, 2 = no barrier/wait, 4 = no LDS fragment reads
  signed char dymin[VG_MAX_PHASES], dxmin[VG_MAX_PHASES];
  // tap offsets again, 4 bits per tap biased by 8: the kernels index them with scalar shifts.  (A dynamic
  // index into the byte tables above compiles to a vector load + s_waitcnt vmcnt(0), which drains the
  // LDS-DMA ring on every step.)
  unsigned long long dyp[VG_MAX_PHASES], dxp[VG_MAX_PHASES];
  int dymin_i[VG_MAX_PHASES], dxmin_i[VG_MAX_PHASES];
  // igemm3 epilogue, optional: y is the gradient of a BatchNorm's OUTPUT (this launch is the input-gradient kernel of
  // the layer that consumed it); take the BatchNorm backward sums (sum dz, sum dz*xhat per channel) while the tile is
  // still on chip instead of a separate two-tensor pass (bn_bwd_reduce).  bs_x = the BatchNorm's raw input, same
  // layout as y; bs_y = its output when the activation mask has to come from there (residual BatchNorm), else the mask
  // is recomputed as x*scale+shift > 0 like the forward pass did.
  vg_bn_bwd_stats bs;
  int det;                // statistics through the fixed-point accumulators (common.h: vg_stats_add)
};

static void pack_taps(IgemmParams& p) {
  for (int ph = 0; ph < VG_MAX_PHASES; ++ph) {
    unsigned long long a = 0, b = 0;
    for (int t = 0; t < VG_MAX_TAPS; ++t) {
      a |= (unsigned long long)((p.dy[ph][t] + 8) & 15) << (4 * t);
      b |= (unsigned long long)((p.dx[ph][t] + 8) & 15) << (4 * t);
    }
    p.dyp[ph] = a; p.dxp[ph] = b;
    p.dymin_i[ph] = p.dymin[ph]; p.dxmin_i[ph] = p.dxmin[ph];
  }
}
__device__ __forceinline__ int tap_off(unsigned long long pack, int tap) { return (int)((pack >> (4 * tap)) & 15) - 8; }

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  __device__ static inline f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // 16 bytes = 4 f32: lane half h holds k = {4*(2j+h)+i}; A and B use the same assignment, so
  // the four k=2 MFMAs below cover 8 distinct k values exactly once.
  __device__ static inline f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[i], c, 0, 0, 0);
    return c;
  }
};

__device__ __attribute__((aligned(64))) const uint32_t vg_zero_block[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

__device__ __forceinline__ int lds_off64(int row, int slot) { return row * 64 + (((slot ^ (row >> 2)) & 3) << 4); }

// igemm3, bf16: fragments of v_mfma_f32_16x16x32_bf16 (lane = row m = lane % 16, 16-byte k slot kb = lane / 16).  For
// that read pattern the slot is swizzled by bit 2 of the row: the four 16-lane groups a ds_read_b128 is served in then
// hit 16 distinct 16-byte bank groups for every tap shift (rows a, a+4, a+8, a+12 of a residue class carry slots
// kb, kb', kb', kb with kb' = kb ^ 1: the xor by 2 * (row/4 % 2) separates them).
template <bool M16> __device__ __forceinline__ int igemm3_swz(int slot, int row) {
  return M16 ? ((slot ^ (((row >> 2) & 1) << 1)) & 3) : ((slot ^ (row >> 2)) & 3);
}
__device__ __forceinline__ f32x4 mma16x16x32(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Sum and sum of squares of a pair of bf16 values into f32 accumulators with one v_dot2c_f32_bf16 each (the values ARE the
// bf16-rounded pre-activation the statistics are defined on: no conversion back to f32, no separate multiply):
// statistics-only epilogues of the weights-stationary kernels, where one wave per SIMD pays for every VALU instruction.
typedef __bf16 vg_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2_sum(vg_bf16x2 q, float s) {
  const vg_bf16x2 one = {(__bf16)1.0f, (__bf16)1.0f};
  return __builtin_amdgcn_fdot2_f32_bf16(q, one, s, false);
}
__device__ __forceinline__ float dot2_sq(vg_bf16x2 q, float s) { return __builtin_amdgcn_fdot2_f32_bf16(q, q, s, false); }

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  // bijective: blocks that share an XCD (bid % 8) get one contiguous range of logical ids
  int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ring depth: as many 64-byte-chunk stages as fit next to a second resident block (<= 80 KiB per block)
__host__ __device__ constexpr int igemm_stages(int bm, int bn) { return (bm + bn) * 64 * 4 <= 65536 ? 4 : 3; }

static thread_local char g_last_kernel[64] = "";
extern "C" const char* vg_last_conv_kernel(void) { return g_last_kernel; }
template <typename T> static const char* tname() { return sizeof(T) == 2 ? "bf16" : "f32"; }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LDS-DMA the compiler does not see.  With the builtin it tracks the DMA as a pending LDS write and, because
// ds_read_b64_tr_b16 (unlike a plain ds_read) carries no alias information, puts `s_waitcnt vmcnt(0)` in front of the
// first transposed read after every DMA: the weight-gradient kernels then waited, once per tile, for the prefetch they
// had just issued (a full memory latency per tile; that was the "latency-bound stream").  The kernels order DMA and
// reads themselves (counted vmcnt + barrier), as they must anyway.  lds_byte_off is wave-uniform.
// M0 (the DMA's LDS base; read when the instruction issues) is put back afterwards: the compiler is not told about it.
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_byte_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_byte_off), "v"(gsrc)
               : "memory");
}
// The same with the global address as a wave-uniform base (an SGPR pair) plus a 32-bit per-lane byte offset: one
// address register per outstanding piece instead of two, and no 64-bit vector add per piece.  (igemm3 kept one 64-bit
// per-lane pointer per halo round live across its K loop; with the 16x16x32 accumulators two of them spilled, and every
// reload came with an `s_waitcnt vmcnt(0)` that drained the DMA ring: +7 % cycles.)
__device__ __forceinline__ void lds_dma16_s(const void* sbase, unsigned voff, unsigned lds_byte_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_byte_off), "v"(voff), "s"(sbase)
               : "memory");
}
// LDS-DMA through a buffer resource, for the one-wave-per-SIMD kernels, where every instruction between two MFMAs counts.
// A lane whose offset is outside the resource's range gets ZEROS written to its LDS slot (tests/micro/buf_lds.hip), so an
// out-of-image halo piece needs no address select against a block of zeros: its offset is replaced by 0xFFFFFFFF (one
// v_cndmask with an inline constant instead of a 64-bit add and a 64-bit select).  M0 is written and not restored: the
// compiler treats it as reserved and neither reads nor writes it in these kernels (checked in the ISA:
// scripts/check_m0.py); a kernel that mixes this with builtins that use M0 must save and restore it as lds_dma16 does.
typedef int vg_srd_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ vg_srd_t vg_make_srd(const void* base) {       // raw buffer: stride 0, 2 GB of records
  const unsigned long long b = (unsigned long long)base;
  vg_srd_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b & 0xFFFFFFFFull));
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((b >> 32) & 0xFFFFull));
  r[2] = (int)0x80000000u;
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ void lds_dma16_buf(vg_srd_t srd, unsigned voff, unsigned lds_byte_off) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds_byte_off), "v"(voff), "s"(srd) : "memory");
}
// ... and with M0 put back, for kernels that also use lds_dma16 / lds_dma16_s
__device__ __forceinline__ void lds_dma16_bufk(vg_srd_t srd, unsigned voff, unsigned lds_byte_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_byte_off), "v"(voff), "s"(srd) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)p;
}

template <typename T, int MI, int NI, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
  constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
  constexpr int EB = sizeof(T), KC = 64 / EB, PIECE = 16 / EB;
  constexpr int AR = BM / 64;                   // A rows staged per thread
  constexpr int BR = (BN + 63) / 64;            // B rows staged per thread (BN=32: threads < 128)
  constexpr int STAGE_A = BM * 64, STAGE_B = BN * 64;
  constexpr int NS = igemm_stages(BM, BN);      // LDS ring depth (chunks resident or in flight)
  constexpr int STAGE = NS * (STAGE_A + STAGE_B);
  constexpr int TILE = BM * BN * 4;
  constexpr int MAIN = STAGE > TILE ? STAGE : TILE;
  constexpr int LPC = AR + (BN >= 64 ? BR : 0); // LDS-DMA instructions every wave issues per chunk (lower bound)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + NS * STAGE_A;
  unsigned* rowpix = (unsigned*)(smem + MAIN);
  float* red = (float*)(smem + MAIN + BM * 4);   // [WM][BN][2]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, l32 = lane & 31;
  const int lid = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
  const int ntile = lid % p.ntiles, mtile = lid / p.ntiles;
  const int phase = blockIdx.y;
  const int py = phase / p.out_s, px = phase % p.out_s;
  const int m0 = mtile * BM, n0 = ntile * BN;
  const int ohw = p.OHl * p.OWl;

  // ---- per-row gather metadata (rows are fixed for the whole K loop)
  const int srow = tid >> 2, sslot = tid & 3;
  int a_nh[AR], a_iy0[AR], a_ix0[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    int m = m0 + srow + 64 * i;
    if (m < p.M) {
      int n = m / ohw, rem = m - n * ohw;
      int oy = rem / p.OWl, ox = rem - oy * p.OWl;
      a_nh[i] = n * p.H; a_iy0[i] = oy * p.in_s; a_ix0[i] = ox * p.in_s;
    } else {
      a_nh[i] = 0; a_iy0[i] = -(1 << 24); a_ix0[i] = 0;
    }
  }
  for (int r = tid; r < BM; r += 256) {
    int m = m0 + r;
    unsigned v = 0xFFFFFFFFu;
    if (m < p.M) {
      int n = m / ohw, rem = m - n * ohw;
      int oy = rem / p.OWl, ox = rem - oy * p.OWl;
      v = (unsigned)((n * p.OH + oy * p.out_s + py) * p.OW + ox * p.out_s + px);
    }
    rowpix[r] = v;
  }

  const T* __restrict__ xg = (const T*)p.x;
  const int cchunks = p.C / KC;
  const int nk_all = p.ntaps * cchunks;
  const int kbeg = (int)((long)nk_all * blockIdx.z / p.ksplit), kend = (int)((long)nk_all * (blockIdx.z + 1) / p.ksplit);
  const int nk = kend - kbeg;
  const size_t ktot = (size_t)p.ntaps * p.C;
  const T* __restrict__ wg = (const T*)p.w + ((size_t)phase * p.Npad + n0) * ktot;

  // Global -> LDS staging uses LDS-DMA (global_load_lds_dwordx4): one wave instruction writes 1 KiB
  // = 16 rows x 64 B linearly, so thread (srow, sslot) lands at physical slot sslot of its row and
  // must FETCH the logical slot sslot ^ ((row>>2)&3) (swizzle applied on the source address).
  // Padding / out-of-range rows read a 64-byte zero block instead.
  const int lslot = (sslot ^ (srow >> 2)) & 3;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  int tap = kbeg / cchunks, cc = kbeg - tap * cchunks;   // position of the NEXT chunk to load
  const unsigned long long dyp = p.dyp[phase], dxp = p.dxp[phase];
  auto load_chunk = [&](int buf) {
    const int dy = tap_off(dyp, tap), dx = tap_off(dxp, tap);
    const int coff = cc * KC + lslot * PIECE;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
      bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const void* src = ok ? (const void*)(xg + ((size_t)(a_nh[i] + iy) * p.W + ix) * p.Cs + coff) : (const void*)vg_zero_block;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (lds_ptr_t)(sA + buf * STAGE_A + (wave * 16 + 64 * i) * 64), 16, 0, 0);
    }
    const size_t koff = (size_t)tap * p.C + coff;
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      int r = srow + 64 * i;
      if (BN >= 64 || wave < 2)   // BN == 32: only rows 0..31 exist (waves 0 and 1), wave-uniform branch
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wg + (size_t)r * ktot + koff),
                                         (lds_ptr_t)(sB + buf * STAGE_B + (wave * 16 + 64 * i) * 64), 16, 0, 0);
    }
    if (++cc == cchunks) { cc = 0; ++tap; }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Ring pipeline: chunks ks .. ks+NS-2 are in flight or resident; the DMA for chunk ks+NS-1 is issued
  // right after the barrier of step ks into the buffer every wave finished reading in step ks-1.
  // Waits are counted (never drained inside the steady state) and the barrier is a raw s_barrier so the
  // outstanding LDS-DMA stays in flight across it.
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0)
    if (s0 < nk) load_chunk(s0);
  int rbuf = 0, wbuf = NS - 1;
  for (int ks = 0; ks < nk; ++ks) {
    if (ks + NS - 2 < nk) wait_vmcnt<(NS - 2) * LPC>();   // chunk ks landed (this wave's share)
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ks + NS - 1 < nk) load_chunk(wbuf);
    const unsigned char* cA = sA + rbuf * STAGE_A;
    const unsigned char* cB = sB + rbuf * STAGE_B;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      u32x4 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = *(const u32x4*)(cA + lds_off64((wm * MI + i) * 32 + l32, 2 * j + h));
#pragma unroll
      for (int i = 0; i < NI; ++i) fb[i] = *(const u32x4*)(cB + lds_off64((wn * NI + i) * 32 + l32, 2 * j + h));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) acc[i][jj] = Mma<T>::run(fa[i], fb[jj], acc[i][jj]);
    }
    rbuf = rbuf + 1 == NS ? 0 : rbuf + 1;
    wbuf = wbuf + 1 == NS ? 0 : wbuf + 1;
  }
  __syncthreads();   // all fragment reads done before the epilogue tile aliases the ring

  if (p.ksplit > 1) {
    // split-K: add this slice's raw partial sums into the f32 workspace (bias/activation happen in the
    // finish kernel); one lane half covers 32 consecutive channels = one 128-byte segment per row
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = n0 + (wn * NI + j) * 32 + l32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned pix = rowpix[(wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
          if (pix != 0xFFFFFFFFu) unsafeAtomicAdd(&p.splitws[(size_t)pix * p.Npad + n], acc[i][j][r]);
        }
      }
    return;
  }

  // ---- epilogue: bias, stats (on dtype-rounded pre-activation), activation, LDS-staged stores
  float* tile = (float*)smem;   // [BM][BN] f32, aliases the staging buffers (all reads done: synced)
  float s1[NI], s2[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int col = (wn * NI + j) * 32 + l32;
    const int n = n0 + col;
    const float bv = (p.bias != nullptr && n < p.Nout) ? p.bias[n] : 0.f;
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = acc[i][j][r] + bv;
        v = Elt<T>::to_f(Elt<T>::from_f(v));
        if (m0 + row < p.M) { a1 += v; a2 += v * v; }
        if (p.act == VG_ACT_LRELU) v = vg_lrelu(v);
        else if (p.act == VG_ACT_SIGMOID) v = vg_sigmoid(v);
        if (n >= p.Nout) v = 0.f;
        tile[row * BN + col] = v;
      }
    }
    s1[j] = a1 + __shfl_xor(a1, 32, 64);
    s2[j] = a2 + __shfl_xor(a2, 32, 64);
  }
  if (p.stats != nullptr && h == 0) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = (wn * NI + j) * 32 + l32;
      red[(wm * BN + col) * 2 + 0] = s1[j];
      red[(wm * BN + col) * 2 + 1] = s2[j];
    }
  }
  __syncthreads();
  if (p.stats != nullptr && tid < BN && n0 + tid < p.Nout) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) { t1 += red[(w * BN + tid) * 2]; t2 += red[(w * BN + tid) * 2 + 1]; }
    vg_stats_add(p.stats, p.Npad, 0, n0 + tid, mtile + phase, (double)t1, p.det);
    vg_stats_add(p.stats, p.Npad, 1, n0 + tid, mtile + phase, (double)t2, p.det);
  }
  T* __restrict__ yg = (T*)p.y;
  constexpr int PPR = BN / PIECE;
  for (int q = tid; q < BM * PPR; q += 256) {
    const int row = q / PPR, pc = q - row * PPR;
    const unsigned pix = rowpix[row];
    const int n = n0 + pc * PIECE;
    if (pix == 0xFFFFFFFFu || n >= p.Ks) continue;
    Piece<T> o;
    const float* src = tile + row * BN + pc * PIECE;
    T* dst = yg + (size_t)pix * p.Ks + n;
    if (p.accumulate) {
      Piece<T> old;
      old.v = *(const decltype(old.v)*)dst;
#pragma unroll
      for (int e = 0; e < PIECE; ++e) o.set(e, src[e] + old.get(e));
    } else {
#pragma unroll
      for (int e = 0; e < PIECE; ++e) o.set(e, src[e]);
    }
    *(decltype(o.v)*)dst = o.v;
  }
}

template <typename T, int MI, int NI, int WM, int WN>
static int launch_igemm(const IgemmParams& p0, hipStream_t st) {
  IgemmParams p = p0;
  constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
  constexpr int STAGE = igemm_stages(BM, BN) * (BM * 64 + BN * 64), TILE = BM * BN * 4;
  constexpr int MAIN = STAGE > TILE ? STAGE : TILE;
  const size_t lds = MAIN + BM * 4 + WM * BN * 2 * 4;
  p.mtiles = (int)vg_ceil_div(p.M, BM);
  p.ntiles = p.Npad / BN;
  auto kern = igemm_kernel<T, MI, NI, WM, WN>;
    VG_HIP(vg_lds_attr((const void*)kern, (int)lds));
  dim3 grid(p.mtiles * p.ntiles, p.nphases, p.ksplit);
  pack_taps(p);
  snprintf(g_last_kernel, sizeof g_last_kernel, "igemm<%s,%d,%d,%d,%d>%s", tname<T>(), MI, NI, WM, WN, p.ksplit > 1 ? "+splitk" : "");
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// =============================================================================================
// igemm2: spatial-tile kernel.  A block owns TH x 16 output pixels of ONE image.  For every 64-byte
// channel chunk the input patch that all taps need (the tile plus its halo) is brought into LDS ONCE by
// LDS-DMA; the R*S taps then read their shifted A fragments from that patch, so the L2->LDS traffic of the
// activations drops by ~R*S (9x for 3x3, 4x for the stride-2 4x4) compared with the im2col-row kernel
// above, which was bound by that traffic (~30 B/clk/CU), not by MFMA.  Weights stream through a 4-deep
// ring (one tap-chunk per step) with counted vmcnt waits and raw s_barrier, as above.
// =============================================================================================
template <typename T, int MI, int NI, int WM, int WN, int TH>
__global__ __launch_bounds__(64 * WM * WN) void igemm2_kernel(const IgemmParams p) {
  constexpr int NT = 64 * WM * WN, TW = 16, BM = TH * TW, BN = 32 * NI * WN;
  static_assert(BM == 32 * MI * WM, "tile / wave layout mismatch");
  constexpr int EB = sizeof(T), KC = 64 / EB, PIECE = 16 / EB;
  constexpr int NSB = 4, PF = NSB - 1;
  constexpr int BSTRIDE = (BN >= 64 ? BN : 64) * 64;             // BN == 32: rows 32..63 are a dummy target (uniform vmcnt)
  constexpr int LPB = (BSTRIDE / 16) / NT;                       // B DMA instructions per wave per step
  static_assert(LPB >= 1, "every wave must issue at least one weight DMA per step");
  constexpr int MAXROWS = TH == 16 ? 19 * 19 : 18 * 34;
  constexpr int HRMAX = (MAXROWS * 4 + NT - 1) / NT;
  constexpr int WTILE = 32 * MI * 32 * NI * EB;                  // per-wave epilogue staging bytes
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int hrows = p.HH * p.HWd;
  const int hbytes = ((hrows + 15) & ~15) * 64;
  unsigned char* sH = smem;                                      // [nsh][hbytes]
  unsigned char* sB = smem + p.nsh * hbytes;                     // [NSB][BSTRIDE]
  const int pipe_bytes = p.nsh * hbytes + NSB * BSTRIDE;
  const int main_bytes = pipe_bytes > WM * WN * WTILE ? pipe_bytes : WM * WN * WTILE;
  unsigned* rowpix = (unsigned*)(smem + main_bytes);
  float* red = (float*)(smem + main_bytes + BM * 4);             // [WM][BN][2]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, l32 = lane & 31;
  const int lid = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
  const int ntile = lid % p.ntiles, mtile = lid / p.ntiles;
  const int phase = blockIdx.y;
  const int py = phase / p.out_s, px = phase % p.out_s;
  const int n0 = ntile * BN;
  const int tpi = p.tilesX * p.tilesY;
  const int img = mtile / tpi;
  const int trem = mtile - img * tpi;
  const int ty0 = (trem / p.tilesX) * TH, tx0 = (trem % p.tilesX) * TW;
  const int dymin = p.dymin_i[phase], dxmin = p.dxmin_i[phase];
  const unsigned long long dyp = p.dyp[phase], dxp = p.dxp[phase];
  const int swz = p.in_s == 2 ? 1 : 2;                            // halo-row swizzle: slot ^= (row >> swz) & 3

  for (int r = tid; r < BM; r += NT) {
    const int oy = ty0 + r / TW, ox = tx0 + r % TW;
    unsigned v = 0xFFFFFFFFu;
    if (oy < p.OHl && ox < p.OWl) v = (unsigned)((img * p.OH + oy * p.out_s + py) * p.OW + ox * p.out_s + px);
    rowpix[r] = v;
  }

  // ---- halo pieces owned by this thread: 16-byte piece q = round*NT + tid -> halo row q>>2, physical slot q&3
  const T* __restrict__ xg = (const T*)p.x;
  unsigned poff[HRMAX];     // source offset in 16-byte units (chunk 0), 0xFFFFFFFF = zero fill
  const int iy_base = ty0 * p.in_s + dymin, ix_base = tx0 * p.in_s + dxmin;
#pragma unroll
  for (int rd = 0; rd < HRMAX; ++rd) {
    const int q = rd * NT + tid;
    const int hr = q >> 2, ps = q & 3;
    unsigned o = 0xFFFFFFFFu;
    if (hr < hrows) {
      const int hy = hr / p.HWd, hx = hr - hy * p.HWd;
      const int iy = iy_base + hy, ix = ix_base + hx;
      if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
        const int ls = (ps ^ (hr >> swz)) & 3;
        o = (unsigned)((((size_t)(img * p.H + iy) * p.W + ix) * p.Cs * EB) >> 4) + ls;
      }
    }
    poff[rd] = o;
  }
  const int cchunks = p.C / KC;
  const int nk = p.ntaps * cchunks;
  const size_t ktot = (size_t)p.ntaps * p.C;
  const T* __restrict__ wg = (const T*)p.w + ((size_t)phase * p.Npad + n0) * ktot;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gptr_t;

  int itap = 0, icc = 0, ihb = 0, ibb = 0;     // issue cursor: tap, chunk, halo buffer, B buffer
  auto issue = [&]() {
    if (itap == 0) {
      unsigned char* dst = sH + ihb * hbytes;
#pragma unroll
      for (int rd = 0; rd < HRMAX; ++rd) {
        const int q0w = rd * NT + wave * 64;                    // first piece of this wave in this round (uniform)
        if (q0w < hrows * 4) {
          const int q = q0w + lane;
          if (q < hrows * 4) {
            const unsigned char* src = poff[rd] == 0xFFFFFFFFu ? (const unsigned char*)vg_zero_block
                                                                : (const unsigned char*)xg + ((size_t)poff[rd] << 4) + (size_t)icc * 64;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_ptr_t)(dst + q0w * 16), 16, 0, 0);
          }
        }
      }
      ihb = ihb + 1 == p.nsh ? 0 : ihb + 1;
    }
    {
      const size_t koff = (size_t)itap * p.C + (size_t)icc * KC;
      unsigned char* dst = sB + ibb * BSTRIDE;
#pragma unroll
      for (int i = 0; i < LPB; ++i) {
        const int q = i * NT + tid;                             // piece of the [rows][64 B] weight chunk
        int row = q >> 2;
        const int ps = q & 3;
        if (BN < 64) row &= (BN - 1);                           // dummy rows re-read real ones
        const int ls = (ps ^ (row >> 2)) & 3;
        __builtin_amdgcn_global_load_lds((gptr_t)(wg + (size_t)row * ktot + koff + ls * PIECE),
                                         (lds_ptr_t)(dst + (i * NT + wave * 64) * 16), 16, 0, 0);
      }
      ibb = ibb + 1 == NSB ? 0 : ibb + 1;
    }
    if (++itap == p.ntaps) { itap = 0; ++icc; }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // halo row of this lane's pixel for each A block (tap offset added per step)
  int arow[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int pr = (wm * MI + i) * 32 + l32;
    arow[i] = (pr / TW) * p.in_s * p.HWd + (pr % TW) * p.in_s;
  }

#pragma unroll
  for (int s0 = 0; s0 < PF; ++s0)
    if (s0 < nk) issue();
  int ctap = 0, chb = 0, cbb = 0;              // compute cursor
  for (int s = 0; s < nk; ++s) {
    if (!VG_DBG(2)) {
      if (s + PF - 1 < nk) wait_vmcnt<(PF - 1) * LPB>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("" ::: "memory");
    if (s + PF < nk && !VG_DBG(1)) issue();
    const unsigned char* cH = sH + chb * hbytes;
    const unsigned char* cB = sB + cbb * BSTRIDE;
    const int toff = (tap_off(dyp, ctap) - dymin) * p.HWd + (tap_off(dxp, ctap) - dxmin);
    int abase[MI], aswz[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = arow[i] + toff;
      abase[i] = row * 64;
      aswz[i] = (row >> swz) & 3;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      u32x4 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = *(const u32x4*)(cH + (VG_DBG(4) ? 0 : abase[i] + ((((2 * j + h) ^ aswz[i]) & 3) << 4)));
#pragma unroll
      for (int i = 0; i < NI; ++i) fb[i] = *(const u32x4*)(cB + (VG_DBG(4) ? 0 : lds_off64((wn * NI + i) * 32 + l32, 2 * j + h)));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) acc[i][jj] = Mma<T>::run(fa[i], fb[jj], acc[i][jj]);
    }
    cbb = cbb + 1 == NSB ? 0 : cbb + 1;
    if (++ctap == p.ntaps) { ctap = 0; chb = chb + 1 == p.nsh ? 0 : chb + 1; }
  }
  __syncthreads();   // every fragment read is done before the staging area aliases the rings

  // ---- epilogue: bias, BN statistics, activation; each wave stages its own sub-tile in T and stores 16 B pieces
  T* wt = (T*)(smem + wave * WTILE);            // [32*MI][32*NI]
  constexpr int WN_COLS = 32 * NI;
  float s1[NI], s2[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + (wn * NI + j) * 32 + l32;
    const float bv = (p.bias != nullptr && n < p.Nout) ? p.bias[n] : 0.f;
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lrow = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = acc[i][j][r] + bv;
        v = Elt<T>::to_f(Elt<T>::from_f(v));
        if (rowpix[wm * MI * 32 + lrow] != 0xFFFFFFFFu) { a1 += v; a2 += v * v; }
        if (p.act == VG_ACT_LRELU) v = vg_lrelu(v);
        else if (p.act == VG_ACT_SIGMOID) v = vg_sigmoid(v);
        if (n >= p.Nout) v = 0.f;
        wt[lrow * WN_COLS + j * 32 + l32] = Elt<T>::from_f(v);
      }
    }
    s1[j] = a1 + __shfl_xor(a1, 32, 64);
    s2[j] = a2 + __shfl_xor(a2, 32, 64);
  }
  if (p.stats != nullptr && h == 0) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = (wn * NI + j) * 32 + l32;
      red[(wm * BN + col) * 2 + 0] = s1[j];
      red[(wm * BN + col) * 2 + 1] = s2[j];
    }
  }
  __syncthreads();
  if (p.stats != nullptr && tid < BN && n0 + tid < p.Nout) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) { t1 += red[(w * BN + tid) * 2]; t2 += red[(w * BN + tid) * 2 + 1]; }
    vg_stats_add(p.stats, p.Npad, 0, n0 + tid, mtile + phase, (double)t1, p.det);
    vg_stats_add(p.stats, p.Npad, 1, n0 + tid, mtile + phase, (double)t2, p.det);
  }
  T* __restrict__ yg = (T*)p.y;
  constexpr int PPR = WN_COLS / PIECE;
  for (int q = lane; q < 32 * MI * PPR; q += 64) {
    const int lrow = q / PPR, pc = q - lrow * PPR;
    const unsigned pix = rowpix[wm * MI * 32 + lrow];
    const int n = n0 + wn * WN_COLS + pc * PIECE;
    if (pix == 0xFFFFFFFFu || n >= p.Ks) continue;
    Piece<T> o;
    o.v = *(const decltype(o.v)*)(wt + lrow * WN_COLS + pc * PIECE);
    T* dst = yg + (size_t)pix * p.Ks + n;
    if (p.accumulate) {
      Piece<T> old;
      old.v = *(const decltype(old.v)*)dst;
#pragma unroll
      for (int e = 0; e < PIECE; ++e) o.set(e, o.get(e) + old.get(e));
    }
    *(decltype(o.v)*)dst = o.v;
  }
}

static int g_dbg = 0;   // timing ablations (vg_set_conv_kernel bits 8+)
#ifdef VG_ABLATE
static void* g_stamp_buf = nullptr;     // ablation build: per-block s_memtime stamps of igemm3 (dbg bit 8)
extern "C" void vg_dbg_set_stamp_buffer(void* p) { g_stamp_buf = p; }
#endif
template <typename T, int MI, int NI, int WM, int WN, int TH>
static int launch_igemm2(const IgemmParams& p0, hipStream_t st, bool* handled) {
  IgemmParams p = p0;
  constexpr int NT = 64 * WM * WN, BM = TH * 16, BN = 32 * NI * WN, EB = sizeof(T);
  constexpr int NSB = 4, PF = NSB - 1;
  constexpr int BSTRIDE = (BN >= 64 ? BN : 64) * 64;
  constexpr int MAXROWS = TH == 16 ? 19 * 19 : 18 * 34;
  constexpr int WTILE = 32 * MI * 32 * NI * EB;
  *handled = false;
  int ext_y = 0, ext_x = 0;
  for (int ph = 0; ph < p.nphases; ++ph) {
    int ymin = 127, ymax = -127, xmin = 127, xmax = -127;
    for (int t = 0; t < p.ntaps; ++t) {
      ymin = p.dy[ph][t] < ymin ? p.dy[ph][t] : ymin; ymax = p.dy[ph][t] > ymax ? p.dy[ph][t] : ymax;
      xmin = p.dx[ph][t] < xmin ? p.dx[ph][t] : xmin; xmax = p.dx[ph][t] > xmax ? p.dx[ph][t] : xmax;
    }
    p.dymin[ph] = (signed char)ymin; p.dxmin[ph] = (signed char)xmin;
    ext_y = ymax - ymin > ext_y ? ymax - ymin : ext_y;
    ext_x = xmax - xmin > ext_x ? xmax - xmin : ext_x;
  }
  p.HH = (TH - 1) * p.in_s + ext_y + 1;
  p.HWd = (16 - 1) * p.in_s + ext_x + 1;
  const int hrows = p.HH * p.HWd;
  if (hrows > MAXROWS) return VG_OK;          // not handled: caller falls back to the im2col-row kernel
  p.nsh = 1 + (PF + p.ntaps - 1) / p.ntaps;
  const int hbytes = ((hrows + 15) & ~15) * 64;
  const int pipe = p.nsh * hbytes + NSB * BSTRIDE;
  const int mainb = pipe > WM * WN * WTILE ? pipe : WM * WN * WTILE;
  const size_t lds = (size_t)mainb + BM * 4 + WM * BN * 2 * 4;
  if (lds > 160 * 1024) return VG_OK;
  p.dbg = g_dbg;
  p.tilesX = (int)vg_ceil_div(p.OWl, 16);
  p.tilesY = (int)vg_ceil_div(p.OHl, TH);
  p.mtiles = p.N * p.tilesX * p.tilesY;
  p.ntiles = p.Npad / BN;
  auto kern = igemm2_kernel<T, MI, NI, WM, WN, TH>;
    VG_HIP(vg_lds_attr((const void*)kern, 160 * 1024));
  pack_taps(p);
  snprintf(g_last_kernel, sizeof g_last_kernel, "igemm2<%s,%d,%d,%d,%d,%d>", tname<T>(), MI, NI, WM, WN, TH);
  hipLaunchKernelGGL(kern, dim3(p.mtiles * p.ntiles, p.nphases), dim3(NT), lds, st, p);
  VG_LAUNCH_CHECK();
  *handled = true;
  return VG_OK;
}


// =============================================================================================
// igemm3: ping-pong spatial-tile kernel for stride-1 gathers with a static R x S tap grid (3x3 and 4x4
// convs and their input gradients, the 2x2-tap phases of stride-2 transposed gathers) whose output rows
// are at least 24 pixels wide.  bf16 only (the f32 parity mode keeps the kernels above).
//
// Block = 8 waves = two groups of four (one wave of each group per SIMD) over a TH x 32 pixel tile of one
// image and BN output channels; a wave owns a (32*MI) x (32*NI) accumulator.  The two groups run one
// barrier apart: while one issues the 8 MFMA blocks of a phase, the other reads its next fragments
// from LDS and issues its share of the LDS-DMA prefetch (the 2-barriers-per-phase schedule of the guide's
// 256^2 GEMM template, applied to the halo gather).  A wave issues at most one instruction per ~4 cycles,
// so the memory part of a phase only hides behind the other group's 256 MFMA cycles if it is ~50
// instructions: the tap loop is fully unrolled (tap geometry, halo round and prefetch tap are compile-time),
// the halo image uses a fixed 35-pixel row stride with the 16-byte slot XOR-swizzled by the COLUMN
// (so a tap shift is one scalar offset and the 32-row blocks of a wave are immediates), out-of-image
// halo pieces are zeroed once and masked off in every DMA, and all counted vmcnt waits are constants.
//
// Weights stream through a 5-slot ring three K tiles ahead.  The halo of channel chunk cc+1 rides on
// the prefetch slots of chunk cc.  A wait keeps the two youngest weight tiles in flight; halo pieces
// among them only make it wait for slightly more (never less) than needed.
// =============================================================================================
template <typename T, int MI, int NI, int WM, int WN, int R, int S>
struct Igemm3Cfg {
  static constexpr int NT = 512, TW = 32, TH = MI * WM, BM = TH * TW, BN = 32 * NI * WN;
  static constexpr int EB = sizeof(T), KC = 64 / EB, PIECE = 16 / EB;
  static constexpr int NTAPS = R * S, PF = 3, NSB = PF + 2;
  // phases per K tile: two of 16 MFMAs (16x16x32) for the 128 x 64 wave tile, except on the 2x2 tap grid, where one phase
  // of 32 runs the stride-2 forwards 2-4 % faster and the transposed phases the same (the 3x3 kernels spill with the second
  // phase's B fragments live: 1293 -> 1062 TFLOP/s)
  static constexpr int NPH = (MI * NI == 8 && R * S != 4) ? 2 : 1;
  static constexpr int KPP = 2 / NPH;                         // 16-byte fragment steps per phase
  static constexpr int BSTRIDE = BN * 64;
  static constexpr int LPB = (BN * 4 + NT - 1) / NT;          // weight DMA instructions per wave per K tile
  static constexpr int BLANES = BN * 4 >= NT ? 64 : BN * 4 / 8;   // active lanes of one (BN = 64: half-masked)
  static constexpr int HS = 35, HSB = HS * 64, HH = TH + R - 1;   // halo image: HH rows of 35 pixels x 64 B
  static constexpr int HRN = (HH * HS * 4 + NT - 1) / NT;     // DMA rounds (512 x 16 B) per halo image
  static constexpr int NSH = NTAPS >= PF + 1 + HRN ? 2 : 3;   // halo buffers
  static constexpr int HSTART = NSH == 2 ? PF + 1 : 0;        // first prefetch tap that carries a halo round
  static constexpr int WTILE = 32 * 32 * NI * EB;             // per-wave epilogue staging: one 32-row block
  // a halo buffer is HRN whole DMA rounds, unless only the exact image (the last round partly masked off) lets the
  // configuration fit: 16 x 32 px x 128 ch with 2x2 taps (three halo buffers) is 6 KB over otherwise
  static constexpr int HPAD = HRN * NT * 16, HEXACT = HH * HS * 64;
  static constexpr int lds_for(int hb) {
    const int pipe = NSH * hb + NSB * BSTRIDE;
    const int mainb = pipe > 8 * WTILE ? pipe : 8 * WTILE;
    return mainb + BM * 4 + WM * BN * 2 * 4 + (mainb >= 8 * WTILE + 4 * BN * 4 ? 0 : 4 * BN * 4);
  }
  static constexpr int HBYTES = lds_for(HPAD) <= 160 * 1024 ? HPAD : HEXACT;
  static constexpr int PIPE = NSH * HBYTES + NSB * BSTRIDE;
  static constexpr int MAINB = PIPE > 8 * WTILE ? PIPE : 8 * WTILE;
  // [4][BN] f32 BatchNorm-backward coefficients (epilogue only): behind the epilogue staging tiles inside the then idle
  // pipeline area when there is room (every configuration in use), else appended
  static constexpr bool COEF_INSIDE = MAINB >= 8 * WTILE + 4 * BN * 4;
  static constexpr int COEF_OFF = COEF_INSIDE ? 8 * WTILE : MAINB + BM * 4 + WM * BN * 2 * 4;
  static constexpr int LDS = MAINB + BM * 4 + WM * BN * 2 * 4 + (COEF_INSIDE ? 0 : 4 * BN * 4);
  static constexpr bool FITS = LDS <= 160 * 1024 && (NSH == 2 || HRN <= 2 * NTAPS);
};

// S2 = true: the 4x4 stride-2 pad-1 FORWARD gather, run as a 2x2-tap stride-1 gather over the four parity planes of
// the input, X_ab[i][j] = x[2i-1+a][2j-1+b] (a virtual (OH+1) x (OW+1) image with 4C channels: plane-major chunks):
//   y[i][j] = sum_{dr,ds in {0,1}} sum_{a,b,c} X_ab[i+dr][j+ds][c] * W[o][c][2dr+a][2ds+b].
// Nothing is materialised: a halo piece of virtual chunk (plane, ic) is fetched from pixel (2i-1+a, 2j-1+b) (one
// scalar addend per plane on top of the piece's offset for plane (0,0)), the weight tile of virtual tap (dr,ds) and
// that chunk is tap (2dr+a, 2ds+b) of the ordinary packed forward weights, and pieces that leave the image in one
// plane but not in another are fetched from a block of zeros for that plane.  The activations then travel L2 -> LDS
// once per tile instead of once per tap (16x) as in the im2col-row kernel that ran these layers at 27 % MFMA busy.
// EPI: the epilogue features an instantiation carries.  0: all of them; 1: forward (bias, activation, BatchNorm
// statistics; plain stores); 2: plain gradient (no activation, no statistics, no fused sums; += allowed); 3: gradient with
// the fused BatchNorm-backward sums.  The launcher picks by the launch's arguments.  One kernel with every feature behind
// run-time tests is ~110 KB of code, most of it epilogue variants a launch never runs, and its size costs time.
template <typename T, int MI, int NI, int WM, int WN, int R, int S, bool S2, int EPI>
__device__ __forceinline__ void igemm3_body(const IgemmParams& p) {
  typedef Igemm3Cfg<T, MI, NI, WM, WN, R, S> Cfg;
  static_assert(WM * WN == 8, "two groups of four waves");
  static_assert(!S2 || (R == 2 && S == 2), "the stride-2 forward runs on the 2x2 tap grid");
  constexpr int NT = Cfg::NT, TW = Cfg::TW, TH = Cfg::TH, BM = Cfg::BM, BN = Cfg::BN;
  constexpr int EB = Cfg::EB, KC = Cfg::KC, PIECE = Cfg::PIECE;
  constexpr int NTAPS = Cfg::NTAPS, PF = Cfg::PF, NSB = Cfg::NSB, NPH = Cfg::NPH, KPP = Cfg::KPP;
  constexpr int BSTRIDE = Cfg::BSTRIDE, LPB = Cfg::LPB, BLANES = Cfg::BLANES;
  constexpr int HS = Cfg::HS, HSB = Cfg::HSB, HH = Cfg::HH, HRN = Cfg::HRN, HBYTES = Cfg::HBYTES, NSH = Cfg::NSH;
  constexpr int HSTART = Cfg::HSTART, WTILE = Cfg::WTILE;
  constexpr int SB_OFF = NSH * HBYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* rowpix = (unsigned*)(smem + Cfg::MAINB);
  float* red = (float*)(smem + Cfg::MAINB + BM * 4);   // [WM][BN][2]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // SGPR: everything derived from it stays scalar
#ifdef VG_ABLATE   // per-block s_memtime stamps (shader cycles) of the kernel's stages: tests/gpu_conv_stamps.py
  unsigned long long* stamps = (VG_DBG(8) && p.splitws) ? (unsigned long long*)p.splitws + (size_t)blockIdx.x * 8 : nullptr;
#define VG_STAMP(i) do { if (stamps && tid == 0) stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define VG_STAMP(i) do {} while (0)
#endif
  VG_STAMP(0);
  const int group = wave >> 2;
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, l32 = lane & 31;
  // bf16: v_mfma_f32_16x16x32_bf16 (the chip holds a higher clock on it than on 32x32x16 at the same cycles per FLOP:
  // +8..16 % on these kernels).  A 32-pixel tile row = two 16-pixel A tiles (mh), 32 channels = two 16-channel B tiles (nh);
  // lane = row / column m16 of its tile, kb = its 16-byte k slot of the 64-byte chunk.  f32: 32x32x2, lane halves as before.
  constexpr bool M16 = sizeof(T) == 2;
  const int m16 = lane & 15, kb = lane >> 4;
  const int lid = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
  const int ntile = lid % p.ntiles, mtile = lid / p.ntiles;
  const int phase = blockIdx.y;
  const int py = phase / p.out_s, px = phase % p.out_s;
  const int n0 = ntile * BN;
  const int tpi = p.tilesX * p.tilesY;
  const int img = mtile / tpi;
  const int trem = mtile - img * tpi;
  const int ty0 = (trem / p.tilesX) * TH, tx0 = (trem % p.tilesX) * TW;
  const int dymin = p.dymin_i[phase], dxmin = p.dxmin_i[phase];
  const bool flip = p.flip != 0;       // taps run high-to-low offsets (input gradients, transposed gathers)

  for (int r = tid; r < BM; r += NT) {
    const int oy = ty0 + r / TW, ox = tx0 + r % TW;
    unsigned v = 0xFFFFFFFFu;
    if (oy < p.OHl && ox < p.OWl) v = (unsigned)((img * p.OH + oy * p.out_s + py) * p.OW + ox * p.out_s + px);
    rowpix[r] = v;
  }

  // ---- halo pieces of this thread: 16-byte piece q = round*NT + tid -> halo pixel q>>2 = hy*35 + hx, physical slot q&3.
  // Pieces outside the image are zeroed here in every halo buffer and never written again.
  const unsigned char* __restrict__ xg = (const unsigned char*)p.x;
  unsigned poff[HRN];       // byte offset of the piece in channel chunk 0, 0xFFFFFFFF = not fetched
  unsigned pval[HRN];       // S2: bit (2a+b) = the piece is inside the image in parity plane (a,b); poff is then SIGNED, plane (0,0)
  const int iy_base = ty0 + dymin, ix_base = tx0 + dxmin;
#pragma unroll
  for (int rd = 0; rd < HRN; ++rd) {
    const int q = rd * NT + tid;
    const int pix = q >> 2, ps = q & 3;
    const int hy = pix / HS, hx = pix - hy * HS;
    const int iy = iy_base + hy, ix = ix_base + hx;
    unsigned o = 0xFFFFFFFFu;
    if constexpr (S2) {
      const int row0 = 2 * iy - 1, col0 = 2 * ix - 1;
      unsigned m = 0;
      if (hy < HH && hx < TW + S - 1) {
#pragma unroll
        for (int ab = 0; ab < 4; ++ab)
          if ((unsigned)(row0 + (ab >> 1)) < (unsigned)p.H && (unsigned)(col0 + (ab & 1)) < (unsigned)p.W) m |= 1u << ab;
      }
      pval[rd] = m;
      const long off = (((long)(img * p.H + row0) * p.W + col0) * p.Cs) * EB + (igemm3_swz<M16>(ps, hx) << 4);
      if (m) o = (unsigned)(int)off;
      poff[rd] = o;
      if (m == 0 && q * 16 < HBYTES) {
#pragma unroll
        for (int b = 0; b < NSH; ++b) *(u32x4*)(smem + b * HBYTES + q * 16) = u32x4{0, 0, 0, 0};
      }
    } else {
      pval[rd] = 0;
      if (hy < HH && hx < TW + S - 1 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        o = (unsigned)((((size_t)(img * p.H + iy) * p.W + ix) * p.Cs) * EB) + (igemm3_swz<M16>(ps, hx) << 4);
      poff[rd] = o;
      if (o == 0xFFFFFFFFu && q * 16 < HBYTES) {
#pragma unroll
        for (int b = 0; b < NSH; ++b) *(u32x4*)(smem + b * HBYTES + q * 16) = u32x4{0, 0, 0, 0};
      }
    }
  }
  const int cch = p.C / KC;                                 // real chunks per pixel
  const int cchunks = S2 ? 4 * cch : cch;                   // S2: plane-major virtual chunks
  const size_t ktot = (size_t)(S2 ? 16 : NTAPS) * p.C;      // S2: rows of the ordinary packed 4x4 forward weights
  const unsigned char* __restrict__ wg = (const unsigned char*)((const T*)p.w + ((size_t)phase * p.Npad + n0) * ktot);
  const int cbytes = p.C * EB;          // one tap of one weight row

  unsigned boff[LPB];       // weight piece of this thread: byte offset inside the [BN][ktot] slab
#pragma unroll
  for (int i = 0; i < LPB; ++i) {
    const int q = i * NT + wave * BLANES + (lane & (BLANES - 1));
    const int row = q >> 2, ps = q & 3;
    boff[i] = (unsigned)((row * ktot + (igemm3_swz<M16>(ps, row) * PIECE)) * EB);
  }
  // fragment addresses.  f32: A = ct[s] + scalar(tile), +i*HSB per 32-row block, ^32 for the second k step;
  // B = bfo + scalar(slot), +2048 per 32-column block, ^32 likewise.  bf16: A tile mh = +1024 (16 pixels: the swizzle bit
  // is the same), B tile (jj, nh) = +(2*jj + nh)*1024
  int ct[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int hx = (M16 ? m16 : l32) + (flip ? S - 1 - s : s);
    ct[s] = hx * 64 + (igemm3_swz<M16>(M16 ? kb : h, hx) << 4);
  }
  const int brow = wn * NI * 32 + (M16 ? m16 : l32);
  const int bfo = brow * 64 + (igemm3_swz<M16>(M16 ? kb : h, brow) << 4) + SB_OFF;

  // accumulators.  f32: acc[i][j] = 32 x 32 (row (r&3) + 8*(r>>2) + 4h, column l32).  bf16: acc4[i][mh][j][nh] = 16 x 16
  // (pixel mh*16 + 4*kb + r, column j*32 + nh*16 + m16)
  f32x16 acc[M16 ? 1 : MI][M16 ? 1 : NI];
  f32x4 acc4[M16 ? MI : 1][2][M16 ? NI : 1][2];
  if constexpr (M16) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) acc4[i][mh][j][nh] = f32x4{0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

  // ---- DMA helpers; tap / round arguments are compile-time after unrolling.  Weight pieces and (stride-1) halo pieces
  // go out as wave-uniform base + 32-bit lane offset (lds_dma16_s): the compiler does not see these DMAs, every wait
  // for them is one of the kernel's own counted vmcnt waits.
  const unsigned smem0 = lds_addr_of(smem);
  const vg_srd_t xsrd = vg_make_srd(xg);
  auto dma_b = [&](int i, int itap, int icc, int slot) {
    const unsigned char* src;
    if constexpr (S2) {
      const int plane = icc >> p.lgc, ic = icc & (cch - 1);
      const int rs = (2 * (itap >> 1) + (plane >> 1)) * 4 + 2 * (itap & 1) + (plane & 1);
      src = wg + ((size_t)rs * cbytes + (size_t)ic * 64);
    } else {
      src = wg + ((size_t)itap * cbytes + (size_t)icc * 64);
    }
    if (VG_DBG(16)) src = wg;                  // timing ablation: every weight tile from the same (cache-hot) 8-16 KB
    const unsigned dst = smem0 + SB_OFF + slot * BSTRIDE + (i * NT + wave * BLANES) * 16;
    if (BLANES == 64 || lane < BLANES) lds_dma16_s(src, boff[i], dst);
  };
  auto dma_h = [&](int rd, int icc, int hboff) {
    if constexpr (S2) {
      if (pval[rd]) {
        // a piece outside the image in this plane gets offset 0xFFFFFFFF: the buffer resource writes zeros for it
        // (lds_dma16_buf).  Piece offsets are 32-bit two's complement from the tensor's start (launcher: < 2^31 bytes).
        const int plane = icc >> p.lgc, ic = icc & (cch - 1);
        const unsigned add = (unsigned)(((plane >> 1) * p.W + (plane & 1)) * p.Cs * EB + ic * 64);
        const unsigned voff = ((pval[rd] >> plane) & 1) ? poff[rd] + add : 0xFFFFFFFFu;
        lds_dma16_bufk(xsrd, voff, smem0 + hboff + (rd * NT + wave * 64) * 16);
      }
    } else {
      if (poff[rd] != 0xFFFFFFFFu) lds_dma16_s(xg + (size_t)icc * 64, poff[rd], smem0 + hboff + (rd * NT + wave * 64) * 16);
    }
  };
  // halo rounds of chunk icc that ride on the prefetch slot of tap itap
  auto halo_rounds = [&](int itap, int icc, int hboff) {
    if (icc < cchunks) {
#pragma unroll
      for (int rd = 0; rd < HRN; ++rd)
        if (NSH == 2 ? rd == itap - HSTART : rd % NTAPS == itap) dma_h(rd, icc, hboff);
    }
  };

  // ---- prologue: halo of chunk 0, weights (and attached halo rounds of chunk 1) of the first PF tiles
#pragma unroll
  for (int rd = 0; rd < HRN; ++rd) dma_h(rd, 0, 0);
#pragma unroll
  for (int t0 = 0; t0 < PF; ++t0) {
#pragma unroll
    for (int i = 0; i < LPB; ++i) dma_b(i, (t0 % R) * S + t0 / R, 0, t0);
    halo_rounds(t0, 1, NSH > 1 ? HBYTES : 0);
  }
  wait_vmcnt<0>();                                   // the prologue DMA (the compiler does not count the asm ones)
  __syncthreads();                                   // ... landed for every wave; zero fill and rowpix visible
  VG_STAMP(1);
  if (group == 1 && !VG_DBG(2)) __builtin_amdgcn_s_barrier();     // group 1 runs one barrier behind group 0

  const int wrow0 = (wm * MI + (flip ? R + MI - 2 : 0)) * HSB;     // window slot 0, and the step to the next slot
  const int wstep = flip ? -HSB : HSB;
  int cbs = 0;                         // weight slot of the tile being computed
  int hb0 = 0;                         // halo buffer offset of the chunk being computed
  for (int cc = 0; cc < cchunks; ++cc) {
    const int hb1 = hb0 + HBYTES == NSH * HBYTES ? 0 : hb0 + HBYTES;     // chunk cc+1
    const int hb2 = hb1 + HBYTES == NSH * HBYTES ? 0 : hb1 + HBYTES;     // chunk cc+2
    const int wcc = cc + 1 < cchunks ? cc + 1 : 0;   // weights prefetched past the end re-read chunk 0 (never consumed)
    // The taps are walked COLUMN by column (step sq = s*R + r): the R taps of a column need the A rows r .. r+MI-1 of
    // the same shifted halo image, so a window of R+MI-1 row fragments per k half stays in registers and every step after
    // the first of a column reads ONE new row instead of MI (3x3, MI = 4: 4 LDS reads per 8 MFMAs instead of 6).
    // Window slot u is halo row u, or R+MI-2-u when the taps run high to low (flip): the set a step needs is
    // u = r .. r+MI-1 either way, and accumulator i then belongs to output row MI-1-i (undone in the epilogue).
    u32x4 win[2][R + MI - 1];
#pragma unroll
    for (int sq = 0; sq < NTAPS; ++sq) {
      const int s = sq / R, r = sq % R;
      const int isq = (sq + PF) % NTAPS;             // step the prefetch of this step belongs to
      const int itap = (isq % R) * S + isq / R;      // ... and its tap in the packed weights
      const bool wrap = sq + PF >= NTAPS;
      const int ibs = cbs + PF >= NSB ? cbs + PF - NSB : cbs + PF;
      const int va0 = ct[s] + hb0 + wrow0;
      const int vb0 = bfo + cbs * BSTRIDE;
#pragma unroll
      for (int ph = 0; ph < NPH; ++ph) {
        u32x4 fb[KPP][NI];
        // ---- memory part: fragments of this phase, DMA prefetch, retire the next tile's DMA.
        // f32: step j is the k half.  bf16: an MFMA takes all 32 k values, step j is the 16-COLUMN half nh of every
        // 32-column block; the new window rows of both 16-pixel halves mh (win[mh]) are read in the K tile's first step
#pragma unroll
        for (int kk = 0; kk < KPP; ++kk) {
          const int j = ph * KPP + kk;
          if constexpr (M16) {
            if (j == 0) {
#pragma unroll
              for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int u = (r == 0 ? 0 : r + MI - 1); u < r + MI; ++u)
                  win[mh][u] = *(const u32x4*)(smem + (VG_DBG(4) ? 0 : va0 + mh * 1024 + u * wstep));
            }
#pragma unroll
            for (int jj = 0; jj < NI; ++jj) fb[kk][jj] = *(const u32x4*)(smem + (VG_DBG(4) ? 0 : vb0 + (2 * jj + j) * 1024));
          } else {
            const int va = j ? va0 ^ 32 : va0, vb = VG_DBG(4) ? 0 : (j ? vb0 ^ 32 : vb0);
#pragma unroll
            for (int u = (r == 0 ? 0 : r + MI - 1); u < r + MI; ++u)
              win[j][u] = *(const u32x4*)(smem + (VG_DBG(4) ? 0 : va + u * wstep));
#pragma unroll
            for (int jj = 0; jj < NI; ++jj) fb[kk][jj] = *(const u32x4*)(smem + vb + jj * 2048);
          }
        }
        if (!VG_DBG(1)) {
#pragma unroll
          for (int i = 0; i < LPB; ++i)
            if (i % NPH == ph) dma_b(i, itap, wrap ? wcc : cc, ibs);
          if (ph == NPH - 1) halo_rounds(isq, wrap ? cc + 2 : cc + 1, wrap ? hb2 : hb1);
        }
        if (ph == NPH - 1 && !VG_DBG(2)) wait_vmcnt<(PF - 1) * LPB>();
        if (!VG_DBG(2)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifndef VG_EXP_NOPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int kk = 0; kk < KPP; ++kk)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int jj = 0; jj < NI; ++jj) {
              if constexpr (M16) {
#pragma unroll
                for (int mh = 0; mh < 2; ++mh)
                  acc4[i][mh][jj][ph * KPP + kk] = mma16x16x32(win[mh][r + i], fb[kk][jj], acc4[i][mh][jj][ph * KPP + kk]);
              } else {
                acc[i][jj] = Mma<T>::run(win[ph * KPP + kk][r + i], fb[kk][jj], acc[i][jj]);
              }
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!VG_DBG(2)) __builtin_amdgcn_s_barrier();
      }
      cbs = cbs + 1 == NSB ? 0 : cbs + 1;
    }
    hb0 = hb1;
  }
  if (group == 0 && !VG_DBG(2)) __builtin_amdgcn_s_barrier();     // both groups have now passed the same number of barriers
  wait_vmcnt<0>();   // weight tiles prefetched past the end
  __syncthreads();   // every fragment read is done before the staging area aliases the rings
  VG_STAMP(2);

  // ---- epilogue: bias, BN statistics, activation; each wave stages one 32-row block (= one tile row) at a time in T.
  // The common case (tile and channel block fully inside, no sigmoid) runs a lean packed-f32 path: with
  // 64..128 accumulators per lane the epilogue is thousands of VALU cycles per block otherwise.
  T* wt = (T*)(smem + wave * WTILE);            // [32][32*NI]
  constexpr int WN_COLS = 32 * NI;
  constexpr int PPR = WN_COLS / PIECE;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  T* __restrict__ yg = (T*)p.y;
  // bf16 staging tile: the lanes of a 32-lane store group hold rows R and R+4 of the same 16 columns (kb and kb+1), which
  // are the same banks in a plain [row][column] tile; all eight waves stage at once, so the conflict would double the
  // LDS-array time of the epilogue.  Rows with bit 2 set keep their two column halves swapped.
  auto wt_off = [&](int lrow, int col) { return lrow * WN_COLS + (M16 ? col ^ (((lrow >> 2) & 1) * (WN_COLS / 2)) : col); };
  constexpr int NH = M16 ? 2 : 1;        // bf16: a lane holds column nh*16 + m16 of each 32-column block, f32: column l32
  float bv[NI][NH];
  f32x2 a1[NI][NH], a2[NI][NH];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int nh = 0; nh < NH; ++nh) {
      const int n = n0 + (wn * NI + j) * 32 + (M16 ? nh * 16 + m16 : l32);
      bv[j][nh] = (p.bias != nullptr && n < p.Nout) ? p.bias[n] : 0.f;
      a1[j][nh] = f32x2{0.f, 0.f}; a2[j][nh] = f32x2{0.f, 0.f};
    }
  constexpr bool E_FWD = EPI <= 1, E_ACC = EPI != 1, E_BS = EPI == 0 || EPI == 3;
  const int act = E_FWD ? p.act : (int)VG_ACT_NONE;
  const bool accumulate = E_ACC && p.accumulate;
  const bool fast = sizeof(T) == 2 && ty0 + TH <= p.OHl && tx0 + TW <= p.OWl && n0 + BN <= p.Nout && act != VG_ACT_SIGMOID;
  const bool want_stats = E_FWD && p.stats != nullptr;
  // BatchNorm-backward sums of the tile (see IgemmParams::bs): coefficients of this block's channels go to LDS once,
  // every lane keeps the partial sums of ITS 16-byte channel piece (lane % PPR is the same in every store round)
  const bool bs = E_BS && p.bs.x != nullptr;
  float* coef = (float*)(smem + Cfg::COEF_OFF);       // [scale | shift | invstd | mean*invstd][BN]
  float bs1[PIECE], bs2[PIECE];
#pragma unroll
  for (int e = 0; e < PIECE; ++e) { bs1[e] = 0.f; bs2[e] = 0.f; }
  if (bs) {
    for (int c = tid; c < BN; c += NT) {
      const int n = n0 + c;
      const bool ok = n < p.Nout;
      const float is = ok ? p.bs.invstd[n] : 0.f;
      coef[c] = (ok && p.bs.scale) ? p.bs.scale[n] : 0.f;
      coef[BN + c] = (ok && p.bs.shift) ? p.bs.shift[n] : 0.f;
      coef[2 * BN + c] = is;
      coef[3 * BN + c] = ok ? p.bs.mean[n] * is : 0.f;
    }
    __syncthreads();
  }
  constexpr int ROUNDS = 32 * PPR / 64;          // store rounds of one 32-row block: 16-byte piece lane + 64*rd
  static_assert(64 % PPR == 0, "a lane keeps its piece column over the store rounds");
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const unsigned* rp = rowpix + (wm * MI + (flip ? MI - 1 - i : i)) * 32;    // see the A window of the main loop
    // BatchNorm-backward sums: this block's pieces of the raw BatchNorm input (and of its output when the mask comes
    // from there) are requested NOW, before the accumulators are converted and staged, so that their memory latency
    // runs under that work (loaded right where they are used, every store round stalled for a full latency)
    Piece<T> xr[ROUNDS], yr[ROUNDS];
    asm volatile("" ::: "memory");      // keep the requests of block i behind the stores of block i-1 (register pressure)
    if (bs) {
#pragma unroll
      for (int rd = 0; rd < ROUNDS; ++rd) {
        const int q = lane + 64 * rd;
        const int lrow = q / PPR, pc = q - lrow * PPR;
        const unsigned pix = rp[lrow];
        const int n = n0 + wn * WN_COLS + pc * PIECE;
        const bool ok = pix != 0xFFFFFFFFu && n < p.Ks;
        const size_t off = ok ? (size_t)pix * p.Ks + n : 0;
        xr[rd].v = *(const decltype(xr[rd].v)*)((const T*)p.bs.x + off);
        if (p.bs.y != nullptr) yr[rd].v = *(const decltype(yr[rd].v)*)((const T*)p.bs.y + off);
        else yr[rd].v = xr[rd].v;      // never looked at
      }
    }
    if (fast) {
      // lane base; everything else below is an immediate.  bf16 (M16): pixel mh*16 + 4*kb + r, column j*32 + nh*16 + m16
      T* wl = wt + 4 * h * WN_COLS + l32;
      T* wlb[NI][NH];       // bf16: rows 4*kb + (0..3) + 16*mh: their swap bit is kb & 1 for every immediate row offset
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int nh = 0; nh < NH; ++nh) wlb[j][nh] = wt + wt_off(4 * kb, j * 32 + nh * 16 + m16);
      auto lean = [&](auto LR, auto ST) {
        if constexpr (M16) {
#pragma unroll
          for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
              for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                  const int row = mh * 16 + r;                                    // rows row, row + 1 (+ 4kb in wl)
                  f32x2 v = f32x2{acc4[i][mh][j][nh][r], acc4[i][mh][j][nh][r + 1]} + bv[j][nh];
                  bf16x2_t q = __builtin_convertvector(v, bf16x2_t);
                  v = __builtin_convertvector(q, f32x2);
                  if (decltype(ST)::value) { a1[j][nh] += v; a2[j][nh] += v * v; }
                  if (decltype(LR)::value) {
                    const f32x2 w = v * VG_LRELU_SLOPE;
                    f32x2 m;
                    asm("v_max_f32 %0, %1, %2" : "=v"(m[0]) : "v"(v[0]), "v"(w[0]));
                    asm("v_max_f32 %0, %1, %2" : "=v"(m[1]) : "v"(v[1]), "v"(w[1]));
                    q = __builtin_convertvector(m, bf16x2_t);
                  }
                  wlb[j][nh][row * WN_COLS] = q[0];
                  wlb[j][nh][(row + 1) * WN_COLS] = q[1];
                }
        }
      };
      if constexpr (sizeof(T) == 2) {
        typedef std::integral_constant<bool, true> Yes;
        typedef std::integral_constant<bool, false> No;
        if (act == VG_ACT_LRELU) { if (want_stats) lean(Yes{}, Yes{}); else lean(Yes{}, No{}); }
        else { if (want_stats) lean(No{}, Yes{}); else lean(No{}, No{}); }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int nh = 0; nh < NH; ++nh) {
          const int col = j * 32 + (M16 ? nh * 16 + m16 : l32);
          const int n = n0 + wn * WN_COLS + col;
#pragma unroll
          for (int r = 0; r < (M16 ? 8 : 16); ++r) {
            int lrow;
            float v;
            if constexpr (M16) { lrow = (r >> 2) * 16 + 4 * kb + (r & 3); v = acc4[i][r >> 2][j][nh][r & 3]; }     // mh = r >> 2
            else { lrow = (r & 3) + 8 * (r >> 2) + 4 * h; v = acc[i][j][r]; }
            v += bv[j][nh];
            v = Elt<T>::to_f(Elt<T>::from_f(v));
            if (rp[lrow] != 0xFFFFFFFFu) { a1[j][nh][0] += v; a2[j][nh][0] += v * v; }
            if (act == VG_ACT_LRELU) v = vg_lrelu(v);
            else if (act == VG_ACT_SIGMOID) v = vg_sigmoid(v);
            if (n >= p.Nout) v = 0.f;
            wt[wt_off(lrow, col)] = Elt<T>::from_f(v);
          }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (fast && !bs) {
      // whole tile, no fused sums (forward launches, plain input gradients): everything the block's store rounds read is
      // requested before the first round uses it - pixel indices and staged pieces (LDS), old values of a += launch
      // (memory) - instead of two LDS latencies in a chain (and a memory latency with +=) per round.  With the fused sums
      // the extra live registers spill and cost more than the waits.
      unsigned pixr[ROUNDS];
      Piece<T> stg[ROUNDS];
#pragma unroll
      for (int rd = 0; rd < ROUNDS; ++rd) {
        const int q = lane + 64 * rd;
        const int lrow = q / PPR, pc = q - lrow * PPR;
        pixr[rd] = rp[lrow];
        stg[rd].v = *(const decltype(stg[rd].v)*)(wt + wt_off(lrow, pc * PIECE));
      }
      const int ncol = n0 + wn * WN_COLS + (lane % PPR) * PIECE;      // 64 % PPR == 0: the same piece column in every round
      if (accumulate) {
        Piece<T> oldr[ROUNDS];
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) oldr[rd].v = *(const decltype(oldr[rd].v)*)(yg + (size_t)pixr[rd] * p.Ks + ncol);
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
#pragma unroll
          for (int e = 0; e < PIECE; ++e) stg[rd].set(e, stg[rd].get(e) + oldr[rd].get(e));
        }
      }
#pragma unroll
      for (int rd = 0; rd < ROUNDS; ++rd) *(decltype(stg[rd].v)*)(yg + (size_t)pixr[rd] * p.Ks + ncol) = stg[rd].v;
    } else
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
      const int q = lane + 64 * rd;
      const int lrow = q / PPR, pc = q - lrow * PPR;
      const unsigned pix = rp[lrow];
      const int n = n0 + wn * WN_COLS + pc * PIECE;
      if (pix == 0xFFFFFFFFu || n >= p.Ks) continue;
      Piece<T> o;
      o.v = *(const decltype(o.v)*)(wt + wt_off(lrow, pc * PIECE));
      T* dst = yg + (size_t)pix * p.Ks + n;
      if (accumulate) {
        Piece<T> old;
        old.v = *(const decltype(old.v)*)dst;
#pragma unroll
        for (int e = 0; e < PIECE; ++e) o.set(e, o.get(e) + old.get(e));
      }
      if (bs) {        // o = the gradient as stored (what the BatchNorm backward will read)
        // Straight-line code: the lane's four coefficient pieces come in with eight 16-byte LDS reads up front and the
        // launch-uniform choices (mask source, activation, mask_out) are selects.  Written with a branch per choice and
        // element, the compiler read every coefficient pair right where it was used, behind a uniform branch, and waited
        // for it: eight LDS latencies in a chain per store round, which is what the fused sums cost (not their VALU work).
        const f32x4* cf4 = (const f32x4*)(coef + wn * WN_COLS + pc * PIECE);
        float c0[PIECE], c1[PIECE], c2[PIECE], c3[PIECE];
#pragma unroll
        for (int q4 = 0; q4 < PIECE / 4; ++q4) {
          const f32x4 t0 = cf4[q4], t1 = cf4[BN / 4 + q4], t2 = cf4[2 * BN / 4 + q4], t3 = cf4[3 * BN / 4 + q4];
#pragma unroll
          for (int k = 0; k < 4; ++k) { c0[4 * q4 + k] = t0[k]; c1[4 * q4 + k] = t1[k]; c2[4 * q4 + k] = t2[k]; c3[4 * q4 + k] = t3[k]; }
        }
        const bool mask_out = p.bs.mask_out != 0;       // store dz (the masked gradient) instead of dout
        const bool use_y = p.bs.y != nullptr;
        const float slope = p.bs.act == VG_ACT_LRELU ? VG_LRELU_SLOPE : 1.f;
        Piece<T> om = o;
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
          const float xf = xr[rd].get(e);
          const float prex = xf * c0[e] + c1[e];
          const float pre = use_y ? yr[rd].get(e) : prex;
          const float dzm = o.get(e) * (pre > 0.f ? 1.f : slope);
          om.set(e, dzm);
          const float dz = mask_out ? om.get(e) : dzm;    // the sums are those of the value as stored
          bs1[e] += dz;
          bs2[e] += dz * (xf * c2[e] - c3[e]);
        }
        if (mask_out) o = om;
      }
      *(decltype(o.v)*)dst = o.v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (bs) {
    // lanes lane, lane + PPR, ... hold the same channel piece: fold them, then the WM waves of a column through LDS
#pragma unroll
    for (int m = PPR; m < 64; m <<= 1) {
#pragma unroll
      for (int e = 0; e < PIECE; ++e) { bs1[e] += __shfl_xor(bs1[e], m, 64); bs2[e] += __shfl_xor(bs2[e], m, 64); }
    }
    if (lane < PPR) {
#pragma unroll
      for (int e = 0; e < PIECE; ++e) {
        const int col = wn * WN_COLS + lane * PIECE + e;
        red[(wm * BN + col) * 2 + 0] = bs1[e];
        red[(wm * BN + col) * 2 + 1] = bs2[e];
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.Nout) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { t1 += red[(w * BN + tid) * 2]; t2 += red[(w * BN + tid) * 2 + 1]; }
      vg_stats_add(p.bs.sums, p.bs.cpad, 0, n0 + tid, mtile + phase, (double)t1, p.det);
      vg_stats_add(p.bs.sums, p.bs.cpad, 1, n0 + tid, mtile + phase, (double)t2, p.det);
    }
    __syncthreads();
  }
  VG_STAMP(3);
  if (want_stats) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int nh = 0; nh < NH; ++nh) {
        const float t1 = a1[j][nh][0] + a1[j][nh][1], t2 = a2[j][nh][0] + a2[j][nh][1];
        float s1 = t1 + __shfl_xor(t1, 32, 64), s2 = t2 + __shfl_xor(t2, 32, 64);
        if constexpr (M16) { s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64); }      // the four kb groups of a column
        if (M16 ? kb == 0 : h == 0) {
          const int col = (wn * NI + j) * 32 + (M16 ? nh * 16 + m16 : l32);
          red[(wm * BN + col) * 2 + 0] = s1;
          red[(wm * BN + col) * 2 + 1] = s2;
        }
      }
    __syncthreads();
    if (tid < BN && n0 + tid < p.Nout) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { t1 += red[(w * BN + tid) * 2]; t2 += red[(w * BN + tid) * 2 + 1]; }
      vg_stats_add(p.stats, p.Npad, 0, n0 + tid, mtile + phase, (double)t1, p.det);
      vg_stats_add(p.stats, p.Npad, 1, n0 + tid, mtile + phase, (double)t2, p.det);
    }
  }
  VG_STAMP(4);
}
#undef VG_STAMP

template <typename T, int MI, int NI, int WM, int WN, int R, int S, int EPI>
__global__ __launch_bounds__(512) void igemm3_kernel(const IgemmParams p) {
  igemm3_body<T, MI, NI, WM, WN, R, S, false, EPI>(p);
}
template <typename T, int MI, int NI, int WM, int WN, int EPI>
__global__ __launch_bounds__(512) void igemm3s2_kernel(const IgemmParams p) {
  igemm3_body<T, MI, NI, WM, WN, 2, 2, true, EPI>(p);
}
// which instantiation a launch needs (see igemm3_body)
static int igemm3_epi(const IgemmParams& p) {
  static const bool generic = getenv("VG_IGEMM3_GENERIC") != nullptr;      // A/B: the all-features kernel for every launch
  if (generic) return 0;
  const bool fwd_features = p.act != VG_ACT_NONE || p.stats != nullptr;
  if (p.bs.x != nullptr) return fwd_features ? 0 : 3;
  if (!fwd_features) return 2;
  return p.accumulate ? 0 : 1;
}

template <typename T, int MI, int NI, int WM, int WN, int R, int S>
static int launch_igemm3(const IgemmParams& p0, hipStream_t st, bool* handled) {
  typedef Igemm3Cfg<T, MI, NI, WM, WN, R, S> Cfg;
  *handled = false;
  if constexpr (Cfg::FITS) {
    IgemmParams p = p0;
    // the tap grid must be R x S, row-major, offsets all ascending or all descending, the same in every phase
    int flip = -1;
    for (int ph = 0; ph < p.nphases; ++ph) {
      int ymin = 127, xmin = 127;
      for (int t = 0; t < p.ntaps; ++t) {
        ymin = p.dy[ph][t] < ymin ? p.dy[ph][t] : ymin;
        xmin = p.dx[ph][t] < xmin ? p.dx[ph][t] : xmin;
      }
      p.dymin[ph] = (signed char)ymin; p.dxmin[ph] = (signed char)xmin;
      bool asc = true, desc = true;
      for (int t = 0; t < p.ntaps; ++t) {
        const int yo = p.dy[ph][t] - ymin, xo = p.dx[ph][t] - xmin;
        asc = asc && yo == t / S && xo == t % S;
        desc = desc && yo == R - 1 - t / S && xo == S - 1 - t % S;
      }
      const int f = asc ? 0 : (desc ? 1 : -2);
      if (f < 0 || (flip >= 0 && f != flip)) return VG_OK;      // not handled: the caller falls back
      flip = f;
    }
    p.flip = flip;
    if ((size_t)p.N * p.H * p.W * p.Cs * Cfg::EB >= ((size_t)1 << 32)) return VG_OK;   // 32-bit piece offsets
    p.dbg = g_dbg;
#ifdef VG_ABLATE
    p.splitws = (float*)g_stamp_buf;
#endif
    p.tilesX = (int)vg_ceil_div(p.OWl, Cfg::TW);
    p.tilesY = (int)vg_ceil_div(p.OHl, Cfg::TH);
    p.mtiles = p.N * p.tilesX * p.tilesY;
    p.ntiles = p.Npad / Cfg::BN;
    pack_taps(p);
    snprintf(g_last_kernel, sizeof g_last_kernel, "igemm3<%s,%d,%d,%d,%d,%dx%d>", tname<T>(), MI, NI, WM, WN, R, S);
    const dim3 grid(p.mtiles * p.ntiles, p.nphases);
#define VG_IGEMM3_LAUNCH(E)                                                                \
  do {                                                                                     \
    auto kern = igemm3_kernel<T, MI, NI, WM, WN, R, S, E>;                                 \
    VG_HIP(vg_lds_attr((const void*)kern, 160 * 1024));                                    \
    hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), Cfg::LDS, st, p);                        \
  } while (0)
    switch (igemm3_epi(p)) {
      case 1: VG_IGEMM3_LAUNCH(1); break;
      case 2: VG_IGEMM3_LAUNCH(2); break;
      case 3: VG_IGEMM3_LAUNCH(3); break;
      default: VG_IGEMM3_LAUNCH(0); break;
    }
#undef VG_IGEMM3_LAUNCH
    VG_LAUNCH_CHECK();
    *handled = true;
  }
  return VG_OK;
}

// 4x4 stride-2 pad-1 forward on the parity planes (igemm3_body<..., S2 = true>)
static bool g_igemm3_s2 = true;
template <typename T, int MI, int NI, int WM, int WN>
static int launch_igemm3s2(const IgemmParams& p0, hipStream_t st, bool* handled) {
  typedef Igemm3Cfg<T, MI, NI, WM, WN, 2, 2> Cfg;
  *handled = false;
  if constexpr (Cfg::FITS) {
    IgemmParams p = p0;
    const int cch = p.C / Cfg::KC;
    if (cch < 1 || (cch & (cch - 1)) != 0) return VG_OK;                        // power-of-two chunk count (plane = chunk >> lgc)
    if ((size_t)p.N * p.H * p.W * p.Cs * Cfg::EB >= ((size_t)1 << 31)) return VG_OK;   // signed 32-bit piece offsets
    p.lgc = 0;
    while ((1 << p.lgc) < cch) ++p.lgc;
    p.ntaps = 4;
    for (int t = 0; t < 4; ++t) { p.dy[0][t] = (signed char)(t >> 1); p.dx[0][t] = (signed char)(t & 1); }
    p.dymin[0] = 0; p.dxmin[0] = 0;
    p.flip = 0;
    p.dbg = g_dbg;
    p.tilesX = (int)vg_ceil_div(p.OWl, Cfg::TW);
    p.tilesY = (int)vg_ceil_div(p.OHl, Cfg::TH);
    p.mtiles = p.N * p.tilesX * p.tilesY;
    p.ntiles = p.Npad / Cfg::BN;
    pack_taps(p);
    snprintf(g_last_kernel, sizeof g_last_kernel, "igemm3s2<%s,%d,%d,%d,%d>", tname<T>(), MI, NI, WM, WN);
    const dim3 grid(p.mtiles * p.ntiles, 1);
#define VG_IGEMM3S2_LAUNCH(E)                                                              \
  do {                                                                                     \
    auto kern = igemm3s2_kernel<T, MI, NI, WM, WN, E>;                                     \
    VG_HIP(vg_lds_attr((const void*)kern, 160 * 1024));                                    \
    hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), Cfg::LDS, st, p);                        \
  } while (0)
    // a forward kernel: a plain convolution (no activation, no statistics: the encoder's downsampling layers) is the forward
    // instantiation too - the all-features one spills (40 bytes of scratch per lane) and ran them at 570-1000 TFLOP/s
    const int epi = igemm3_epi(p);
    if (epi == 1 || (epi == 2 && !p.accumulate)) VG_IGEMM3S2_LAUNCH(1);
    else if (epi == 3) VG_IGEMM3S2_LAUNCH(3);        // input gradient of a transposed convolution with the fused BatchNorm-backward sums
    else VG_IGEMM3S2_LAUNCH(0);
#undef VG_IGEMM3S2_LAUNCH
    VG_LAUNCH_CHECK();
    *handled = true;
  }
  return VG_OK;
}
template <typename T> static int dispatch_igemm3s2(const IgemmParams& p, hipStream_t st, bool* handled) {
  *handled = false;
  if (!g_igemm3_s2 || p.ntaps != 16 || p.nphases != 1 || (p.H & 1) || (p.W & 1) || p.OHl != p.H / 2 || p.OWl != p.W / 2) return VG_OK;
  for (int t = 0; t < 16; ++t)
    if (p.dy[0][t] != t / 4 - 1 || p.dx[0][t] != t % 4 - 1) return VG_OK;      // pad 1, row-major taps
  if (p.Npad % 256 == 0) return launch_igemm3s2<T, 4, 2, 2, 4>(p, st, handled);
  if (p.Npad == 128) {
    if constexpr (Igemm3Cfg<T, 4, 2, 4, 2, 2, 2>::FITS) return launch_igemm3s2<T, 4, 2, 4, 2>(p, st, handled);
    else return launch_igemm3s2<T, 2, 2, 4, 2>(p, st, handled);
  }
  if (p.Npad == 64) return launch_igemm3s2<T, 2, 2, 8, 1>(p, st, handled);
  return VG_OK;
}

static int g_conv_kernel = 3;
static bool g_conv_kernel_is_old() { return g_conv_kernel < 3; }

// =============================================================================================
// igemm4: weights-stationary persistent kernel for the 64 -> 64 channel 3x3 stride-1 gathers (the full-resolution
// convolutions and their input gradients).  These layers hold the biggest tensors of the step and sit closest to the
// HBM roofline, but in igemm3 every wave re-read its weight fragments from LDS on every K step (3 fragment reads per
// 2 MFMAs: LDS-bound at ~2/3 of the MFMA rate), every block paid a prologue of one full memory latency, and the
// epilogue of a block overlapped with nothing.
// Here the WHOLE weight image (9 taps x 64 x 64 bf16 = 72 fragments = 288 registers) lives in the registers of every
// wave, loaded once per block: a block is 4 waves (one per SIMD, so each may use the SIMD's 512 registers), it is
// persistent and walks over 8 x 32 pixel tiles, a wave owning 2 pixel rows x 64 output channels (4 accumulators).
// The halo image of tile i+1 arrives by LDS-DMA (out-of-image pieces from a block of zeros) while tile i is computed,
// so the K loop has no weight traffic, no barriers and no waits: 2 LDS fragment reads per 4 MFMAs.  The output stores
// of tile i drain under the MFMAs of tile i+1.  BatchNorm statistics stay in registers until the block ends.
// Tiles are handed out so that the 32 blocks of an XCD work on 32 consecutive tiles (4 tile rows of a 256-wide image)
// at a time: the halo rows two tiles share come out of that XCD's L2.
// =============================================================================================
// igemm4's timing switches perturb its schedule: only in a -DVG_ABLATE=2 build (=1: the cycle stamps alone)
#define VG_DBG4(bit) (VG_ABLATE_LEVEL >= 2 && (p.dbg & (bit)))
#ifndef VG_IGEMM4_PD
#define VG_IGEMM4_PD 2
#endif
#ifndef VG_HALO_SPLIT
#define VG_HALO_SPLIT 1
#endif
// PH ("a phase per wave"): stride-2 transposed gathers (transposed convolutions, input gradients of stride-2
// convolutions) = four output phases (py, px) of 2x2 taps over the same input pixels.  The four waves of a block take
// one phase each, with that phase's 2x2 x C x 64 weights in registers, over the SAME 2 x 32 input pixel tile: one halo
// image (tile + one pixel all round) serves all four phases, where one launch dimension per phase fetched it four times.
template <int R, int S, int CCH, bool PH>
struct Igemm4Cfg {
  static constexpr int NT = 256, TW = 32, TH = PH ? 2 : 8, BM = TH * TW, BN = 64;
  static constexpr int NTAPS = R * S, HS = 35, HSB = HS * 64, HH = TH + (PH ? 2 : R - 1);
  static constexpr int WCOLS = TW + (PH ? 2 : S - 1);           // halo columns that are read
  static constexpr int CHB = HH * HSB;                          // one channel chunk of a halo image
  static constexpr int NPIECE = CCH * HH * HS * 4;              // 16-byte pieces per tile
  static constexpr int HRN = (NPIECE + NT - 1) / NT;            // DMA rounds per tile
  static constexpr int HBYTES = HRN * NT * 16;
  static constexpr int WSTRIDE = BN + 8;                        // staging row: 64 channels + 16 B (bank spread)
  static constexpr int WTILE = 2 * 32 * WSTRIDE * 2;            // per-wave output staging: 2 rows x 32 px x 64 ch bf16
  static constexpr int OFF_STAGE = 2 * HBYTES;
  static constexpr int OFF_TAB = OFF_STAGE + 4 * WTILE;         // [HRN][NT] halo piece table (offset, position)
  static constexpr int LDS = OFF_TAB + HRN * NT * 8;
  static constexpr bool FITS = LDS <= 160 * 1024 && NTAPS * CCH * 16 <= 320 && 4 * BN * 2 * 8 <= 4 * WTILE;
};

template <int R, int S, int CCH, bool LR, bool ST, bool ACC, bool PH>
__global__ __launch_bounds__(256) void igemm4_kernel(const IgemmParams p) {
  typedef Igemm4Cfg<R, S, CCH, PH> Cfg;
  static_assert(!PH || (R == 2 && S == 2 && !ACC), "a phase per wave: 2x2 taps, plain stores");
  typedef bf16_t T;
  constexpr int NT = Cfg::NT, TW = Cfg::TW, TH = Cfg::TH, BN = Cfg::BN, NTAPS = Cfg::NTAPS;
  constexpr int HS = Cfg::HS, HSB = Cfg::HSB, HH = Cfg::HH, CHB = Cfg::CHB, HRN = Cfg::HRN, HBYTES = Cfg::HBYTES;
  constexpr int PIECE = 8, WTILE = Cfg::WTILE, WSTRIDE = Cfg::WSTRIDE;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = pixel-row pair of the tile, or (PH) the phase
  // v_mfma_f32_16x16x32_bf16 (round 3; the chip holds a higher clock on it than on 32x32x16): lane = row / column m16 of
  // a 16-wide tile, kb = its 16-byte k slot of the 64-byte channel chunk
  const int m16 = lane & 15, kb = lane >> 4;
  const int phase = PH ? wave : (int)blockIdx.y;
  const int py = phase / p.out_s, px = phase % p.out_s;
  const int tpi = p.tilesX * p.tilesY;
  // origin of the halo image relative to the tile: the smallest tap offset (PH: over all four phases)
  int dymin = p.dymin_i[phase], dxmin = p.dxmin_i[phase];
  if (PH) {
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) { dymin = min(dymin, p.dymin_i[ph]); dxmin = min(dxmin, p.dxmin_i[ph]); }
  }
  const int rowoff = (PH ? 0 : 2 * wave) + (p.dymin_i[phase] - dymin);     // halo row of this wave's first tap, first pixel row
  const int coloff = p.dxmin_i[phase] - dxmin;
  const bool flip = p.flip != 0;
  const unsigned char* __restrict__ xg = (const unsigned char*)p.x;
  const unsigned smem0 = lds_addr_of(smem);

  // ---- all weights: registers, once per block.  fb[tap][cc][nt]: k = cc*32 + 8*kb.., channel nt*16 + m16
  u32x4 fb[NTAPS][CCH][4];
  {
    const size_t ktot = (size_t)NTAPS * p.C;
    const T* wg = (const T*)p.w + ((size_t)phase * p.Npad + m16) * ktot;
#pragma unroll
    for (int tap = 0; tap < NTAPS; ++tap)
#pragma unroll
      for (int cc = 0; cc < CCH; ++cc)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          fb[tap][cc][nt] = *(const u32x4*)(wg + (size_t)nt * 16 * ktot + (size_t)tap * p.C + cc * 32 + kb * PIECE);
  }
  // ---- halo pieces of this thread (the same in every tile): piece q = rd*NT + tid of [CCH][HH*HS pixels][4 slots].
  // The table lives in LDS: 2*HRN more live registers per lane would spill, and a spill reload is a vector memory
  // load that queues behind the DMA it sits between.
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  i32x2* ptab = (i32x2*)(smem + Cfg::OFF_TAB) + tid;      // [rd * NT]: x = byte offset from the tile's origin pixel
#pragma unroll                                           // (ty0 + dymin, tx0 + dxmin), chunk and slot included;
  for (int rd = 0; rd < HRN; ++rd) {                      // y = (hy << 8) | hx, or -1: padding, never read
    const int q = rd * NT + tid;
    const int cc = q / (HH * HS * 4), rem = q - cc * (HH * HS * 4);
    const int pix = rem >> 2, ps = rem & 3;
    const int hy = pix / HS, hx = pix - hy * HS;
    const bool ok = q < Cfg::NPIECE && hx < Cfg::WCOLS;
    ptab[rd * NT] = i32x2{((hy * p.W + hx) * p.Cs) * 2 + cc * 64 + (igemm3_swz<true>(ps, hx) << 4), ok ? ((hy << 8) | hx) : -1};
  }
  // tile k of this block: with the full grid of 256 blocks, the 32 blocks of an XCD (blockIdx % 8) take 32 consecutive tiles
  const bool xcd_order = gridDim.x == 256;
  auto tile_of = [&](int k) { return xcd_order ? ((k * 8 + (int)(blockIdx.x & 7)) * 32 + (int)(blockIdx.x >> 3)) : k * (int)gridDim.x + (int)blockIdx.x; };
  auto tile_origin = [&](int tile, int& img, int& ty0, int& tx0) {
    img = tile / tpi;
    const int trem = tile - img * tpi;
    const int ty = trem / p.tilesX;
    ty0 = ty * TH; tx0 = (trem - ty * p.tilesX) * TW;
  };
  // A fragment addresses: column term per horizontal tap (pixels 0..15 of the row; +1024 = pixels 16..31: the slot swizzle
  // bit is the same); rows are immediates
  int ct[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int hx = m16 + (flip ? S - 1 - s : s) + coloff;
    ct[s] = hx * 64 + (igemm3_swz<true>(kb, hx) << 4) + rowoff * HSB;
  }
  float bv[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    bv[nt] = p.bias != nullptr ? p.bias[nt * 16 + m16] : 0.f;
  }
  double st1[4] = {0.0, 0.0, 0.0, 0.0}, st2[4] = {0.0, 0.0, 0.0, 0.0};
  T* __restrict__ yg = (T*)p.y;
  T* wt = (T*)(smem + Cfg::OFF_STAGE + wave * WTILE);      // [2 rows][32 px][WSTRIDE]

  // One halo round of tile `tile` into buffer `buf`: every lane issues (padding pieces fetch zeros into the buffer's
  // padding), so a wave's DMA count per tile is always HRN.  `e` is this round's table entry, read a round ahead.
  i32x2 e = ptab[0];
  vg_srd_t hsrd = vg_make_srd(xg);            // buffer resource at the origin pixel of the tile being fetched
  int hiy0 = 0, hix0 = 0;
  auto halo_begin = [&](int tile) {
    int img, ty0, tx0;
    tile_origin(tile, img, ty0, tx0);
    hiy0 = ty0 + dymin; hix0 = tx0 + dxmin;
    hsrd = vg_make_srd(xg + ((long)((long)img * p.H + hiy0) * p.W + hix0) * p.Cs * 2);
  };
  // A halo round is dealt out over two MFMA gaps of the K loop (bounds test and offset select | DMA issue + next table
  // entry): one wave per SIMD hides ~6 instructions behind a pair of MFMAs, and the round as one block stalled the matrix
  // pipe for the rest (43 instructions with the short-circuit && of round 2, which compiled to nested divergent branches
  // around a 64-bit select against the block of zeros; 13 now, through a buffer resource: lds_dma16_buf).  The empty asm
  // statement pins the first part to the gap it is written in.
  unsigned h_off = 0xFFFFFFFFu;
  auto halo_a = [&]() {
    const int hy = e.y >> 8, hx = e.y & 255;
    const bool in = (e.y >= 0) & ((unsigned)(hiy0 + hy) < (unsigned)p.H) & ((unsigned)(hix0 + hx) < (unsigned)p.W);      // no short-circuit
    h_off = in ? (unsigned)e.x : 0xFFFFFFFFu;
    asm volatile("" : "+v"(h_off));
  };
  auto halo_c = [&](int rd, int buf) {
    lds_dma16_buf(hsrd, h_off, smem0 + buf * HBYTES + (rd * NT + wave * 64) * 16);
    e = ptab[(rd + 1 == HRN ? 0 : rd + 1) * NT];
  };
  auto halo_round = [&](int rd, int buf) { halo_a(); halo_c(rd, buf); };      // the prologue's rounds
  // The output rows of a tile go to memory during the K loop of the NEXT tile, one 1 KB store every other step, and
  // the halo of the next tile arrives the same way: a block that issued its loads and stores in bursts stalled at
  // issue (the memory queues of a CU are short) with the matrix cores idle, and since every block runs the same
  // schedule the whole chip alternated between an HBM-bound and an MFMA-bound phase.
  auto stage_piece = [&](int sidx) {          // 16 bytes of this lane of deferred store sidx (row sidx / 4)
    return *(const u32x4*)(wt + (sidx >> 2) * 32 * WSTRIDE + ((lane + 64 * (sidx & 3)) >> 3) * WSTRIDE + (lane & 7) * PIECE);
  };
  T* pdst = nullptr;                          // previous tile: first output row of this wave, this lane's piece
  auto store_ptr = [&](int sidx) { return pdst + (size_t)(sidx >> 2) * p.out_s * p.OW * BN + (sidx & 3) * 8 * p.out_s * BN; };
  auto store_piece = [&](int sidx, u32x4 v, u32x4 old) {
    if (ACC) {               // y += conv: bf16(bf16(conv) + y), the rounding order of the other kernels' read-modify-write
      Piece<T> o, q;
      o.v = __builtin_bit_cast(bf16x8, v); q.v = __builtin_bit_cast(bf16x8, old);
#pragma unroll
      for (int e_ = 0; e_ < PIECE; ++e_) o.set(e_, o.get(e_) + q.get(e_));
      v = __builtin_bit_cast(u32x4, o.v);
    }
    if (!VG_DBG4(32)) *(u32x4*)store_ptr(sidx) = v;
  };
  // ACC: the old values of the previous tile's rows are requested in one burst when the K loop starts and the stores
  // sit in its second half; the end-of-tile wait then lets the stores issued after the last halo round stay in flight
  // (all variants since round 3: with the stores in the first half of the K loop the end-of-tile wait for the halo also
  // waited for every store of the previous tile, ~15 % of a block's time in the s_memtime stamps)
  constexpr int ST0 = CCH * NTAPS * 2 - 18 < 18 ? CCH * NTAPS * 2 - 18 : 18;      // K steps: CCH * NTAPS * 2

#ifdef VG_ABLATE   // s_memtime ticks per stage, summed over the block's tiles (wave 0): tests/gpu_igemm4_stamps.py
  unsigned long long tacc[4] = {0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
  const unsigned long long tstart = tlast;
#define VG_T(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define VG_T(i) do {} while (0)
#endif
  int tile = tile_of(0);
  if (tile < p.mtiles) {
    halo_begin(tile);
#pragma unroll
    for (int rd = 0; rd < HRN; ++rd) halo_round(rd, 0);
  }
  wait_vmcnt<0>();
  bool has_prev = false;
  int buf = 0;
  for (int k = 0; tile < p.mtiles; ++k, buf ^= 1) {
    const int next = tile_of(k + 1);
    // The halo rounds of the K loop are unconditional (a uniform `has_next` test cost four instructions and a branch in
    // front of every round): the block's last tile fetches its own halo again, into the buffer nobody reads.
    const bool has_next = next < p.mtiles;
    int img, ty0, tx0;
    tile_origin(tile, img, ty0, tx0);
    halo_begin(has_next ? next : tile);
    __syncthreads();                   // halo(tile) has landed for every wave; nobody reads the other buffer any more
    VG_T(0);
    f32x4 acc[2][2][4];                // [pixel row][16-pixel half][16-channel tile]; first written with C = 0
    const unsigned char* hb = smem + buf * HBYTES;
    // K loop: 36 steps (tap, channel chunk, 16-pixel half mh = t & 1) of 8 MFMAs (2 rows x 4 channel tiles, all 32 k of
    // the chunk at once), fully unrolled; the two A fragments of step t+PD are requested before the MFMAs of step t (one
    // wave per SIMD: nobody else hides the LDS latency)
    constexpr int KSTEPS = CCH * NTAPS * 2;
    static_assert(2 * HRN <= KSTEPS - 8 && ST0 + 18 <= KSTEPS, "halo rounds on the even steps, done well before the loop ends; 8 stores fit");
    auto a_addr = [&](int t) {
      const int cc = t / (NTAPS * 2), tap = (t / 2) % NTAPS, mh = t & 1;
      const int r = tap / S, s = tap % S;
      return ct[s] + cc * CHB + (flip ? R - 1 - r : r) * HSB + mh * 1024;
    };
    constexpr int PD = VG_IGEMM4_PD;      // prefetch distance in K steps
    constexpr int AFRAGS = 64;
    u32x4 fa[PD + 1][2];
    u32x4 sreg = u32x4{0, 0, 0, 0};
    u32x4 oreg[8];
#pragma unroll
    for (int sidx = 0; sidx < 8; ++sidx) oreg[sidx] = u32x4{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < PD; ++t) {
      const int va = a_addr(t);
      fa[t][0] = *(const u32x4*)(hb + va); fa[t][1] = *(const u32x4*)(hb + va + HSB);
    }
    // The first AFRAGS weight fragments live in accumulation registers, the rest (and the accumulators) in ordinary
    // ones: 288 + 64 registers do not fit either file alone.  With the builtin the compiler kept the weights in
    // accumulation registers but copied every fragment into the same four ordinary ones in front of its MFMAs (it never
    // names an accumulation register as a multiplicand), and those copies had to wait for the previous MFMA to read them.
    auto mfma = [&](int t, int i, int nt) {
      const int cc = t / (NTAPS * 2), tap = (t / 2) % NTAPS, mh = t & 1;
      const bool in_a = (tap * CCH + cc) * 4 + nt < AFRAGS;
      static_assert(AFRAGS >= 4, "the first step's fragments are in accumulation registers");
      if (t < 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc[i][mh][nt]) : "v"(fa[t][i]), "a"(fb[tap][cc][nt]));
      else if (in_a) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][mh][nt]) : "v"(fa[t % (PD + 1)][i]), "a"(fb[tap][cc][nt]));
      else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][mh][nt]) : "v"(fa[t % (PD + 1)][i]), "v"(fb[tap][cc][nt]));
    };
    static_assert(PD >= 2, "steps 0 and 1 (the two pixel halves) read fa[0] and fa[1]");
    // One K step = 8 MFMAs (16 cycles each; an MFMA holds the wave's issue for 8 of them).  The step's other work is
    // dealt out over the first four gaps: the two A-fragment prefetches in the first two, a halo round or a staging read
    // in the third, a deferred store in the fourth; the other four gaps stay empty.
#pragma clang loop unroll(full)
    for (int t = 0; t < KSTEPS; ++t) {
      const int vp = t + PD < KSTEPS ? a_addr(t + PD) : 0;
      __builtin_amdgcn_sched_barrier(0);
      mfma(t, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (t + PD < KSTEPS) fa[(t + PD) % (PD + 1)][0] = *(const u32x4*)(hb + vp);
      __builtin_amdgcn_sched_barrier(0);
      mfma(t, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (t + PD < KSTEPS) fa[(t + PD) % (PD + 1)][1] = *(const u32x4*)(hb + vp + HSB);
      __builtin_amdgcn_sched_barrier(0);
      mfma(t, 0, 1);
      mfma(t, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (VG_HALO_SPLIT) { if (t % 2 == 0 && t / 2 < HRN && !VG_DBG4(1)) halo_a(); }
      else if (t % 2 == 0 && t / 2 < HRN && !VG_DBG4(1)) halo_round(t / 2, buf ^ 1);
      if (ACC && t == 1) {
        if (has_prev) {
#pragma unroll
          for (int sidx = 0; sidx < 8; ++sidx) oreg[sidx] = *(const u32x4*)store_ptr(sidx);
        }
      }
      if (t >= ST0 && (t - ST0) % 2 == 1 && (t - ST0) / 2 < 8) { if (has_prev) sreg = stage_piece((t - ST0) / 2); }
      __builtin_amdgcn_sched_barrier(0);
      mfma(t, 0, 2);
      mfma(t, 1, 2);
      __builtin_amdgcn_sched_barrier(0);
      if (t >= ST0 + 2 && (t - ST0) % 2 == 0 && (t - ST0 - 2) / 2 < 8) { if (has_prev) store_piece((t - ST0 - 2) / 2, sreg, oreg[(t - ST0 - 2) / 2]); }
      if (VG_HALO_SPLIT && t % 2 == 1 && t / 2 < HRN && !VG_DBG4(1)) halo_c(t / 2, buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma(t, 0, 3);
      mfma(t, 1, 3);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15");     // MFMA result -> VALU / LDS read (no hazard recognizer for inline MFMAs)
    __builtin_amdgcn_sched_barrier(0);
    VG_T(1);
    // ---- epilogue (every tile is whole, all 64 channels are real: the launcher's conditions): BatchNorm statistics of
    // the bf16-rounded pre-activation, activation; both pixel rows into the wave's staging tile (stored next tile)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (VG_DBG4(64)) continue;
      // accumulator (mh, nt), register r = pixel mh*16 + 4*kb + r of row i, channel nt*16 + m16.  The 144-byte staging
      // rows put the rows 4 apart that a store's 32-lane groups hold (kb, kb + 1) on different banks.
      T* wl = wt + i * 32 * WSTRIDE + 4 * kb * WSTRIDE + m16;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        f32x2 a1 = f32x2{0.f, 0.f}, a2 = f32x2{0.f, 0.f};
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const int row = mh * 16 + r;
            f32x2 v = f32x2{acc[i][mh][nt][r], acc[i][mh][nt][r + 1]} + bv[nt];
            bf16x2_t q = __builtin_convertvector(v, bf16x2_t);
            if (ST && !LR) { a1[0] = dot2_sum(q, a1[0]); a2[0] = dot2_sq(q, a2[0]); }      // (a1[1], a2[1] stay zero)
            if (LR) v = __builtin_convertvector(q, f32x2);
            if (ST && LR) { a1 += v; a2 += v * v; }
            if (LR) {
              const f32x2 w = v * VG_LRELU_SLOPE;
              f32x2 m;
              asm("v_max_f32 %0, %1, %2" : "=v"(m[0]) : "v"(v[0]), "v"(w[0]));
              asm("v_max_f32 %0, %1, %2" : "=v"(m[1]) : "v"(v[1]), "v"(w[1]));
              q = __builtin_convertvector(m, bf16x2_t);
            }
            wl[row * WSTRIDE + nt * 16] = q[0];
            wl[(row + 1) * WSTRIDE + nt * 16] = q[1];
          }
        if (ST) { st1[nt] += (double)(a1[0] + a1[1]); st2[nt] += (double)(a2[0] + a2[1]); }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    pdst = yg + ((((size_t)img * p.OH + (ty0 + (PH ? 0 : 2 * wave)) * p.out_s + py) * p.OW + tx0 * p.out_s + px) * BN +
                 (lane >> 3) * p.out_s * BN + (lane & 7) * PIECE);
    has_prev = !VG_DBG4(64);
    // halo rounds and stores were issued in the first 2/3 of the K loop: (nearly) free by now, and it makes the barrier
    // at the top of the next iteration mean "the next halo is in LDS"
    VG_T(2);
    {
      // stores issued after the last halo round (K step 2*HRN-1): they may stay in flight
      constexpr int LATE = [] { int n = 0; for (int s_ = 0; s_ < 8; ++s_) n += ST0 + 2 + 2 * s_ >= 2 * HRN; return n; }();      // (a lower bound without the split)      // the last DMA goes out in step 2 * HRN - 1
      if (has_prev && VG_ABLATE_LEVEL < 2) wait_vmcnt<LATE>(); else wait_vmcnt<0>();
    }
    VG_T(3);
    tile = next;
  }
#ifdef VG_ABLATE
  if (VG_DBG(8) && p.splitws && tid == 0) {
    unsigned long long* stamps = (unsigned long long*)p.splitws + (size_t)blockIdx.x * 8;
    for (int i = 0; i < 4; ++i) stamps[i] = tacc[i];
    stamps[4] = __builtin_amdgcn_s_memtime() - tstart;
    stamps[5] = tstart;
  }
#endif
#undef VG_T
  if (has_prev) {
#pragma unroll
    for (int sidx = 0; sidx < 8; ++sidx) {
      u32x4 old = u32x4{0, 0, 0, 0};
      if (ACC) old = *(const u32x4*)store_ptr(sidx);
      store_piece(sidx, stage_piece(sidx), old);
    }
  }
  if (ST) {
    __syncthreads();
    double* red = (double*)(smem + Cfg::OFF_STAGE);      // [4 waves][BN][2]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      double s1 = st1[nt] + __shfl_xor(st1[nt], 32, 64), s2 = st2[nt] + __shfl_xor(st2[nt], 32, 64);
      s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);      // the four k-slot groups hold the same column
      if (kb == 0) { red[(wave * BN + nt * 16 + m16) * 2 + 0] = s1; red[(wave * BN + nt * 16 + m16) * 2 + 1] = s2; }
    }
    __syncthreads();
    if (tid < BN && tid < p.Nout) {
      double t1 = 0.0, t2 = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) { t1 += red[(w * BN + tid) * 2]; t2 += red[(w * BN + tid) * 2 + 1]; }
      vg_stats_add(p.stats, p.Npad, 0, tid, blockIdx.x + phase, t1, p.det);
      vg_stats_add(p.stats, p.Npad, 1, tid, blockIdx.x + phase, t2, p.det);
    }
  }
}

// =============================================================================================
// igemm4c: the weights-stationary kernel for 128 output channels (the 128 -> 128 3x3 convolutions and their input
// gradients).  9 x 128 x 128 weights do not fit one wave's registers, so the four waves of a block split the OUTPUT
// CHANNELS: wave w keeps the 72 fragments of channels 32w .. 32w+31 (all taps, all 128 input channels) and computes
// them for every pixel of a 4 x 32 tile (4 accumulators).  All four waves read the same A fragments: one LDS read
// per MFMA per wave, half of what the LDS array delivers.  Halo prefetch, deferred output stores, statistics and tile
// hand-out as in igemm4; the halo piece table is packed into 4 bytes per entry to fit two 54 KB halo images.
// =============================================================================================
#ifndef VG_IGEMM4C_PD
#define VG_IGEMM4C_PD 3
#endif
template <int CCH>
struct Igemm4cCfg {
  static constexpr int R = 3, S = 3, NT = 256, TW = 32, TH = 4, BN = 128, NTAPS = 9;
  static constexpr int HS = 35, HSB = HS * 64, HH = TH + R - 1, CHB = HH * HSB;
  static constexpr int NPIECE = CCH * HH * HS * 4;
  static constexpr int HRN = (NPIECE + NT - 1) / NT;
  static constexpr int HBYTES = NPIECE * 16;                    // exact: lanes past the image's end do not write
  static constexpr int WTILE = TH * 32 * 32 * 2;                // per-wave output staging: 4 rows x 32 px x 32 ch bf16
  static constexpr int OFF_STAGE = 2 * HBYTES;
  static constexpr int OFF_TAB = OFF_STAGE + 4 * WTILE;         // [HRN][NT] packed halo piece table
  static constexpr int LDS = OFF_TAB + HRN * NT * 4;
  static constexpr int KSTEPS = CCH * NTAPS * 2;
  static constexpr bool FITS = LDS <= 160 * 1024 && NTAPS * CCH * 2 * 4 <= 320;
};

template <int CCH, bool LR, bool ST, bool ACC>
__global__ __launch_bounds__(256) void igemm4c_kernel(const IgemmParams p) {
  typedef Igemm4cCfg<CCH> Cfg;
  typedef bf16_t T;
  constexpr int R = 3, S = 3, NT = Cfg::NT, TW = Cfg::TW, TH = Cfg::TH, BN = Cfg::BN, NTAPS = Cfg::NTAPS;
  constexpr int HS = Cfg::HS, HSB = Cfg::HSB, HH = Cfg::HH, CHB = Cfg::CHB, HRN = Cfg::HRN, HBYTES = Cfg::HBYTES;
  constexpr int PIECE = 8, WTILE = Cfg::WTILE, KSTEPS = Cfg::KSTEPS;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = block of 32 output channels
  const int m16 = lane & 15, kb = lane >> 4;                      // v_mfma_f32_16x16x32_bf16 lane roles (see igemm4)
  const int tpi = p.tilesX * p.tilesY;
  const int dymin = p.dymin_i[0], dxmin = p.dxmin_i[0];
  const bool flip = p.flip != 0;
  const unsigned char* __restrict__ xg = (const unsigned char*)p.x;
  const unsigned smem0 = lds_addr_of(smem);

  // ---- this wave's weights: fb[tap][cc][nt]: channel 32*wave + 16*nt + m16, k = cc*32 + 8*kb..
  u32x4 fb[NTAPS][CCH][2];
  {
    const size_t ktot = (size_t)NTAPS * p.C;
    const T* wg = (const T*)p.w + ((size_t)32 * wave + m16) * ktot;
#pragma unroll
    for (int tap = 0; tap < NTAPS; ++tap)
#pragma unroll
      for (int cc = 0; cc < CCH; ++cc)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          fb[tap][cc][nt] = *(const u32x4*)(wg + (size_t)nt * 16 * ktot + (size_t)tap * p.C + cc * 32 + kb * PIECE);
  }
  // ---- halo piece table, 4 bytes per entry (two 54 KB halo images leave no room for more): byte offset of the piece from
  // the halo's origin pixel (21 bits: the launcher checks HH * W * Cs * 2 < 2^21) | hy << 21 | hx << 24 | inside the image
  // columns << 30 | exists << 31.  (Round 2 packed (hy, hx, chunk, slot) and rebuilt the offset with two 32-bit multiplies
  // per round: 52 instructions between two MFMAs for every halo round, a quarter of the K loop.)
  unsigned* ptab = (unsigned*)(smem + Cfg::OFF_TAB) + tid;
  {
    const int wcs2 = p.W * p.Cs * 2, cs2 = p.Cs * 2;
#pragma unroll
    for (int rd = 0; rd < HRN; ++rd) {
      const int q = rd * NT + tid;
      const int cc = q / (HH * HS * 4), rem = q - cc * (HH * HS * 4);
      const int pix = rem >> 2, ps = rem & 3;
      const int hy = pix / HS, hx = pix - hy * HS;
      const unsigned off = (unsigned)(hy * wcs2 + hx * cs2 + cc * 64 + igemm3_swz<true>(ps, hx) * 16) & 0x1FFFFFu;
      ptab[rd * NT] = off | (unsigned)hy << 21 | (unsigned)hx << 24 | (unsigned)(hx < TW + S - 1) << 30 |
                      (unsigned)(q < Cfg::NPIECE) << 31;
    }
  }
  const bool xcd_order = gridDim.x == 256;
  auto tile_of = [&](int k) { return xcd_order ? ((k * 8 + (int)(blockIdx.x & 7)) * 32 + (int)(blockIdx.x >> 3)) : k * (int)gridDim.x + (int)blockIdx.x; };
  auto tile_origin = [&](int tile, int& img, int& ty0, int& tx0) {
    img = tile / tpi;
    const int trem = tile - img * tpi;
    const int ty = trem / p.tilesX;
    ty0 = ty * TH; tx0 = (trem - ty * p.tilesX) * TW;
  };
  int ct[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int hx = m16 + (flip ? S - 1 - s : s);
    ct[s] = hx * 64 + (igemm3_swz<true>(kb, hx) << 4);
  }
  float bv[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) bv[nt] = p.bias != nullptr ? p.bias[32 * wave + 16 * nt + m16] : 0.f;
  double st1[2] = {0.0, 0.0}, st2[2] = {0.0, 0.0};
  T* __restrict__ yg = (T*)p.y;
  T* wt = (T*)(smem + Cfg::OFF_STAGE + wave * WTILE);      // [4 rows][32 px][32 ch]

  unsigned e = ptab[0];
  vg_srd_t hsrd = vg_make_srd(xg);
  int hiy0 = 0, hix0 = 0;
  auto halo_begin = [&](int tile) {
    int img, ty0, tx0;
    tile_origin(tile, img, ty0, tx0);
    hiy0 = ty0 + dymin; hix0 = tx0 + dxmin;
    hsrd = vg_make_srd(xg + ((long)((long)img * p.H + hiy0) * p.W + hix0) * p.Cs * 2);
  };
  // two parts over two MFMA gaps, through a buffer resource: see igemm4
  unsigned h_off = 0xFFFFFFFFu;
  auto halo_a = [&]() {
    const int hy = (e >> 21) & 7, hx = (e >> 24) & 63;
    const bool in = ((e >> 30 & 1) != 0) & ((unsigned)(hiy0 + hy) < (unsigned)p.H) & ((unsigned)(hix0 + hx) < (unsigned)p.W);
    h_off = in ? (e & 0x1FFFFFu) : 0xFFFFFFFFu;
    asm volatile("" : "+v"(h_off));
  };
  auto halo_c = [&](int rd, int buf) {
    // only the last round has lanes past the image's end; they must not write, not even zeros: the buffers are exact
    if (rd + 1 < HRN || (int)e < 0) lds_dma16_buf(hsrd, h_off, smem0 + buf * HBYTES + (rd * NT + wave * 64) * 16);
    e = ptab[(rd + 1 == HRN ? 0 : rd + 1) * NT];
  };
  auto halo_round = [&](int rd, int buf) { halo_a(); halo_c(rd, buf); };      // the prologue's rounds
  // deferred stores of the previous tile: piece sidx = row sidx / 2, pixels 16 * (sidx & 1) + lane / 4, 16 bytes lane % 4
  auto stage_piece = [&](int sidx) {
    return *(const u32x4*)(wt + (sidx >> 1) * 1024 + ((lane + 64 * (sidx & 1)) >> 2) * 32 + ((lane & 3) ^ (((lane >> 4) & 1) << 1)) * PIECE);
  };
  T* pdst = nullptr;
  auto store_ptr = [&](int sidx) { return pdst + (size_t)(sidx >> 1) * p.OW * BN + (sidx & 1) * 16 * BN; };
  auto store_piece = [&](int sidx, u32x4 v, u32x4 old) {
    if (ACC) {
      Piece<T> o, q;
      o.v = __builtin_bit_cast(bf16x8, v); q.v = __builtin_bit_cast(bf16x8, old);
#pragma unroll
      for (int e_ = 0; e_ < PIECE; ++e_) o.set(e_, o.get(e_) + q.get(e_));
      v = __builtin_bit_cast(u32x4, o.v);
    }
    *(u32x4*)store_ptr(sidx) = v;
  };
  constexpr int NST = 2 * TH;                  // stores per wave per tile
  constexpr int ST0 = 2 * HRN + 2;             // stores behind the halo rounds (ACC: old values requested at K step 0): they stay in flight over the tile's end
  static_assert(2 * HRN <= KSTEPS - 8 && ST0 + 2 * NST + 2 <= KSTEPS, "halo rounds and stores fit the K loop");

  int tile = tile_of(0);
  if (tile < p.mtiles) {
    halo_begin(tile);
#pragma unroll
    for (int rd = 0; rd < HRN; ++rd) halo_round(rd, 0);
  }
  wait_vmcnt<0>();
  bool has_prev = false;
  int buf = 0;
  for (int k = 0; tile < p.mtiles; ++k, buf ^= 1) {
    const int next = tile_of(k + 1);
    const bool has_next = next < p.mtiles;
    int img, ty0, tx0;
    tile_origin(tile, img, ty0, tx0);
    halo_begin(has_next ? next : tile);      // unconditional halo rounds: see igemm4
    __syncthreads();                   // halo(tile) has landed for every wave; nobody reads the other buffer any more
    f32x4 acc[TH][2][2];               // [pixel row][16-pixel half mh][16-channel tile nt]
    const unsigned char* hb = smem + buf * HBYTES;
    auto a_addr = [&](int t) {         // step t = (channel chunk, tap, pixel half mh = t & 1): all 32 k of the chunk
      const int cc = t / (NTAPS * 2), tap = (t / 2) % NTAPS, mh = t & 1;
      const int r = tap / S, s = tap % S;
      return ct[s] + cc * CHB + (flip ? R - 1 - r : r) * HSB + mh * 1024;
    };
    constexpr int PD = ACC ? 2 : VG_IGEMM4C_PD, AFRAGS = 64;      // prefetch distance of the A fragments in K steps
    u32x4 fa[PD + 1][TH];
    u32x4 sreg = u32x4{0, 0, 0, 0};
    u32x4 oreg[NST];
#pragma unroll
    for (int sidx = 0; sidx < NST; ++sidx) oreg[sidx] = u32x4{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < PD; ++t) {
      const int va = a_addr(t);
#pragma unroll
      for (int i = 0; i < TH; ++i) fa[t][i] = *(const u32x4*)(hb + va + i * HSB);
    }
    // One K step = 8 MFMAs (4 rows x 2 channel tiles, 16 cycles each; an MFMA holds the wave's issue for 8 of them).  The
    // step's other work is dealt out over the gaps behind each row's pair: A-fragment prefetch behind the first two, a
    // halo round or a staging read behind the third, a deferred store behind the fourth.
    auto mfma = [&](int t, int i) {
      const int cc = t / (NTAPS * 2), tap = (t / 2) % NTAPS, mh = t & 1;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const bool in_a = (tap * CCH + cc) * 2 + nt < AFRAGS;      // 64 fragments in accumulation registers, 8 in ordinary ones
        if (t < 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc[i][mh][nt]) : "v"(fa[t][i]), "a"(fb[tap][cc][nt]));
        else if (in_a) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][mh][nt]) : "v"(fa[t % (PD + 1)][i]), "a"(fb[tap][cc][nt]));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][mh][nt]) : "v"(fa[t % (PD + 1)][i]), "v"(fb[tap][cc][nt]));
      }
    };
    static_assert(PD >= 2, "steps 0 and 1 (the two pixel halves) read fa[0] and fa[1]");
#pragma clang loop unroll(full)
    for (int t = 0; t < KSTEPS; ++t) {
      const int vp = t + PD < KSTEPS ? a_addr(t + PD) : 0;
      __builtin_amdgcn_sched_barrier(0);
      mfma(t, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (t + PD < KSTEPS) {
        fa[(t + PD) % (PD + 1)][0] = *(const u32x4*)(hb + vp);
        fa[(t + PD) % (PD + 1)][1] = *(const u32x4*)(hb + vp + HSB);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma(t, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (t + PD < KSTEPS) {
        fa[(t + PD) % (PD + 1)][2] = *(const u32x4*)(hb + vp + 2 * HSB);
        fa[(t + PD) % (PD + 1)][3] = *(const u32x4*)(hb + vp + 3 * HSB);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma(t, 2);
      __builtin_amdgcn_sched_barrier(0);
      if (VG_HALO_SPLIT) { if (t % 2 == 0 && t / 2 < HRN) halo_a(); }
      else if (t % 2 == 0 && t / 2 < HRN) halo_round(t / 2, buf ^ 1);
      if (ACC && t == 1) {
        if (has_prev) {
#pragma unroll
          for (int sidx = 0; sidx < NST; ++sidx) oreg[sidx] = *(const u32x4*)store_ptr(sidx);
        }
      }
      if (t >= ST0 && (t - ST0) % 2 == 1 && (t - ST0) / 2 < NST) { if (has_prev) sreg = stage_piece((t - ST0) / 2); }
      __builtin_amdgcn_sched_barrier(0);
      mfma(t, 3);
      __builtin_amdgcn_sched_barrier(0);
      if (t >= ST0 + 2 && (t - ST0) % 2 == 0 && (t - ST0 - 2) / 2 < NST) { if (has_prev) store_piece((t - ST0 - 2) / 2, sreg, oreg[(t - ST0 - 2) / 2]); }
      if (VG_HALO_SPLIT && t % 2 == 1 && t / 2 < HRN) halo_c(t / 2, buf ^ 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15");     // MFMA result -> VALU / LDS read (no hazard recognizer for inline MFMAs)
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue: statistics of the bf16-rounded pre-activation, activation; the four rows into the staging tile
    {
      // accumulator (mh, nt), register r = pixel mh*16 + 4*kb + r of row i, channel 16*nt + m16.  The staging rows are 64
      // bytes: pixels with bit 2 set keep their two 16-channel halves swapped, so that the rows 4 apart held by a store's
      // 32-lane group (kb, kb + 1) fall on different banks (the deferred-store reads undo it).
      f32x2 a1[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}}, a2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
      for (int i = 0; i < TH; ++i) {
        T* wlb[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) wlb[nt] = wt + i * 1024 + 4 * kb * 32 + ((nt ^ (kb & 1)) * 16 + m16);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
              const int row = mh * 16 + r;
              f32x2 v = f32x2{acc[i][mh][nt][r], acc[i][mh][nt][r + 1]} + bv[nt];
              bf16x2_t q = __builtin_convertvector(v, bf16x2_t);
              if (ST && !LR) { a1[nt][0] = dot2_sum(q, a1[nt][0]); a2[nt][0] = dot2_sq(q, a2[nt][0]); }
              if (LR) v = __builtin_convertvector(q, f32x2);
              if (ST && LR) { a1[nt] += v; a2[nt] += v * v; }
              if (LR) {
                const f32x2 w = v * VG_LRELU_SLOPE;
                f32x2 m;
                asm("v_max_f32 %0, %1, %2" : "=v"(m[0]) : "v"(v[0]), "v"(w[0]));
                asm("v_max_f32 %0, %1, %2" : "=v"(m[1]) : "v"(v[1]), "v"(w[1]));
                q = __builtin_convertvector(m, bf16x2_t);
              }
              wlb[nt][row * 32] = q[0];
              wlb[nt][(row + 1) * 32] = q[1];
            }
      }
      if (ST) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) { st1[nt] += (double)(a1[nt][0] + a1[nt][1]); st2[nt] += (double)(a2[nt][0] + a2[nt][1]); }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    pdst = yg + ((((size_t)img * p.OH + ty0) * p.OW + tx0 + (lane >> 2)) * BN + 32 * wave + (lane & 3) * PIECE);
    has_prev = true;
    if (k > 0) wait_vmcnt<NST>(); else wait_vmcnt<0>();            // this tile's stores all sit behind the last halo round
    tile = next;
  }
  if (has_prev) {
#pragma unroll
    for (int sidx = 0; sidx < NST; ++sidx) {
      u32x4 old = u32x4{0, 0, 0, 0};
      if (ACC) old = *(const u32x4*)store_ptr(sidx);
      store_piece(sidx, stage_piece(sidx), old);
    }
  }
  if (ST) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      double s1 = st1[nt] + __shfl_xor(st1[nt], 32, 64), s2 = st2[nt] + __shfl_xor(st2[nt], 32, 64);
      s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);      // the four k-slot groups hold the same column
      if (kb == 0) {
        vg_stats_add(p.stats, p.Npad, 0, 32 * wave + 16 * nt + m16, blockIdx.x, s1, p.det);
        vg_stats_add(p.stats, p.Npad, 1, 32 * wave + 16 * nt + m16, blockIdx.x, s2, p.det);
      }
    }
  }
}

static bool g_igemm4 = true, g_igemm4p = true, g_igemm4c = true;
template <int R, int S, int CCH, bool LR, bool ST, bool ACC, bool PH>
static int launch_igemm4(const IgemmParams& p0, hipStream_t st, bool* handled) {
  typedef Igemm4Cfg<R, S, CCH, PH> Cfg;
  static_assert(Cfg::FITS, "igemm4 configuration does not fit");
  *handled = false;
  IgemmParams p = p0;
  int flip = -1;       // the tap grid must be R x S, row-major, offsets all ascending or all descending, in every phase
  for (int ph = 0; ph < p.nphases; ++ph) {
    int ymin = 127, xmin = 127;
    for (int t = 0; t < p.ntaps; ++t) {
      ymin = p.dy[ph][t] < ymin ? p.dy[ph][t] : ymin;
      xmin = p.dx[ph][t] < xmin ? p.dx[ph][t] : xmin;
    }
    p.dymin[ph] = (signed char)ymin; p.dxmin[ph] = (signed char)xmin;
    bool asc = true, desc = true;
    for (int t = 0; t < p.ntaps; ++t) {
      const int yo = p.dy[ph][t] - ymin, xo = p.dx[ph][t] - xmin;
      asc = asc && yo == t / S && xo == t % S;
      desc = desc && yo == R - 1 - t / S && xo == S - 1 - t % S;
    }
    const int f = asc ? 0 : (desc ? 1 : -2);
    if (f < 0 || (flip >= 0 && f != flip)) return VG_OK;
    flip = f;
  }
  if (PH) {            // the four phases together reach one pixel each way: a (TH + 2) x (TW + 2) halo
    int lo_y = 127, hi_y = -127, lo_x = 127, hi_x = -127;
    for (int ph = 0; ph < 4; ++ph) {
      lo_y = p.dymin[ph] < lo_y ? p.dymin[ph] : lo_y; hi_y = p.dymin[ph] > hi_y ? p.dymin[ph] : hi_y;
      lo_x = p.dxmin[ph] < lo_x ? p.dxmin[ph] : lo_x; hi_x = p.dxmin[ph] > hi_x ? p.dxmin[ph] : hi_x;
    }
    if (hi_y - lo_y > 1 || hi_x - lo_x > 1) return VG_OK;
  }
  p.flip = flip;
  if ((size_t)p.H * p.W * p.Cs * 2 >= ((size_t)1 << 31)) return VG_OK;        // piece offsets inside one image are 32-bit
  p.tilesX = p.OWl / Cfg::TW;
  p.tilesY = p.OHl / Cfg::TH;
  p.mtiles = p.N * p.tilesX * p.tilesY;
  p.ntiles = 1;
  p.dbg = g_dbg;
#ifdef VG_ABLATE
  p.splitws = (float*)g_stamp_buf;
#endif
  auto kern = igemm4_kernel<R, S, CCH, LR, ST, ACC, PH>;
    VG_HIP(vg_lds_attr((const void*)kern, Cfg::LDS));
  pack_taps(p);
  const int blocks = p.mtiles < 256 ? p.mtiles : 256;    // one persistent block per CU
  snprintf(g_last_kernel, sizeof g_last_kernel, PH ? "igemm4p<bf16,%dx%d,%d>" : "igemm4<bf16,%dx%d,%d>", R, S, CCH);
  hipLaunchKernelGGL(kern, dim3(blocks, 1), dim3(Cfg::NT), Cfg::LDS, st, p);
  VG_LAUNCH_CHECK();
  *handled = true;
  return VG_OK;
}
template <int CCH, bool LR, bool ST, bool ACC>
static int launch_igemm4c(const IgemmParams& p0, hipStream_t st, bool* handled) {
  typedef Igemm4cCfg<CCH> Cfg;
  static_assert(Cfg::FITS, "igemm4c configuration does not fit");
  *handled = false;
  IgemmParams p = p0;
  int ymin = 127, xmin = 127;
  for (int t = 0; t < 9; ++t) {
    ymin = p.dy[0][t] < ymin ? p.dy[0][t] : ymin;
    xmin = p.dx[0][t] < xmin ? p.dx[0][t] : xmin;
  }
  p.dymin[0] = (signed char)ymin; p.dxmin[0] = (signed char)xmin;
  bool asc = true, desc = true;
  for (int t = 0; t < 9; ++t) {
    const int yo = p.dy[0][t] - ymin, xo = p.dx[0][t] - xmin;
    asc = asc && yo == t / 3 && xo == t % 3;
    desc = desc && yo == 2 - t / 3 && xo == 2 - t % 3;
  }
  if (!asc && !desc) return VG_OK;
  p.flip = asc ? 0 : 1;
  if ((size_t)p.H * p.W * p.Cs * 2 >= ((size_t)1 << 31)) return VG_OK;
  if ((size_t)Cfg::HH * p.W * p.Cs * 2 >= ((size_t)1 << 21)) return VG_OK;    // 21-bit piece offsets in the packed halo table
  p.tilesX = p.OWl / Cfg::TW;
  p.tilesY = p.OHl / Cfg::TH;
  p.mtiles = p.N * p.tilesX * p.tilesY;
  p.ntiles = 1;
  auto kern = igemm4c_kernel<CCH, LR, ST, ACC>;
    VG_HIP(vg_lds_attr((const void*)kern, Cfg::LDS));
  pack_taps(p);
  const int blocks = p.mtiles < 256 ? p.mtiles : 256;
  snprintf(g_last_kernel, sizeof g_last_kernel, "igemm4c<bf16,3x3,%d>", CCH);
  hipLaunchKernelGGL(kern, dim3(blocks, 1), dim3(Cfg::NT), Cfg::LDS, st, p);
  VG_LAUNCH_CHECK();
  *handled = true;
  return VG_OK;
}
template <typename T> static int dispatch_igemm4(const IgemmParams& p, hipStream_t st, bool* handled) {
  *handled = false;
  if constexpr (sizeof(T) == 2) {
    // 128 -> 128 3x3: output channels split over the waves
    if (g_igemm4c && !g_conv_kernel_is_old() && p.ksplit <= 1 && p.in_s == 1 && p.nphases == 1 && p.out_s == 1 && p.ntaps == 9 &&
        p.C == 128 && p.Npad == 128 && p.Nout == 128 && p.Ks == 128 && p.OWl % 32 == 0 && p.OHl % 4 == 0 &&
        (p.act == VG_ACT_NONE || p.act == VG_ACT_LRELU)) {
      const bool lr = p.act == VG_ACT_LRELU, st_ = p.stats != nullptr;
      if (p.accumulate) {
        if (lr || st_) return VG_OK;
        return launch_igemm4c<4, false, false, true>(p, st, handled);
      }
      if (lr) return st_ ? launch_igemm4c<4, true, true, false>(p, st, handled) : launch_igemm4c<4, true, false, false>(p, st, handled);
      return st_ ? launch_igemm4c<4, false, true, false>(p, st, handled) : launch_igemm4c<4, false, false, false>(p, st, handled);
    }
    // whole tiles, 64 real output channels at stride 64: everything else stays on igemm3
    if (!g_igemm4 || g_conv_kernel_is_old() || p.ksplit > 1 || p.in_s != 1 || p.Npad != 64 || p.Nout != 64 || p.Ks != 64 ||
        p.OWl % 32 != 0 || (p.act != VG_ACT_NONE && p.act != VG_ACT_LRELU))
      return VG_OK;
    const bool lr = p.act == VG_ACT_LRELU, st_ = p.stats != nullptr;
    if (p.nphases == 1 && p.out_s == 1 && p.ntaps == 9 && p.C == 64 && p.OHl % 8 == 0) {
      if (p.accumulate) {      // input gradients onto a parked skip gradient: no activation, no statistics
        if (lr || st_) return VG_OK;
        return launch_igemm4<3, 3, 2, false, false, true, false>(p, st, handled);
      }
      if (lr) return st_ ? launch_igemm4<3, 3, 2, true, true, false, false>(p, st, handled) : launch_igemm4<3, 3, 2, true, false, false, false>(p, st, handled);
      return st_ ? launch_igemm4<3, 3, 2, false, true, false, false>(p, st, handled) : launch_igemm4<3, 3, 2, false, false, false, false>(p, st, handled);
    }
    // 128 -> 64 stride-2 transposed gathers: a phase per wave
    if (g_igemm4p && p.nphases == 4 && p.out_s == 2 && p.ntaps == 4 && p.C == 128 && p.OHl % 2 == 0 && !p.accumulate && !lr)
      return st_ ? launch_igemm4<2, 2, 4, false, true, false, true>(p, st, handled) : launch_igemm4<2, 2, 4, false, false, false, true>(p, st, handled);
  }
  return VG_OK;
}

template <typename T, int R, int S> static int dispatch_igemm3_rs(const IgemmParams& p, hipStream_t st, bool* handled) {
  // >= 256 output channels: 16x32 px x 128 ch tiles (two or more channel blocks per pixel tile) stream a third less through
  // the LDS-DMA per MFMA than 8x32 px x 256 ch (8 KB of weights + 4.5 KB of halo per K tile against 16 + 2.5): +3 % on the
  // 256- and 512-channel layers (tests/gpu_conv_perf2.py; VG_IGEMM3_WIDE_TILE=1 keeps the 256-channel tile)
  static const bool wide = getenv("VG_IGEMM3_WIDE_TILE") != nullptr && atoi(getenv("VG_IGEMM3_WIDE_TILE")) != 0;
  if constexpr (Igemm3Cfg<T, 4, 2, 4, 2, R, S>::FITS) {
    if (!wide && p.Npad % 256 == 0 && p.OHl % 16 == 0) return launch_igemm3<T, 4, 2, 4, 2, R, S>(p, st, handled);
  }
  if (p.Npad % 256 == 0) return launch_igemm3<T, 4, 2, 2, 4, R, S>(p, st, handled);   // 8x32 px x 256 ch
  if (p.Npad == 128) {
    if constexpr (Igemm3Cfg<T, 4, 2, 4, 2, R, S>::FITS) return launch_igemm3<T, 4, 2, 4, 2, R, S>(p, st, handled);   // 16x32 px x 128 ch
    else return launch_igemm3<T, 2, 2, 4, 2, R, S>(p, st, handled);               // 8x32 px x 128 ch (three halo buffers fit)
  }
  if (p.Npad == 64) {
    // 3x3: 8x32 px x 64 ch (75 KB of LDS, 94 VGPRs: two blocks per CU overlap each other's prologue and epilogue)
    if constexpr (R == 3) return launch_igemm3<T, 1, 2, 8, 1, R, S>(p, st, handled);
    else return launch_igemm3<T, 2, 2, 8, 1, R, S>(p, st, handled);                   // 16x32 px x 64 ch
  }
  if constexpr (R == 3) {
    if (p.Npad == 32) return launch_igemm3<T, 1, 1, 8, 1, R, S>(p, st, handled);      // thin outputs (64->3, d/dx of 3->64)
  }
  return VG_OK;
}
template <typename T> static int dispatch_igemm3(const IgemmParams& p, hipStream_t st, bool* handled) {
  *handled = false;
  if constexpr (sizeof(T) == 2) {
    if (g_conv_kernel < 3 || p.ksplit > 1 || p.OWl < 24) return VG_OK;
    if (p.in_s == 2) return dispatch_igemm3s2<T>(p, st, handled);
    if (p.in_s != 1) return VG_OK;
    if (p.ntaps == 9) return dispatch_igemm3_rs<T, 3, 3>(p, st, handled);
    if (p.ntaps == 16) return dispatch_igemm3_rs<T, 4, 4>(p, st, handled);
    if (p.ntaps == 4) return dispatch_igemm3_rs<T, 2, 2>(p, st, handled);
  }
  return VG_OK;
}

static bool g_use_igemm2 = true;
extern "C" void vg_set_conv_kernel(int version) {
  g_conv_kernel = version & 0x0f;                 // bit 4 (16): stride-2 forward stays on the im2col-row kernel
  g_igemm3_s2 = !(version & 0x10);
  g_igemm4 = !(version & 0x20);                   // bit 5 (32): no weights-stationary kernel (the 64-channel layers stay on igemm3)
  g_igemm4c = !(version & 0x80);                  // bit 7 (128): no channel-split variant (128 -> 128 3x3 stays on igemm3)
  g_igemm4p = !(version & 0x40);                  // bit 6 (64): no phase-per-wave variant of it (128 -> 64 transposed gathers)
  g_use_igemm2 = g_conv_kernel != 1;
  g_dbg = version >> 8;
}

template <typename T> static int dispatch_igemm2(const IgemmParams& p, hipStream_t st, bool* handled) {
  *handled = false;
  if (!g_use_igemm2 || p.ksplit > 1 || p.in_s != 1) return VG_OK;   // stride-2 gathers: the im2col-row kernel is faster
  if (p.OHl * p.OWl == 1) return VG_OK;   // Linear layers ([B,1,1,F] "images"): one spatial tile per image would be 1 pixel wide
  if (p.in_s == 1) {
    if (p.Npad % 128 == 0) return launch_igemm2<T, 2, 2, 4, 2, 16>(p, st, handled);
    if (p.Npad == 64) return launch_igemm2<T, 2, 2, 4, 1, 16>(p, st, handled);
    if (p.Npad == 32) return launch_igemm2<T, 2, 1, 4, 1, 16>(p, st, handled);
  } else {
    if (p.Npad % 128 == 0) return launch_igemm2<T, 2, 2, 2, 2, 8>(p, st, handled);
    if (p.Npad == 64) return launch_igemm2<T, 1, 2, 4, 1, 8>(p, st, handled);
    if (p.Npad == 32) return launch_igemm2<T, 1, 1, 4, 1, 8>(p, st, handled);
  }
  return VG_OK;
}

template <typename T> static int dispatch_igemm(const IgemmParams& p, hipStream_t st) {
  bool handled = false;
  int rc4 = dispatch_igemm4<T>(p, st, &handled);
  if (rc4 != VG_OK || handled) return rc4;
  int rc3 = dispatch_igemm3<T>(p, st, &handled);
  if (rc3 != VG_OK || handled) return rc3;
  int rc2 = dispatch_igemm2<T>(p, st, &handled);
  if (rc2 != VG_OK || handled) return rc2;
  if (p.Npad % 128 == 0) return launch_igemm<T, 2, 2, 2, 2>(p, st);   // 128 x 128
  if (p.Npad == 64) return launch_igemm<T, 2, 2, 4, 1>(p, st);        // 256 x 64
  if (p.Npad == 32) return launch_igemm<T, 2, 1, 4, 1>(p, st);        // 256 x 32
  vg_set_error("igemm: unsupported padded N %d", p.Npad);
  return VG_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// geometry helpers shared with pack.hip
// ---------------------------------------------------------------------------------------------
int vg_npad(int n) { return n <= 32 ? 32 : (n <= 64 ? 64 : (int)vg_ceil_div(n, 128) * 128); }

static int kc_of(int dtype) { return dtype == VG_BF16 ? 32 : 16; }

// Tap table of the transposed (dgrad / convT-forward) formulation: output row Y = y*s + py takes
// input rows iy = y + dy with kernel row r, (py + pad - r) % s == 0, dy = (py + pad - r)/s.
int vg_transposed_taps(const vg_conv_desc* d, int phase, signed char* dy, signed char* dx, int* rs) {
  const int s = d->stride, py = phase / s, px = phase % s;
  int n = 0;
  for (int r = 0; r < d->r; ++r) {
    if ((py + d->pad - r) % s != 0) continue;
    for (int c = 0; c < d->s; ++c) {
      if ((px + d->pad - c) % s != 0) continue;
      if (n >= VG_MAX_TAPS) return -1;
      dy[n] = (signed char)((py + d->pad - r) / s);
      dx[n] = (signed char)((px + d->pad - c) / s);
      if (rs) rs[n] = r * d->s + c;
      ++n;
    }
  }
  return n;
}

static int check_desc(const vg_conv_desc* d) {
  VG_REQUIRE(d != nullptr, "conv desc is null");
  VG_REQUIRE(d->dtype == VG_F32 || d->dtype == VG_BF16, "conv: bad dtype %d", d->dtype);
  const int kc = kc_of(d->dtype);
  VG_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->ci > 0 && d->co > 0, "conv: non-positive dims");
  VG_REQUIRE(d->r > 0 && d->s > 0 && d->r * d->s <= VG_MAX_TAPS, "conv: kernel %dx%d unsupported", d->r, d->s);
  VG_REQUIRE(d->stride >= 1 && d->stride <= 2, "conv: stride %d unsupported", d->stride);
  VG_REQUIRE(d->r % d->stride == 0 || d->stride == 1, "conv: kernel %d not a multiple of stride %d", d->r, d->stride);
  VG_REQUIRE(d->oh == (d->h + 2 * d->pad - d->r) / d->stride + 1 && d->ow == (d->w + 2 * d->pad - d->s) / d->stride + 1,
             "conv: output size %dx%d inconsistent with input %dx%d k%d s%d p%d", d->oh, d->ow, d->h, d->w, d->r,
             d->stride, d->pad);
  VG_REQUIRE(d->xcs % kc == 0 && d->xcs >= d->ci, "conv: x channel stride %d must be a multiple of %d and >= %d", d->xcs, kc, d->ci);
  VG_REQUIRE(d->ycs % kc == 0 && d->ycs >= d->co, "conv: y channel stride %d must be a multiple of %d and >= %d", d->ycs, kc, d->co);
  VG_REQUIRE((int64_t)d->n * d->h * d->w < (1ll << 31) && (int64_t)d->n * d->oh * d->ow < (1ll << 31), "conv: too many pixels");
  return VG_OK;
}

static int cpad_of(int c, int dtype) { int kc = kc_of(dtype); return (int)vg_ceil_div(c, kc) * kc; }

// BatchNorm-backward sums in the epilogue: only the ping-pong spatial-tile kernels implement them.  Returns whether
// the launch that just ran took them (g_last_kernel names it).
static int check_bs(const vg_bn_bwd_stats* bs, int nout) {
  VG_REQUIRE(bs->x && bs->mean && bs->invstd && bs->sums, "conv bnstats: x, mean, invstd and sums are required");
  VG_REQUIRE(bs->y || (bs->scale && bs->shift) || bs->act != VG_ACT_LRELU,
             "conv bnstats: the LeakyReLU mask needs y, or scale and shift to recompute it from x");
  VG_REQUIRE(bs->act == VG_ACT_NONE || bs->act == VG_ACT_LRELU, "conv bnstats: activation %d unsupported", bs->act);
  VG_REQUIRE(bs->cpad >= nout, "conv bnstats: cpad %d < %d channels", bs->cpad, nout);
  return VG_OK;
}
static bool last_kernel_takes_bs() { return strncmp(g_last_kernel, "igemm3", 6) == 0; }

static int conv_fwd_impl(const vg_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y,
                         double* stats, int act, int accumulate, const vg_bn_bwd_stats* bs, int* fused, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  VG_REQUIRE(x && w_packed && y, "conv_fwd: null pointer");
  // every gather-GEMM tile is at least 128 pixels: an upper bound of the blocks that add to one channel's statistics
  if (stats || bs) VG_DET_GUARD(vg_ceil_div((long)d->n * d->oh * d->ow, 128));
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.det = vg_det_mode();
  if (fused) *fused = 0;
  if (bs) {
    rc = check_bs(bs, d->co);
    if (rc) return rc;
    if (bias == nullptr && stats == nullptr && act == VG_ACT_NONE && d->dtype == VG_BF16) p.bs = *bs;
  }
  p.x = x; p.w = w_packed; p.bias = bias; p.y = y; p.stats = stats;
  p.N = d->n; p.H = d->h; p.W = d->w; p.C = cpad_of(d->ci, d->dtype); p.Cs = d->xcs;
  p.OHl = d->oh; p.OWl = d->ow; p.OH = d->oh; p.OW = d->ow; p.Ks = d->ycs;
  p.Nout = d->co; p.Npad = vg_npad(d->co);
  p.in_s = d->stride; p.out_s = 1; p.ntaps = d->r * d->s; p.nphases = 1;
  p.act = act; p.accumulate = accumulate; p.ksplit = 1;
  p.M = d->n * d->oh * d->ow;
  for (int r = 0; r < d->r; ++r)
    for (int s = 0; s < d->s; ++s) { p.dy[0][r * d->s + s] = (signed char)(r - d->pad); p.dx[0][r * d->s + s] = (signed char)(s - d->pad); }
  rc = d->dtype == VG_BF16 ? dispatch_igemm<bf16_t>(p, (hipStream_t)stream) : dispatch_igemm<float>(p, (hipStream_t)stream);
  if (rc == VG_OK && fused && p.bs.x != nullptr && last_kernel_takes_bs()) *fused = 1;
  return rc;
}

extern "C" int vg_conv_fwd(const vg_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y,
                           double* stats, int act, int accumulate, void* stream) {
  return conv_fwd_impl(d, x, w_packed, bias, y, stats, act, accumulate, nullptr, nullptr, stream);
}

extern "C" int vg_conv_fwd_bnstats(const vg_conv_desc* d, const void* x, const void* w_packed, void* y, int accumulate,
                                   const vg_bn_bwd_stats* bs, int* fused, void* stream) {
  VG_REQUIRE(bs && fused, "conv_fwd_bnstats: null pointer");
  return conv_fwd_impl(d, x, w_packed, nullptr, y, nullptr, VG_ACT_NONE, accumulate, bs, fused, stream);
}

// ---- split-K forward: few output tiles but a long reduction (Linear layers viewed as 1x1 convs over
// [B,1,1,F] tensors, the PatchGAN tail).  Partials are summed in an f32 workspace, then finished.
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                                            T* __restrict__ y, long pixels, int Npad, int Ks, int Nout, int act) {
  const long total = pixels * Ks;
  for (long t = blockIdx.x * 256L + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const long pix = t / Ks;
    const int n = (int)(t - pix * Ks);
    float v = 0.f;
    if (n < Nout) {
      v = ws[pix * Npad + n] + (bias ? bias[n] : 0.f);
      v = Elt<T>::to_f(Elt<T>::from_f(v));
      if (act == VG_ACT_LRELU) v = vg_lrelu(v);
      else if (act == VG_ACT_SIGMOID) v = vg_sigmoid(v);
    }
    y[t] = Elt<T>::from_f(v);
  }
}

extern "C" size_t vg_conv_splitk_ws_bytes(const vg_conv_desc* d) {
  if (!d) return 0;
  return (size_t)d->n * d->oh * d->ow * vg_npad(d->co) * sizeof(float);
}

extern "C" int vg_conv_fwd_splitk(const vg_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y,
                                  int act, int ksplit, void* ws, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  VG_REQUIRE(x && w_packed && y && ws, "conv_fwd_splitk: null pointer");
  hipStream_t st = (hipStream_t)stream;
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.det = vg_det_mode();
  p.x = x; p.w = w_packed; p.bias = nullptr; p.y = y; p.stats = nullptr;
  p.N = d->n; p.H = d->h; p.W = d->w; p.C = cpad_of(d->ci, d->dtype); p.Cs = d->xcs;
  p.OHl = d->oh; p.OWl = d->ow; p.OH = d->oh; p.OW = d->ow; p.Ks = d->ycs;
  p.Nout = d->co; p.Npad = vg_npad(d->co);
  p.in_s = d->stride; p.out_s = 1; p.ntaps = d->r * d->s; p.nphases = 1;
  p.M = d->n * d->oh * d->ow;
  for (int r = 0; r < d->r; ++r)
    for (int s = 0; s < d->s; ++s) { p.dy[0][r * d->s + s] = (signed char)(r - d->pad); p.dx[0][r * d->s + s] = (signed char)(s - d->pad); }
  const int nk = p.ntaps * (p.C / kc_of(d->dtype));
  if (ksplit <= 0) {   // heuristic: enough blocks for ~2 per CU, at least 8 K-chunks per slice
    const long tiles = vg_ceil_div(p.M, p.Npad % 128 == 0 ? 128 : 256) * (p.Npad % 128 == 0 ? p.Npad / 128 : 1);
    ksplit = (int)vg_ceil_div(512, tiles);
  }
  if (ksplit > nk / 8) ksplit = nk / 8;
  if (vg_det_mode()) ksplit = 1;      // the slices meet in f32 atomics: reproducible runs do not split
  if (ksplit < 1) ksplit = 1;
  if (ksplit > 65535) ksplit = 65535;
  p.ksplit = ksplit; p.splitws = (float*)ws; p.act = VG_ACT_NONE;
  if (ksplit == 1) {
    p.bias = bias; p.act = act;
    return d->dtype == VG_BF16 ? dispatch_igemm<bf16_t>(p, st) : dispatch_igemm<float>(p, st);
  }
  VG_HIP(hipMemsetAsync(ws, 0, vg_conv_splitk_ws_bytes(d), st));
  rc = d->dtype == VG_BF16 ? dispatch_igemm<bf16_t>(p, st) : dispatch_igemm<float>(p, st);
  if (rc) return rc;
  const long total = (long)p.M * p.Ks;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (d->dtype == VG_BF16) hipLaunchKernelGGL(splitk_finish_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const float*)ws, bias, (bf16_t*)y, (long)p.M, p.Npad, p.Ks, p.Nout, act);
  else hipLaunchKernelGGL(splitk_finish_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)ws, bias, (float*)y, (long)p.M, p.Npad, p.Ks, p.Nout, act);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

static int conv_dgrad_impl(const vg_conv_desc* d, const void* dy, const void* w_packed_t, const float* bias, void* dx,
                           double* stats, int act, int accumulate, const vg_bn_bwd_stats* bs, int* fused, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  VG_REQUIRE(dy && w_packed_t && dx, "conv_dgrad: null pointer");
  const int s = d->stride;
  VG_REQUIRE(d->h % s == 0 && d->w % s == 0, "conv_dgrad: input %dx%d not a multiple of stride %d", d->h, d->w, s);
  if (stats || bs) VG_DET_GUARD(vg_ceil_div((long)d->n * d->h * d->w, 128));
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.det = vg_det_mode();
  if (fused) *fused = 0;
  if (bs) {
    rc = check_bs(bs, d->ci);
    if (rc) return rc;
    if (bias == nullptr && stats == nullptr && act == VG_ACT_NONE && d->dtype == VG_BF16) p.bs = *bs;
  }
  p.x = dy; p.w = w_packed_t; p.bias = bias; p.y = dx; p.stats = stats;
  p.N = d->n; p.H = d->oh; p.W = d->ow; p.C = cpad_of(d->co, d->dtype); p.Cs = d->ycs;
  p.OHl = d->h / s; p.OWl = d->w / s; p.OH = d->h; p.OW = d->w; p.Ks = d->xcs;
  p.Nout = d->ci; p.Npad = vg_npad(d->ci);
  p.in_s = 1; p.out_s = s; p.nphases = s * s;
  p.act = act; p.accumulate = accumulate; p.ksplit = 1;
  p.M = d->n * p.OHl * p.OWl;
  int nt = -1;
  for (int ph = 0; ph < s * s; ++ph) {
    int n = vg_transposed_taps(d, ph, p.dy[ph], p.dx[ph], nullptr);
    VG_REQUIRE(n > 0 && (nt < 0 || n == nt), "conv_dgrad: uneven tap count across phases (k%d s%d p%d)", d->r, s, d->pad);
    nt = n;
  }
  p.ntaps = nt;
  rc = d->dtype == VG_BF16 ? dispatch_igemm<bf16_t>(p, (hipStream_t)stream) : dispatch_igemm<float>(p, (hipStream_t)stream);
  if (rc == VG_OK && fused && p.bs.x != nullptr && last_kernel_takes_bs()) *fused = 1;
  return rc;
}

extern "C" int vg_conv_dgrad(const vg_conv_desc* d, const void* dy, const void* w_packed_t, const float* bias, void* dx,
                             double* stats, int act, int accumulate, void* stream) {
  return conv_dgrad_impl(d, dy, w_packed_t, bias, dx, stats, act, accumulate, nullptr, nullptr, stream);
}

extern "C" int vg_conv_dgrad_bnstats(const vg_conv_desc* d, const void* dy, const void* w_packed_t, void* dx, int accumulate,
                                     const vg_bn_bwd_stats* bs, int* fused, void* stream) {
  VG_REQUIRE(bs && fused, "conv_dgrad_bnstats: null pointer");
  return conv_dgrad_impl(d, dy, w_packed_t, nullptr, dx, nullptr, VG_ACT_NONE, accumulate, bs, fused, stream);
}

// =============================================================================================
// weight gradient: dW[o][tap][c] = sum_m dy[m][o] * x[pix(m,tap)][c]   (K dimension = pixels)
// =============================================================================================
struct WgradParams {
  const void* x;
  const void* dy;
  float* dw;            // packed f32 [Opad][ntaps][Cpad], zeroed by the launcher
  int N, H, W, Cpad, Cs;
  int OH, OW, Opad, Os;
  int in_s, ntaps;
  int M, chunks_per_split, nchunks;
  int otiles, ctiles;
  signed char dy_[VG_MAX_TAPS];
  signed char dx_[VG_MAX_TAPS];
  int dbg;              // timing ablations only (VG_ABLATE build): 16 = one B fragment for all taps, 32 = no in-loop DMA, 64 = no barrier
  int det;              // reproducible accumulation (vg_set_deterministic): one pixel split, waves add in order
  size_t part_elems;    // spatial-tile kernels: dw is [groups * KS][part_elems] partial images, written with plain stores
  int xcd;              // spatial-tile kernels: deal tile ranges to the XCDs (see wg3_walk)
};

// 64-byte-segment swizzle for [pixel][channel] tiles read through ds_read_b64_tr_b16 (bf16 only)
template <typename T, int RS> __device__ __forceinline__ int wg_off(int row, int pc) {
  if (sizeof(T) == 4) return row * RS + pc * 16;
  constexpr int NSEG = RS / 64;
  int seg = pc >> 2;
  if (NSEG >= 4) seg ^= (row & 3);
  else if (NSEG == 2) seg ^= ((row >> 1) & 1);
  return row * RS + (seg << 6) + ((pc & 3) << 4);
}

template <typename T, int OI, int CI, int WO, int WC, int WK>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
  constexpr int BO = 32 * OI * WO, BC = 32 * CI * WC;
  constexpr int EB = sizeof(T), PIECE = 16 / EB;
  constexpr int KPW = 64 / EB;          // pixels per wave per chunk (32 bf16 / 16 f32)
  constexpr int KP = KPW * WK;          // pixels per chunk
  constexpr int RSO = BO * EB, RSC = BC * EB;   // tile row strides in bytes
  constexpr int TO = KP * RSO, TC = KP * RSC;
  constexpr int PPRO = BO / PIECE, PPRC = BC / PIECE;
  constexpr int NPO = (KP * PPRO + 255) / 256, NPC = (KP * PPRC + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sO = smem;             // [2][KP][BO]
  unsigned char* sC = smem + 2 * TO;    // [2][KP][BC]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave % WK, wc = (wave / WK) % WC, wo = wave / (WK * WC);
  const int h = lane >> 5, l32 = lane & 31;

  int bid = blockIdx.x;
  const int tap = bid % p.ntaps; bid /= p.ntaps;
  const int ctile = bid % p.ctiles; const int otile = bid / p.ctiles;
  const int o0 = otile * BO, c0 = ctile * BC;
  const int chunk0 = blockIdx.y * p.chunks_per_split;
  int chunk1 = chunk0 + p.chunks_per_split;
  if (chunk1 > p.nchunks) chunk1 = p.nchunks;
  const int tdy = p.dy_[tap], tdx = p.dx_[tap];

  const T* __restrict__ xg = (const T*)p.x;
  const T* __restrict__ dyg = (const T*)p.dy;

  // staging assignment: piece q = tid + 256*i  ->  (row = q / PPR, pc = q % PPR)
  // x rows need (n, oy, ox) of pixel m = chunk*KP + row, advanced incrementally by KP per chunk.
  int xr_n[NPC], xr_oy[NPC], xr_ox[NPC];
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    int q = tid + 256 * i;
    int row = q / PPRC;
    long m = (long)chunk0 * KP + row;
    int ohw = p.OH * p.OW;
    int n = (int)(m / ohw);
    int rem = (int)(m - (long)n * ohw);
    xr_n[i] = n; xr_oy[i] = rem / p.OW; xr_ox[i] = rem - xr_oy[i] * p.OW;
  }
  u32x4 ro[NPO], rc[NPC];
  int lchunk = chunk0;   // next chunk to load
  auto load_chunk = [&]() {
    const long mbase = (long)lchunk * KP;
#pragma unroll
    for (int i = 0; i < NPO; ++i) {
      int q = tid + 256 * i;
      int row = q / PPRO, pc = q - row * PPRO;
      u32x4 v = {0u, 0u, 0u, 0u};
      long m = mbase + row;
      if (q < KP * PPRO && m < p.M && o0 + pc * PIECE < p.Os) v = *(const u32x4*)(dyg + (size_t)m * p.Os + o0 + pc * PIECE);
      ro[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      int q = tid + 256 * i;
      int row = q / PPRC, pc = q - row * PPRC;
      u32x4 v = {0u, 0u, 0u, 0u};
      int iy = xr_oy[i] * p.in_s + tdy, ix = xr_ox[i] * p.in_s + tdx;
      bool ok = q < KP * PPRC && xr_n[i] < p.N && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W &&
                c0 + pc * PIECE < p.Cpad;
      if (ok) v = *(const u32x4*)(xg + ((size_t)(xr_n[i] * p.H + iy) * p.W + ix) * p.Cs + c0 + pc * PIECE);
      rc[i] = v;
      // advance this row's pixel by KP for the next chunk
      int ox = xr_ox[i] + KP, oy = xr_oy[i], n = xr_n[i];
      while (ox >= p.OW) { ox -= p.OW; ++oy; }
      while (oy >= p.OH) { oy -= p.OH; ++n; }
      xr_ox[i] = ox; xr_oy[i] = oy; xr_n[i] = n;
    }
    ++lchunk;
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NPO; ++i) {
      int q = tid + 256 * i;
      int row = q / PPRO, pc = q - row * PPRO;
      if (q < KP * PPRO) *(u32x4*)(sO + buf * TO + wg_off<T, RSO>(row, pc)) = ro[i];
    }
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      int q = tid + 256 * i;
      int row = q / PPRC, pc = q - row * PPRC;
      if (q < KP * PPRC) *(u32x4*)(sC + buf * TC + wg_off<T, RSC>(row, pc)) = rc[i];
    }
  };

  f32x16 acc[OI][CI];
#pragma unroll
  for (int i = 0; i < OI; ++i)
#pragma unroll
    for (int j = 0; j < CI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (chunk0 < chunk1) {
    load_chunk();
    store_chunk(0);
  }
  __syncthreads();
  for (int ch = chunk0; ch < chunk1; ++ch) {
    const int buf = (ch - chunk0) & 1;
    if (ch + 1 < chunk1) load_chunk();
    const unsigned char* cO = sO + buf * TO;
    const unsigned char* cC = sC + buf * TC;
    if constexpr (sizeof(T) == 2) {
      // transposed LDS reads: lane (q = li>>2, pp = li&3) of each 16-lane group supplies the address of
      // row q, columns 4pp..4pp+3 of a 4x16 block; lane li receives column li of the 4 rows.
      const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        u32x4 fa[OI], fb[CI];
#pragma unroll
        for (int i = 0; i < OI; ++i) {
          bf16x4 lo, hi;
          const int col = (wo * OI + i) * 32 + 16 * g16 + 4 * tp;   // channel (element) index in tile
          const int r0 = wk * KPW + kk * 16 + 8 * h + tq;
          lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(cO + wg_off<T, RSO>(r0, col >> 3) + (col & 7) * 2));
          hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(cO + wg_off<T, RSO>(r0 + 4, col >> 3) + (col & 7) * 2));
          bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          fa[i] = __builtin_bit_cast(u32x4, f);
        }
#pragma unroll
        for (int i = 0; i < CI; ++i) {
          bf16x4 lo, hi;
          const int col = (wc * CI + i) * 32 + 16 * g16 + 4 * tp;
          const int r0 = wk * KPW + kk * 16 + 8 * h + tq;
          lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(cC + wg_off<T, RSC>(r0, col >> 3) + (col & 7) * 2));
          hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(cC + wg_off<T, RSC>(r0 + 4, col >> 3) + (col & 7) * 2));
          bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          fb[i] = __builtin_bit_cast(u32x4, f);
        }
#pragma unroll
        for (int i = 0; i < OI; ++i)
#pragma unroll
          for (int j = 0; j < CI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
      }
    } else {
      // f32: A[i = channel l32][k = pixel 2*step + h] is a plain 4-byte LDS read
#pragma unroll
      for (int st = 0; st < KPW / 2; ++st) {
        const int row = wk * KPW + 2 * st + h;
        float fa[OI], fb[CI];
#pragma unroll
        for (int i = 0; i < OI; ++i) fa[i] = *(const float*)(cO + row * RSO + ((wo * OI + i) * 32 + l32) * 4);
#pragma unroll
        for (int i = 0; i < CI; ++i) fb[i] = *(const float*)(cC + row * RSC + ((wc * CI + i) * 32 + l32) * 4);
#pragma unroll
        for (int i = 0; i < OI; ++i)
#pragma unroll
          for (int j = 0; j < CI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
    if (ch + 1 < chunk1) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // acc[i][j][r]: row (o) = (r&3) + 8*(r>>2) + 4*h, column (c) = l32
  const size_t rowstride = (size_t)p.ntaps * p.Cpad;
  // p.det (one pixel split per dW tile, so only the WK waves of this block meet on an element): the waves add in wave
  // order (their atomics are performed at the L2 before the barrier lets the next wave go): the same bits on every run
  for (int turn = 0; turn < (p.det ? WK : 1); ++turn) {
    if (!p.det || wk == turn) {
#pragma unroll
      for (int i = 0; i < OI; ++i)
#pragma unroll
        for (int j = 0; j < CI; ++j) {
          const int c = c0 + (wc * CI + j) * 32 + l32;
          if (c >= p.Cpad) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = o0 + (wo * OI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (o >= p.Opad) continue;
            unsafeAtomicAdd(&p.dw[(size_t)o * rowstride + (size_t)tap * p.Cpad + c], acc[i][j][r]);
          }
        }
    }
    if (p.det && WK > 1) { __threadfence(); __syncthreads(); }
  }
}

template <typename T, int OI, int CI, int WO, int WC, int WK>
static int launch_wgrad(const WgradParams& p0, hipStream_t st) {
  WgradParams p = p0;
  constexpr int BO = 32 * OI * WO, BC = 32 * CI * WC, EB = sizeof(T);
  constexpr int KP = (64 / EB) * WK;
  const size_t lds = 2 * (size_t)KP * (BO + BC) * EB;
  p.otiles = (int)vg_ceil_div(p.Opad, BO);
  p.ctiles = (int)vg_ceil_div(p.Cpad, BC);
  p.nchunks = (int)vg_ceil_div(p.M, KP);
  const int base_blocks = p.otiles * p.ctiles * p.ntaps;
  int splits = (int)vg_ceil_div(1024, base_blocks);        // aim for ~4 blocks per CU
  const int min_chunks = 8;
  if (splits > p.nchunks / min_chunks) splits = p.nchunks / min_chunks;
  p.det = vg_det_mode();
  if (p.det) splits = 1;              // pixel splits meet in f32 atomics: reproducible runs keep one block per dW tile
  if (splits < 1) splits = 1;
  p.chunks_per_split = (int)vg_ceil_div(p.nchunks, splits);
  splits = (int)vg_ceil_div(p.nchunks, p.chunks_per_split);
  auto kern = wgrad_kernel<T, OI, CI, WO, WC, WK>;
    VG_HIP(vg_lds_attr((const void*)kern, (int)lds));
  hipLaunchKernelGGL(kern, dim3(base_blocks, splits), dim3(256), lds, st, p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// =============================================================================================
// wgrad3: spatial-tile weight gradient for 3x3 stride-1 convolutions (bf16).
//
// The kernel above gives every tap its own blocks, so x and dy travel L2 -> LDS once per tap (18 tensor
// passes for a 3x3) and it is bound by that traffic.  Here a block of 8 waves owns one (BO x BC) slab of
// dW for ALL nine taps and walks over TH x 32 pixel tiles of the output (TH = 4; 8 for the 64 x 64 slab, whose stages
// then still fit twice): per tile the dy tile and the x halo ((TH + 2) x 34 pixels) are brought in ONCE by LDS-DMA,
// two or three tiles deep (what 160 KB hold), and the taps read their shifted [pixel][channel] operands with transposed
// LDS reads (ds_read_b64_tr_b16 on plain 64-byte pixel rows: a tap shift is an immediate offset).  A wave keeps one
// 32 x 32 block of every tap in registers (144 accumulators, four 16 x 16 MFMA blocks per tap: v_mfma_f32_16x16x32_bf16,
// a 32-pixel tile row is its K) over its share of the tile rows; every tile group writes its slab to a partial image
// of its own, summed in a fixed order by wgrad_unpack_kernel.  One barrier per tile, counted vmcnt.
// =============================================================================================
// tile groups per slab pair: one block per CU (LDS-limited), at least; never more groups than tiles
static int wg3_groups(int pairs, int n, int oh, int ow, int th, int tw) {
  const int ntiles = n * (int)vg_ceil_div(oh, th) * (int)vg_ceil_div(ow, tw);
  int G = (int)vg_ceil_div(256, pairs);
  return G > ntiles ? ntiles : G;
}

// Which tiles a block of the spatial-tile kernels walks.  Plain dealing (xcd == 0): block b is slab pair b % pairs of
// tile group b / pairs, and group g takes tiles g, g + G, ...  With pairs a multiple of 8 that puts every block of a
// slab pair on ONE XCD (blocks go to XCD b % 8), so each L2 streams all of dy (or x) for its pairs and HBM sees the
// tensors once per XCD.  XCD dealing (G a multiple of 8): XCD x = b % 8 owns the contiguous tile range
// [x * ntiles / 8, (x + 1) * ntiles / 8) and holds G / 8 tile groups with ALL their slab pairs, so the pairs' re-reads
// of a tile and the halo rows shared by neighbouring tiles hit that XCD's L2 and HBM sees x and dy once.  A single
// slab pair (64 -> 64 channels) has no re-reads to share and keeps the plain dealing: eight far-apart streams at
// power-of-two distances measured 4-6 % slower there.
static int wg3_xcd_dealing() {                         // VG_WG3_XCD=0: plain dealing (A/B runs)
  static const int v = [] { const char* e = getenv("VG_WG3_XCD"); return e ? atoi(e) : 1; }();
  return v;
}
struct Wg3Walk { int pair, g, t0, step, end; };
__device__ inline Wg3Walk wg3_walk(int b, int pairs, int G, int ntiles, int xcd) {
  Wg3Walk w;
  if (xcd && (G & 7) == 0 && pairs > 1) {
    const int x = b & 7, k = b >> 3, J = G >> 3;
    const int j = k / pairs;
    w.pair = k - j * pairs;
    w.g = x * J + j;
    w.t0 = (int)(((long)x * ntiles) >> 3) + j;
    w.step = J;
    w.end = (int)(((long)(x + 1) * ntiles) >> 3);
  } else {
    w.pair = b % pairs; w.g = b / pairs; w.t0 = w.g; w.step = G; w.end = ntiles;
  }
  return w;
}

#ifndef VG_WG3_TH222
#define VG_WG3_TH222 8
#endif
static constexpr int wg3_th(int nob, int ncb, int ks) { return (nob == 2 && ncb == 2 && ks == 2) ? VG_WG3_TH222 : 4; }
template <int NOB, int NCB, int KS>
struct Wg3Cfg {
  static constexpr int TH = wg3_th(NOB, NCB, KS), TW = 32, R = 3, S = 3, NTAPS = 9;
  static constexpr int HH = TH + R - 1, HS = TW + S - 1;              // 6 x 34 halo pixels
  static constexpr int DYP = TH * TW / 16;                            // 16-pixel (1 KiB) DMA pieces per dy chunk image
  static constexpr int XP = (HH * HS + 15) / 16;                      // ... per x halo chunk image
  static constexpr int DYB = DYP * 1024, XB = XP * 1024;
  static constexpr int STAGE = NOB * DYB + NCB * XB;
  static constexpr int NPIECE = NOB * DYP + NCB * XP;
  static constexpr int PR = (NPIECE + 7) / 8;                         // DMA pieces per wave per tile
  static constexpr int NST = 3 * STAGE <= 160 * 1024 ? 3 : 2;         // stages: the tile being read + NST - 1 in flight
  static constexpr int LDS = NST * STAGE;
  static constexpr int BO = 32 * NOB, BC = 32 * NCB;
};

template <int NOB, int NCB, int KS>
__global__ __launch_bounds__(512) void wgrad3_kernel(const WgradParams p) {
  typedef Wg3Cfg<NOB, NCB, KS> Cfg;
  static_assert(NOB * NCB * KS == 8, "8 waves");
  constexpr int TH = Cfg::TH, TW = Cfg::TW, HS = Cfg::HS, HH = Cfg::HH, NTAPS = Cfg::NTAPS, PR = Cfg::PR;
  constexpr int DYP = Cfg::DYP, XP = Cfg::XP, DYB = Cfg::DYB, XB = Cfg::XB, STAGE = Cfg::STAGE, NST = Cfg::NST;
  constexpr int ROWS = TH / KS, KSTEPS = ROWS * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) bf16x4* tr_ptr_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = wave % KS, cb = (wave / KS) % NCB, ob = wave / (KS * NCB);
  const int h = lane >> 5, l32 = lane & 31;
  const int pairs = p.otiles * p.ctiles;
  const int tilesX = (p.OW + TW - 1) / TW, tilesY = (p.OH + TH - 1) / TH;
  const int tpi = tilesX * tilesY;
  const Wg3Walk walk = wg3_walk(blockIdx.x, pairs, gridDim.x / pairs, p.N * tpi, p.xcd);
  const int pair = walk.pair, g = walk.g, G = walk.step, ntiles = walk.end;
  const int o0 = (pair / p.ctiles) * Cfg::BO, c0 = (pair % p.ctiles) * Cfg::BC;
  const int pad = -p.dy_[0];
  const unsigned char* __restrict__ xg = (const unsigned char*)p.x;
  const unsigned char* __restrict__ dyg = (const unsigned char*)p.dy;

  // ---- DMA pieces of this wave: round r moves piece u = r*8 + wave (wrapped: the spare slots of the last
  // round repeat early pieces, same bytes to the same place, so every wave issues exactly PR per tile)
  const unsigned smem_base = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr_of(smem));
  unsigned poff[PR];        // byte offset from the tile's origin pixel (dy: (ty0,tx0); x: (ty0-pad, tx0-pad))
  int pyx[PR];              // local (y << 16 | x) of the lane's pixel; y = 0x4000 marks a lane that always reads zeros
  int pdst[PR];             // LDS offset inside a stage (wave-uniform)
  bool pisdy[PR];
#pragma unroll
  for (int r = 0; r < PR; ++r) {
    int u = r * 8 + wave;
    if (u >= Cfg::NPIECE) u -= Cfg::NPIECE;
    const bool isdy = u < NOB * DYP;
    const int v = isdy ? u : u - NOB * DYP;
    const int chunk = isdy ? v / DYP : v / XP, grp = isdy ? v % DYP : v % XP;
    const int lpx = grp * 16 + (lane >> 2), slot = lane & 3;
    pisdy[r] = isdy;
    pdst[r] = isdy ? chunk * DYB + grp * 1024 : NOB * DYB + chunk * XB + grp * 1024;
    if (isdy) {
      const int y = lpx / TW, x = lpx % TW;
      const bool ok = o0 + chunk * 32 < p.Os;
      poff[r] = (unsigned)((((y * p.OW + x) * p.Os) + o0 + chunk * 32) * 2 + slot * 16);
      pyx[r] = ((ok ? y : 0x4000) << 16) | x;
    } else {
      const int y = lpx / HS, x = lpx % HS;
      const bool ok = c0 + chunk * 32 < p.Cs && lpx < HH * HS;
      poff[r] = (unsigned)((((y * p.W + x) * p.Cs) + c0 + chunk * 32) * 2 + slot * 16);
      pyx[r] = ((ok ? y : 0x4000) << 16) | x;
    }
  }
  auto tile_pos = [&](int t, int& img, int& ty0, int& tx0) {
    img = t / tpi;
    const int rem = t - img * tpi;
    const int ty = rem / tilesX;
    ty0 = ty * TH; tx0 = (rem - ty * tilesX) * TW;
  };
  // One piece = ~18 instructions, half of them scalar: the tile's two origins are buffer resources, a lane outside its
  // image (or in the padding of the piece grid: row 0x4000) gets offset 0xFFFFFFFF and the DMA writes zeros for it
  // (lds_dma16_buf).  Round 2 selected between a 64-bit address and a block of zeros behind wave-uniform and divergent
  // branches: 36 instructions per piece, 8 pieces per wave and tile.
  struct TileSrc { vg_srd_t dy, x; int ty0, tx0; };      // the two origins of a tile as buffer resources (once per tile)
  auto tile_src = [&](int img, int ty0, int tx0) {
    TileSrc ts;
    ts.dy = vg_make_srd(dyg + ((size_t)((size_t)img * p.OH + ty0) * p.OW + tx0) * p.Os * 2);
    ts.x = vg_make_srd(xg + ((long)((long)img * p.H + ty0 - pad) * p.W + tx0 - pad) * p.Cs * 2);
    ts.ty0 = ty0; ts.tx0 = tx0;
    return ts;
  };
  auto issue_piece = [&](int r, const TileSrc& ts, int stage) {
    const int ly = pyx[r] >> 16, lx = pyx[r] & 0xffff;
    const bool isdy = pisdy[r];                              // wave-uniform
    const int oy = isdy ? ts.ty0 : ts.ty0 - pad, ox = isdy ? ts.tx0 : ts.tx0 - pad;
    const unsigned limy = isdy ? (unsigned)p.OH : (unsigned)p.H, limx = isdy ? (unsigned)p.OW : (unsigned)p.W;
    const bool ok = ((unsigned)(oy + ly) < limy) & ((unsigned)(ox + lx) < limx);
    vg_srd_t srd = ts.x;
    if (isdy) srd = ts.dy;
    lds_dma16_buf(srd, ok ? poff[r] : 0xFFFFFFFFu, smem_base + (unsigned)(stage * STAGE + pdst[r]));
  };

  // v_mfma_f32_16x16x32_bf16 (the chip holds a higher clock on it than on 32x32x16): a tile row of 32 pixels is the K of
  // one MFMA, the 32 x 32 block of a tap is four 16 x 16 blocks (o half oh, c half ch).  acc[tap][oh*2 + ch][r]:
  // o = oh*16 + 4*kb + r, c = ch*16 + m16.
  f32x4 acc[NTAPS][4];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposed-read lane term: lane (tq = li>>2, tp = li&3) of a 16-lane group addresses pixel row tq, channels 4tp..4tp+3;
  // the group kb = lane / 16 holds pixels 8*kb .. 8*kb + 7 of the row (the 16-channel half is + 32 bytes)
  const int m16 = lane & 15, kb = lane >> 4;
  const int li = lane & 15, tq = li >> 2, tp = li & 3;
  const int lterm = (8 * kb + tq) * 64 + (4 * tp) * 2;
  const int abase = ob * DYB + ks * TW * 64 + lterm;                 // dy image of this wave's o block, first own row
  const int bbase = NOB * DYB + cb * XB + ks * HS * 64 + lterm;      // x halo image of this wave's c block

  int t = walk.t0;
  {
    int img, ty0, tx0;
#pragma unroll
    for (int pre = 0; pre < NST - 1; ++pre) {
      if (t + pre * G < ntiles) {
        tile_pos(t + pre * G, img, ty0, tx0);
        const TileSrc ts = tile_src(img, ty0, tx0);
#pragma unroll
        for (int r = 0; r < PR; ++r) issue_piece(r, ts, pre);
      }
    }
  }
  int stage = 0;
  for (; t < ntiles; t += G) {
    if (!VG_DBG(64)) {
      if (NST == 3 && t + G < ntiles) wait_vmcnt<PR>();              // the next tile's pieces may stay in flight
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("" ::: "memory");
    const bool more = t + (NST - 1) * G < ntiles;
    int nimg = 0, nty0 = 0, ntx0 = 0;
    if (more) tile_pos(t + (NST - 1) * G, nimg, nty0, ntx0);
    const int nstage = stage + NST - 1 >= NST ? stage - 1 : stage + NST - 1;
    const unsigned char* sA = smem + stage * STAGE + abase;
    const unsigned char* sB = smem + stage * STAGE + bbase;
    // prefetch of tile t+2G, all of it before the tile's own work: the stream is latency-bound, and the earlier the
    // pieces are in flight the better (spread over the k steps: 0.484 ms, all at once: 0.407 ms on 64->64@256^2)
    if (more && !VG_DBG(32)) {
      const TileSrc ts = tile_src(nimg, nty0, ntx0);
#pragma unroll
      for (int r = 0; r < PR; ++r) issue_piece(r, ts, nstage);
    }
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      const int y = j * KS;                                           // + ks (in the base)
      u32x4 a[2];
#pragma unroll
      for (int oh = 0; oh < 2; ++oh) {
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tr_ptr_t)(sA + (y * TW) * 64 + oh * 32));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tr_ptr_t)(sA + (y * TW + 4) * 64 + oh * 32));
        a[oh] = __builtin_bit_cast(u32x4, bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
      }
#pragma unroll
      for (int tap = 0; tap < NTAPS; ++tap) {
        const int hp = VG_DBG(16) ? (y * HS) : (y + tap / 3) * HS + tap % 3;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tr_ptr_t)(sB + hp * 64 + ch * 32));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tr_ptr_t)(sB + (hp + 4) * 64 + ch * 32));
          const u32x4 b = __builtin_bit_cast(u32x4, bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
#pragma unroll
          for (int oh = 0; oh < 2; ++oh) acc[tap][oh * 2 + ch] = mma16x16x32(a[oh], b, acc[tap][oh * 2 + ch]);
        }
      }
    }
    stage = stage + 1 == NST ? 0 : stage + 1;
  }

  // acc[tap][oh*2 + ch][r]: row (o) = oh*16 + 4*kb + r, column (c) = ch*16 + m16
  // Every (tile group, row split) pair owns one partial image of dW and writes it with plain stores; the unpack kernel
  // sums the partials in a fixed order.  (One pass of f32 atomics into a single image cost 50-100 us per launch, a
  // fifth of the kernel, and made the gradient depend on the order the blocks finished in.)
  // With two row splits the second wave of each (o, c) block hands its accumulators to the first through the (now
  // idle) stage buffers, five taps at a time, so a block writes ONE partial image (half the partial traffic).
  if constexpr (KS == 2) {
    static_assert(Cfg::LDS >= 4 * 5 * 16 * 64 * 4, "exchange buffer");
    float* xch = (float*)smem + (size_t)(wave / KS) * (5 * 16 * 64) + lane;
    __syncthreads();                                   // every wave is done with the stage buffers
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      constexpr int T0[2] = {0, 5}, T1[2] = {5, NTAPS};
      if (ks == 1) {
#pragma unroll
        for (int tap = T0[half]; tap < T1[half]; ++tap)
#pragma unroll
          for (int r = 0; r < 16; ++r) xch[((tap - T0[half]) * 16 + r) * 64] = acc[tap][r >> 2][r & 3];
      }
      __syncthreads();
      if (ks == 0) {
#pragma unroll
        for (int tap = T0[half]; tap < T1[half]; ++tap)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[tap][r >> 2][r & 3] += xch[((tap - T0[half]) * 16 + r) * 64];
      }
      __syncthreads();
    }
    if (ks != 0) return;
  }
  const size_t rowstride = (size_t)NTAPS * p.Cpad;
  float* __restrict__ part = p.dw + (size_t)(KS == 2 ? g : g * KS + ks) * p.part_elems;
  if (!VG_DBG(128)) {
#pragma unroll
    for (int tap = 0; tap < NTAPS; ++tap)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = c0 + cb * 32 + (q & 1) * 16 + m16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = o0 + ob * 32 + (q >> 1) * 16 + 4 * kb + r;
          if (c < p.Cpad && o < p.Opad) part[(size_t)o * rowstride + (size_t)tap * p.Cpad + c] = acc[tap][q][r];
        }
      }
  }
  if (VG_DBG(128) && acc[0][0][0] == 123.f) p.dw[0] = acc[1][0][1] + acc[8][3][3];    // keep the accumulators alive
}

template <int NOB, int NCB, int KS>
static int launch_wgrad3(const WgradParams& p0, hipStream_t st) {
  typedef Wg3Cfg<NOB, NCB, KS> Cfg;
  WgradParams p = p0;
  p.dbg = g_dbg;
  p.otiles = (int)vg_ceil_div(p.Opad, Cfg::BO);
  p.ctiles = (int)vg_ceil_div(p.Cpad, Cfg::BC);
  const int pairs = p.otiles * p.ctiles;
  const int G = wg3_groups(pairs, p.N, p.OH, p.OW, Cfg::TH, Cfg::TW);
  p.xcd = wg3_xcd_dealing();
  auto kern = wgrad3_kernel<NOB, NCB, KS>;
    VG_HIP(vg_lds_attr((const void*)kern, Cfg::LDS));
  hipLaunchKernelGGL(kern, dim3(pairs * G), dim3(512), Cfg::LDS, st, p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// =============================================================================================
// wgrad3s2: the same for 4x4 stride-2 pad-1 convolutions (bf16).  x[2oy+r-1][2ox+s-1] is a stride-1 gather on
// the four parity planes X_ab[i][j] = x[2i-a][2j-b] (a = (r+1)&1, row offset r>>1; likewise b, s), so the
// halo of a 4 x 16 output tile is kept in LDS as four 5 x 17 pixel planes and every tap again reads consecutive
// pixel rows (conflict-free transposed reads at immediate offsets).  A wave keeps one 32 x 32 block for the 8
// taps of one row parity (128 accumulators); waves = NOB x NCB x 2 tap halves x KS row splits.
// =============================================================================================
#ifndef VG_WG3S2_TH411
#define VG_WG3S2_TH411 8
#endif
static constexpr int wg3s2_th(int nob, int ncb, int ks) { return (nob == 4 && ncb == 1 && ks == 1) ? VG_WG3S2_TH411 : 4; }
template <int NOB, int NCB, int KS>
struct Wg3s2Cfg {
  static constexpr int TH = wg3s2_th(NOB, NCB, KS), TW = 16, PH = TH + 1, PW = TW + 1, PLANE = PH * PW;      // 85 pixels per parity plane (TH = 4)
  static constexpr int DYP = TH * TW / 16;                           // 4 pieces per dy chunk image
  static constexpr int XP = (4 * PLANE + 15) / 16;                   // 22 pieces per x chunk image (4 planes)
  static constexpr int DYB = DYP * 1024, XB = XP * 1024;
  static constexpr int STAGE = NOB * DYB + NCB * XB;
  static constexpr int NPIECE = NOB * DYP + NCB * XP;
  static constexpr int PR = (NPIECE + 7) / 8;
  static constexpr int NST = 3 * STAGE <= 160 * 1024 ? 3 : 2;
  static constexpr int LDS = NST * STAGE;
  static constexpr int BO = 32 * NOB, BC = 32 * NCB;
};

template <int NOB, int NCB, int KS>
__global__ __launch_bounds__(512) void wgrad3s2_kernel(const WgradParams p) {
  typedef Wg3s2Cfg<NOB, NCB, KS> Cfg;
  static_assert(NOB * NCB * KS * 2 == 8, "8 waves");
  constexpr int TH = Cfg::TH, TW = Cfg::TW, PW = Cfg::PW, PLANE = Cfg::PLANE, PR = Cfg::PR;
  constexpr int DYP = Cfg::DYP, XP = Cfg::XP, DYB = Cfg::DYB, XB = Cfg::XB, STAGE = Cfg::STAGE, NST = Cfg::NST;
  constexpr int ROWS = TH / KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) bf16x4* tr_ptr_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int th = wave & 1;                                // tap half: 0 -> kernel rows 1,3 (plane a=0), 1 -> rows 0,2 (a=1)
  const int ks = (wave >> 1) % KS, cb = (wave / (2 * KS)) % NCB, ob = wave / (2 * KS * NCB);
  const int h = lane >> 5, l32 = lane & 31;
  const int pairs = p.otiles * p.ctiles;
  const int tilesX = (p.OW + TW - 1) / TW, tilesY = (p.OH + TH - 1) / TH;
  const int tpi = tilesX * tilesY;
  const Wg3Walk walk = wg3_walk(blockIdx.x, pairs, gridDim.x / pairs, p.N * tpi, p.xcd);
  const int pair = walk.pair, g = walk.g, G = walk.step, ntiles = walk.end;
  const int o0 = (pair / p.ctiles) * Cfg::BO, c0 = (pair % p.ctiles) * Cfg::BC;
  const unsigned char* __restrict__ xg = (const unsigned char*)p.x;
  const unsigned char* __restrict__ dyg = (const unsigned char*)p.dy;

  const unsigned smem_base = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr_of(smem));
  unsigned poff[PR];        // dy: from pixel (ty0, tx0); x: from input pixel (2*ty0 - 1, 2*tx0 - 1)
  int pyx[PR];              // dy: local (y << 16 | x); x: input offset (dy+1 << 16 | dx+1) from that origin; y = 0x4000: zeros
  int pdst[PR];
  bool pisdy[PR];
#pragma unroll
  for (int r = 0; r < PR; ++r) {
    int u = r * 8 + wave;
    if (u >= Cfg::NPIECE) u -= Cfg::NPIECE;
    const bool isdy = u < NOB * DYP;
    const int v = isdy ? u : u - NOB * DYP;
    const int chunk = isdy ? v / DYP : v / XP, grp = isdy ? v % DYP : v % XP;
    const int lpx = grp * 16 + (lane >> 2), slot = lane & 3;
    pisdy[r] = isdy;
    pdst[r] = isdy ? chunk * DYB + grp * 1024 : NOB * DYB + chunk * XB + grp * 1024;
    if (isdy) {
      const int y = lpx / TW, x = lpx % TW;
      const bool ok = o0 + chunk * 32 < p.Os;
      poff[r] = (unsigned)((((y * p.OW + x) * p.Os) + o0 + chunk * 32) * 2 + slot * 16);
      pyx[r] = ((ok ? y : 0x4000) << 16) | x;
    } else {
      const int plane = lpx / PLANE, rem = lpx % PLANE;
      const int li = rem / PW, lj = rem % PW;
      const int yy = 2 * li - (plane >> 1) + 1, xx = 2 * lj - (plane & 1) + 1;       // >= 0
      const bool ok = c0 + chunk * 32 < p.Cs && lpx < 4 * PLANE;
      poff[r] = (unsigned)((((yy * p.W + xx) * p.Cs) + c0 + chunk * 32) * 2 + slot * 16);
      pyx[r] = ((ok ? yy : 0x4000) << 16) | xx;
    }
  }
  auto tile_pos = [&](int t, int& img, int& ty0, int& tx0) {
    img = t / tpi;
    const int rem = t - img * tpi;
    const int ty = rem / tilesX;
    ty0 = ty * TH; tx0 = (rem - ty * tilesX) * TW;
  };
  struct TileSrc { vg_srd_t dy, x; int ty0, tx0; };      // the two origins of a tile as buffer resources (once per tile)
  auto tile_src = [&](int img, int ty0, int tx0) {
    TileSrc ts;
    ts.dy = vg_make_srd(dyg + ((size_t)((size_t)img * p.OH + ty0) * p.OW + tx0) * p.Os * 2);
    ts.x = vg_make_srd(xg + ((long)((long)img * p.H + 2 * ty0 - 1) * p.W + 2 * tx0 - 1) * p.Cs * 2);
    ts.ty0 = ty0; ts.tx0 = tx0;
    return ts;
  };
  auto issue_piece = [&](int r, const TileSrc& ts, int stage) {
    const int ly = pyx[r] >> 16, lx = pyx[r] & 0xffff;
    const bool isdy = pisdy[r];                              // wave-uniform
    const int oy = isdy ? ts.ty0 : 2 * ts.ty0 - 1, ox = isdy ? ts.tx0 : 2 * ts.tx0 - 1;
    const unsigned limy = isdy ? (unsigned)p.OH : (unsigned)p.H, limx = isdy ? (unsigned)p.OW : (unsigned)p.W;
    const bool ok = ((unsigned)(oy + ly) < limy) & ((unsigned)(ox + lx) < limx);
    vg_srd_t srd = ts.x;
    if (isdy) srd = ts.dy;
    lds_dma16_buf(srd, ok ? poff[r] : 0xFFFFFFFFu, smem_base + (unsigned)(stage * STAGE + pdst[r]));
  };

  f32x16 acc[8];            // local tap lt = u*4 + s: kernel row r = 2u + 1 - th, column s
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;
  const int lterm = (8 * h + tq) * 64 + (16 * g16 + 4 * tp) * 2;
  const int abase = ob * DYB + ks * TW * 64 + lterm;
  const int bbase = NOB * DYB + cb * XB + (th * 2 * PLANE + ks * PW) * 64 + lterm;     // planes (a = th, b = 0..1)

  int t = walk.t0;
  {
    int img, ty0, tx0;
#pragma unroll
    for (int pre = 0; pre < NST - 1; ++pre) {
      if (t + pre * G < ntiles) {
        tile_pos(t + pre * G, img, ty0, tx0);
        const TileSrc ts = tile_src(img, ty0, tx0);
#pragma unroll
        for (int r = 0; r < PR; ++r) issue_piece(r, ts, pre);
      }
    }
  }
  int stage = 0;
  for (; t < ntiles; t += G) {
    if (NST == 3 && t + G < ntiles) wait_vmcnt<PR>();                 // the next tile's pieces may stay in flight
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const bool more = t + (NST - 1) * G < ntiles;
    int nimg = 0, nty0 = 0, ntx0 = 0;
    if (more) tile_pos(t + (NST - 1) * G, nimg, nty0, ntx0);
    const int nstage = stage + NST - 1 >= NST ? stage - 1 : stage + NST - 1;
    const unsigned char* sA = smem + stage * STAGE + abase;
    const unsigned char* sB = smem + stage * STAGE + bbase;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      const int y = j * KS;                                           // + ks (in the bases)
      bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tr_ptr_t)(sA + (y * TW) * 64));
      bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tr_ptr_t)(sA + (y * TW + 4) * 64));
      const bf16x8 a = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
#pragma unroll
      for (int lt = 0; lt < 8; ++lt) {
        const int u = lt >> 2, s = lt & 3;
        const int b = (s + 1) & 1;                                    // column parity plane
        const int hp = b * PLANE + (y + u) * PW + (s >> 1);
        bf16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tr_ptr_t)(sB + hp * 64));
        bf16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tr_ptr_t)(sB + (hp + 4) * 64));
        const bf16x8 bf = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
        acc[lt] = Mma<bf16_t>::run(__builtin_bit_cast(u32x4, a), __builtin_bit_cast(u32x4, bf), acc[lt]);
      }
      if (more && j == 0) {
        const TileSrc ts = tile_src(nimg, nty0, ntx0);
#pragma unroll
        for (int r = 0; r < PR; ++r) issue_piece(r, ts, nstage);
      }
    }
    stage = stage + 1 == NST ? 0 : stage + 1;
  }

  const size_t rowstride = (size_t)16 * p.Cpad;
  float* __restrict__ part = p.dw + (size_t)(g * KS + ks) * p.part_elems;      // see wgrad3_kernel
  const int c = c0 + cb * 32 + l32;
  if (c < p.Cpad) {
#pragma unroll
    for (int lt = 0; lt < 8; ++lt) {
      const int tap = (2 * (lt >> 2) + 1 - th) * 4 + (lt & 3);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (o < p.Opad) part[(size_t)o * rowstride + (size_t)tap * p.Cpad + c] = acc[lt][r];
      }
    }
  }
}

template <int NOB, int NCB, int KS>
static int launch_wgrad3s2(const WgradParams& p0, hipStream_t st) {
  typedef Wg3s2Cfg<NOB, NCB, KS> Cfg;
  WgradParams p = p0;
  p.otiles = (int)vg_ceil_div(p.Opad, Cfg::BO);
  p.ctiles = (int)vg_ceil_div(p.Cpad, Cfg::BC);
  const int pairs = p.otiles * p.ctiles;
  const int G = wg3_groups(pairs, p.N, p.OH, p.OW, Cfg::TH, Cfg::TW);
  p.xcd = wg3_xcd_dealing();
  auto kern = wgrad3s2_kernel<NOB, NCB, KS>;
    VG_HIP(vg_lds_attr((const void*)kern, Cfg::LDS));
  hipLaunchKernelGGL(kern, dim3(pairs * G), dim3(512), Cfg::LDS, st, p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

static int g_wgrad_kernel = 4;      // 1: per-tap kernel only; 3: + spatial-tile kernels with 64 x 64 slabs; 4: + 128 x 64 slabs (default)
extern "C" void vg_set_wgrad_kernel(int version) { g_wgrad_kernel = version; }

// Which spatial-tile configuration takes a weight gradient (0 = none: the per-tap kernel), and how many partial images
// of dW it writes (tile groups x row splits).  Shared by the workspace query and the launch.
struct Wg3Plan { int kind, nob, ncb, ks, parts; };     // kind 1: 3x3 stride 1, 2: 4x4 stride 2
static Wg3Plan wg3_plan(const WgradParams& p, int R, int S, int dtype) {
  Wg3Plan q = {0, 0, 0, 0, 1};
  if (g_wgrad_kernel < 3 || dtype != VG_BF16) return q;
  if ((size_t)p.H * p.W * p.Cs * 2 >= ((size_t)1 << 31) || (size_t)p.OH * p.OW * p.Os * 2 >= ((size_t)1 << 31))
    return q;                                                // piece offsets are 32-bit inside one image
  int th, tw;
  if (p.in_s == 2 && R == 4 && S == 4 && p.dy_[0] == -1 && p.dx_[0] == -1 && p.OW >= 8) {
    q.kind = 2; tw = 16;
    if (p.Opad >= 128) { q.nob = 4; q.ncb = 1; q.ks = 1; }
    else if (p.Opad > 32) { q.nob = 2; q.ncb = 1; q.ks = 2; }
    else { q.nob = 1; q.ncb = 1; q.ks = 4; }
    th = wg3s2_th(q.nob, q.ncb, q.ks);
  } else if (p.in_s == 1 && R == 3 && S == 3 && p.OW >= 24) {
    q.kind = 1; tw = 32;
    if (p.Opad <= 32) { q.nob = 1; q.ncb = 2; q.ks = 4; }
    else if (p.Cpad <= 32) { q.nob = 2; q.ncb = 1; q.ks = 4; }
    else if (g_wgrad_kernel >= 4 && p.Opad >= 128) { q.nob = 4; q.ncb = 2; q.ks = 1; }      // 128 x 64 slab, no row split
    else { q.nob = 2; q.ncb = 2; q.ks = 2; }
    th = wg3_th(q.nob, q.ncb, q.ks);
  } else {
    return q;
  }
  const int pairs = (int)vg_ceil_div(p.Opad, 32 * q.nob) * (int)vg_ceil_div(p.Cpad, 32 * q.ncb);
  // row splits: the 64x64-slab 3x3 kernel combines its two inside the block, the others write one image per split
  q.parts = wg3_groups(pairs, p.N, p.OH, p.OW, th, tw) * (q.kind == 1 && q.ks == 2 ? 1 : q.ks);
  return q;
}
static thread_local char g_last_wgrad[48] = "";
extern "C" const char* vg_last_wgrad_kernel(void) { return g_last_wgrad; }
static int dispatch_wgrad3(const WgradParams& p, const Wg3Plan& q, hipStream_t st) {
  snprintf(g_last_wgrad, sizeof g_last_wgrad, "%s<%d,%d,%d>", q.kind == 2 ? "wgrad3s2" : "wgrad3", q.nob, q.ncb, q.ks);
  if (q.kind == 2) {
    if (q.nob == 4) return launch_wgrad3s2<4, 1, 1>(p, st);
    if (q.nob == 2) return launch_wgrad3s2<2, 1, 2>(p, st);
    return launch_wgrad3s2<1, 1, 4>(p, st);
  }
  if (q.nob == 4) return launch_wgrad3<4, 2, 1>(p, st);
  if (q.nob == 1) return launch_wgrad3<1, 2, 4>(p, st);
  if (q.ncb == 1) return launch_wgrad3<2, 1, 4>(p, st);
  return launch_wgrad3<2, 2, 2>(p, st);
}

template <typename T> static int dispatch_wgrad(const WgradParams& p, hipStream_t st) {
  const bool o128 = p.Opad >= 128, c128 = p.Cpad >= 128;
  const bool o32 = p.Opad <= 32, c32 = p.Cpad <= 32;
  if (o128 && c128) return launch_wgrad<T, 2, 2, 2, 2, 1>(p, st);      // 128 x 128
  if (o128 && c32) return launch_wgrad<T, 2, 1, 2, 1, 2>(p, st);       // 128 x 32
  if (o128) return launch_wgrad<T, 2, 2, 2, 1, 2>(p, st);              // 128 x 64
  if (o32 && c128) return launch_wgrad<T, 1, 2, 1, 2, 2>(p, st);       // 32 x 128
  if (o32) return launch_wgrad<T, 1, 2, 1, 1, 4>(p, st);               // 32 x 64 (covers c<=64)
  if (c128) return launch_wgrad<T, 2, 2, 1, 2, 2>(p, st);              // 64 x 128
  if (c32) return launch_wgrad<T, 2, 1, 1, 1, 4>(p, st);               // 64 x 32
  return launch_wgrad<T, 2, 2, 1, 1, 4>(p, st);                        // 64 x 64
}

// dw[o][i][rs] (+)= sum over the partial images of packed[part][o][rs][i], in a fixed order: a block takes 64
// consecutive (o, rs, i) items (i fastest: coalesced reads), its four waves split the partials, LDS combines them.
__global__ __launch_bounds__(256) void wgrad_unpack_kernel(const float* __restrict__ packed, int parts, size_t part_elems,
                                                           float* __restrict__ dw, int O, int I, int RS, int Cpad,
                                                           int accumulate) {
  __shared__ float red[4][64];
  const long total = (long)O * I * RS;
  const int e = threadIdx.x & 63, gy = threadIdx.x >> 6;
  const long t = blockIdx.x * 64L + e;
  float v = 0.f;
  long dst = 0;
  if (t < total) {
    const int i = (int)(t % I);
    const long ors = t / I;
    const int rs = (int)(ors % RS), o = (int)(ors / RS);
    const float* src = packed + (size_t)ors * Cpad + i;
    // eight loads in flight per thread (the partial images are megabytes apart: one dependent chain of loads per
    // thread left the kernel latency-bound); the order of the additions is fixed, so the sum is reproducible
    int g = gy;
    float a[8];
    for (; g + 28 < parts; g += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = src[(size_t)(g + 4 * u) * part_elems];
      v += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    for (; g < parts; g += 4) v += src[(size_t)g * part_elems];
    dst = ((long)o * I + i) * RS + rs;
  }
  red[gy][e] = v;
  __syncthreads();
  if (gy == 0 && t < total) {
    const float sum = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
    dw[dst] = accumulate ? dw[dst] + sum : sum;
  }
}

static size_t wgrad_packed_elems(const vg_conv_desc* d) {
  int opad = (int)vg_ceil_div(d->co, 32) * 32;
  return (size_t)opad * d->r * d->s * cpad_of(d->ci, d->dtype);
}
static void wgrad_fill_params(const vg_conv_desc* d, WgradParams& p) {
  memset(&p, 0, sizeof(p));
  p.N = d->n; p.H = d->h; p.W = d->w; p.Cpad = cpad_of(d->ci, d->dtype); p.Cs = d->xcs;
  p.OH = d->oh; p.OW = d->ow; p.Opad = (int)vg_ceil_div(d->co, 32) * 32; p.Os = d->ycs;
  if (p.Opad > p.Os) p.Opad = p.Os;   // f32 strides are multiples of 16: never read past the row
  p.in_s = d->stride; p.ntaps = d->r * d->s;
  p.M = d->n * d->oh * d->ow;
  for (int r = 0; r < d->r; ++r)
    for (int s = 0; s < d->s; ++s) { p.dy_[r * d->s + s] = (signed char)(r - d->pad); p.dx_[r * d->s + s] = (signed char)(s - d->pad); }
}

extern "C" size_t vg_conv_wgrad_ws_bytes(const vg_conv_desc* d) {
  if (!d) return 0;
  WgradParams p;
  wgrad_fill_params(d, p);
  const Wg3Plan q = wg3_plan(p, d->r, d->s, d->dtype);
  return wgrad_packed_elems(d) * sizeof(float) * (size_t)q.parts;
}

extern "C" int vg_conv_wgrad(const vg_conv_desc* d, const void* x, const void* dy, float* dw_oihw, void* ws,
                             int accumulate, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  VG_REQUIRE(x && dy && dw_oihw && ws, "conv_wgrad: null pointer");
  hipStream_t st = (hipStream_t)stream;
  WgradParams p;
  wgrad_fill_params(d, p);
  p.x = x; p.dy = dy; p.dw = (float*)ws;
  p.part_elems = wgrad_packed_elems(d);
  const Wg3Plan q = wg3_plan(p, d->r, d->s, d->dtype);
  if (q.kind) {
    rc = dispatch_wgrad3(p, q, st);                          // plain stores into q.parts partial images: nothing to clear
  } else {
    VG_HIP(hipMemsetAsync(ws, 0, p.part_elems * sizeof(float), st));
    snprintf(g_last_wgrad, sizeof g_last_wgrad, "wgrad<%s,per-tap>", d->dtype == VG_BF16 ? "bf16" : "f32");
    rc = d->dtype == VG_BF16 ? dispatch_wgrad<bf16_t>(p, st) : dispatch_wgrad<float>(p, st);
  }
  if (rc) return rc;
  const long total = (long)d->co * d->ci * d->r * d->s;
  hipLaunchKernelGGL(wgrad_unpack_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, (const float*)ws, q.parts,
                     p.part_elems, dw_oihw, d->co, d->ci, d->r * d->s, p.Cpad, accumulate);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// Does `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer resource) write ZEROS to LDS for lanes whose offset is
// out of the resource's range?  If so, an out-of-image halo piece needs no address select against a block of zeros:
// the lane's offset is replaced by 0xFFFFFFFF (one v_cndmask with an inline constant).
// Build: hipcc -O3 --offload-arch=gfx950 buf_lds.hip -o buf_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const unsigned char* src, unsigned num_records, u32x4* out) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[128];
  const int lane = threadIdx.x;
  lds[lane] = u32x4{0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu};
  lds[lane + 64] = u32x4{0xBBBBBBBBu, 0xBBBBBBBBu, 0xBBBBBBBBu, 0xBBBBBBBBu};
  __syncthreads();
  const uint64_t base = (uint64_t)src;
  i32x4 srd;
  srd[0] = __builtin_amdgcn_readfirstlane((int)(base & 0xFFFFFFFFu));
  srd[1] = __builtin_amdgcn_readfirstlane((int)((base >> 32) & 0xFFFFu));       // stride 0
  srd[2] = __builtin_amdgcn_readfirstlane((int)num_records);
  srd[3] = 0x00020000;
  // lanes 0..39: offset lane*16 (lanes 32..39 are beyond num_records = 512); lanes 40..63: offset 0xFFFFFFFF
  const unsigned voff = lane < 40 ? (unsigned)lane * 16u : 0xFFFFFFFFu;
  const unsigned ldsbase = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)lds;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)"
               : : "s"(ldsbase), "v"(voff), "s"(srd) : "memory", "m0");
  __syncthreads();
  out[lane] = lds[lane];
  out[lane + 64] = lds[lane + 64];
}

int main() {
  std::vector<unsigned> h(4096);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x1000u + (unsigned)i;
  unsigned char* d; u32x4* out;
  CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&out, 128 * 16));
  CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 512u, out);
  CK(hipDeviceSynchronize());
  std::vector<unsigned> r(128 * 4);
  CK(hipMemcpy(r.data(), out, 128 * 16, hipMemcpyDeviceToHost));
  int ok_in = 0, zero_rng = 0, zero_big = 0, stale = 0;
  for (int l = 0; l < 64; ++l) {
    const unsigned* v = &r[l * 4];
    const bool is_data = v[0] == 0x1000u + l * 4 && v[3] == 0x1000u + l * 4 + 3;
    const bool is_zero = v[0] == 0 && v[1] == 0 && v[2] == 0 && v[3] == 0;
    const bool is_stale = v[0] == 0xAAAAAAAAu;
    if (l < 32) ok_in += is_data;
    else if (l < 40) zero_rng += is_zero;
    else zero_big += is_zero;
    stale += is_stale;
    if (l == 0 || l == 31 || l == 32 || l == 39 || l == 40 || l == 63) printf("lane %2d: %08x %08x %08x %08x\n", l, v[0], v[1], v[2], v[3]);
  }
  int second_untouched = 0;
  for (int l = 64; l < 128; ++l) second_untouched += r[l * 4] == 0xBBBBBBBBu;
  printf("in-range lanes with data: %d/32; beyond num_records zeroed: %d/8; offset 0xFFFFFFFF zeroed: %d/24; stale: %d; rows 64..127 untouched: %d/64\n",
         ok_in, zero_rng, zero_big, stale, second_untouched);
  return 0;
}

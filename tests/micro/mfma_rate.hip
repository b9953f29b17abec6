// Calibration: s_memtime ticks per v_mfma_f32_32x32x16_bf16 with 4 independent accumulators, one or two waves per SIMD,
// operands in ordinary / accumulation registers; and ticks per microsecond.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int iters, const u32x4* src) {
  u32x4 a0 = src[threadIdx.x], a1 = src[threadIdx.x + 512], b0 = src[threadIdx.x + 1024], b1 = src[threadIdx.x + 1536];
  f32x16 c0, c1, c2, c3;
  for (int r = 0; r < 16; ++r) { c0[r] = 0; c1[r] = 0; c2[r] = 0; c3[r] = 0; }
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a0), "v"(b0));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a1), "v"(b0));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(a0), "v"(b1));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(a1), "v"(b1));
    } else if (MODE == 1) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a0), "a"(b0));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a1), "a"(b0));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(a0), "a"(b1));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(a1), "a"(b1));
    } else {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "v"(a0), "v"(b0));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(a1), "v"(b0));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c2) : "v"(a0), "v"(b1));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c3) : "v"(a1), "v"(b1));
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  if (s == 12345.f) out[1] = 1;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int MODE> void run(const char* name, int threads, int blocks) {
  unsigned long long* d; hipMalloc(&d, 16); hipMemset(d, 0, 16);
  u32x4* src; hipMalloc(&src, 2048 * 16); hipMemset(src, 0, 2048 * 16);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, threads>>>(d, 100, src);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, threads>>>(d, iters, src);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long t; hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
  const double mf = 4.0 * iters;
  printf("%-28s threads %3d blocks %3d: %.1f ticks/MFMA/wave, %.3f ms, ticks/us %.0f, chip %.0f TFLOP/s\n", name, threads, blocks,
         t / mf, ms, t / (ms * 1e3), mf * (threads / 64) * blocks * 32768.0 / (ms * 1e-3) / 1e12);
  hipFree(d); hipFree(src);
}
int main() {
  run<0>("B in VGPR, 1 wave/SIMD", 256, 256);
  run<1>("B in AGPR, 1 wave/SIMD", 256, 256);
  run<2>("acc in AGPR, 1 wave/SIMD", 256, 256);
  run<0>("B in VGPR, 2 waves/SIMD", 512, 256);
  run<0>("B in VGPR, 1 wave, 1 block", 64, 1);
  run<0>("B in VGPR, 1 wave/SIMD 32 blk", 256, 32);
  return 0;
}

// What the fp16 matrix pipe sustains on RANDOM operands (the clock under matrix load is set by power): dense
// TFLOP/s and in-kernel clock of bare loops of v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16 / the older
// v_mfma_f32_16x16x16_f16 (half the K per instruction), operands in
// registers, 1 or 2 waves per SIMD, zero or random data.  hipcc --offload-arch=gfx950 -O3 scripts/ubench_mfma_f16_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

template <int SHAPE>   // 0: 32x32x16, 4 accumulators; 1: 16x16x32, 16 accumulators; 2: 16x16x16, 16 accumulators
__global__ void __launch_bounds__(256) kern(float* out, unsigned long long* clk, int iters, int random) {
  unsigned s = 12345u + threadIdx.x * 7919u + blockIdx.x * 104729u;
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    u32x4 x, y;
    for (int k = 0; k < 4; ++k) {           // fp16 values in [0.5, 2) with random mantissas and signs
      x[k] = random ? ((rnd(s) & 0x83ff83ffu) | 0x38003800u) : 0u;
      y[k] = random ? ((rnd(s) & 0x83ff83ffu) | 0x38003800u) : 0u;
    }
    a[i] = __builtin_bit_cast(f16x8, x); b[i] = __builtin_bit_cast(f16x8, y);
  }
  f32x16 c32[4]; f32x4 c16[16];
  for (int i = 0; i < 4; ++i) for (int v = 0; v < 16; ++v) c32[i][v] = 0.f;
  for (int i = 0; i < 16; ++i) c16[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (SHAPE == 0) {
#pragma unroll
      for (int u = 0; u < 16; ++u) c32[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 3], b[(u >> 2) & 3], c32[u & 3], 0, 0, 0);
    } else if (SHAPE == 1) {
#pragma unroll
      for (int u = 0; u < 32; ++u) c16[u & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 3], b[(u >> 2) & 3], c16[u & 15], 0, 0, 0);
    } else {
#pragma unroll
      for (int u = 0; u < 64; ++u)
        c16[u & 15] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_shufflevector(a[u & 3], a[u & 3], 0, 1, 2, 3),
                                                            __builtin_shufflevector(b[(u >> 2) & 3], b[(u >> 2) & 3], 4, 5, 6, 7),
                                                            c16[u & 15], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
  for (int i = 0; i < 4; ++i) for (int v = 0; v < 16; ++v) sum += c32[i][v];
  for (int i = 0; i < 16; ++i) sum += c16[i][0] + c16[i][1] + c16[i][2] + c16[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
void run(const char* name, int waves_per_simd, int random, float* out, unsigned long long* clk) {
  const int blocks = 256 * waves_per_simd, iters = 200000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {       // the last of three back-to-back launches is the one reported
    hipEventRecord(e0);
    kern<SHAPE><<<blocks, 256>>>(out, clk, iters, random);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  const double flops = (double)blocks * 4 * iters * 16 * 32768.0;   // per wave and iteration: 16 x 32768 flop either shape
  printf("%-10s waves/SIMD=%d %-6s : %8.1f dense TFLOP/s, in-kernel clock %.2f GHz, %.1f cycles per 32768 flop per SIMD\n", name,
         waves_per_simd, random ? "random" : "zeros", flops / (ms * 1e-3) / 1e12, (double)h[0] / ((double)h[1] * 10.0) ,
         (double)h[0] / ((double)iters * 16 * waves_per_simd));
}

int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 512 * 256 * 4); hipMalloc(&clk, 512 * 16);
  for (int w = 1; w <= 2; ++w)
    for (int r = 0; r <= 1; ++r) { run<0>("32x32x16", w, r, out, clk); run<1>("16x16x32", w, r, out, clk); run<2>("16x16x16", w, r, out, clk); }
  return 0;
}

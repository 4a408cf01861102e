// cycles per (v_mfma_f32_16x16x32_bf16 + K vector instructions) on one SIMD, 1 or 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int K, int TYPE, int DEP>
__global__ void __launch_bounds__(256) kern(float* out, int iters, unsigned seed) {
  u32x4 a4 = {seed + threadIdx.x, seed * 3u, seed * 5u + threadIdx.x, seed * 7u};
  u32x4 b4 = {seed * 11u + threadIdx.x, seed * 13u, seed * 17u, seed * 19u + threadIdx.x};
  a4 &= 0x3f803f80u; b4 &= 0x3f803f80u;
  bf16x8 a = __builtin_bit_cast(bf16x8, a4), b = __builtin_bit_cast(bf16x8, b4);
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float x[8];
  unsigned m = 0xffff0000u;
  for (int i = 0; i < 8; ++i) x[i] = 1.0f + threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[DEP ? 0 : (u & 3)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[DEP ? 0 : (u & 3)], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float& r = x[(u * K + k) & 7];
        if (TYPE == 0) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r) : "v"(m));
        if (TYPE == 1) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r) : "v"(1.0e-9f));
        if (TYPE == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(x[(k + 1) & 7]), "s"(0x07060302u));
        if (TYPE == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r) : "v"(1.0000001f));
        if (TYPE == 4) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r) : "v"(1.0e-9f), "v"(x[(k + 1) & 7]));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int K, int TYPE, int DEP>
void run(const char* name, int blocks_per_cu, float* out) {
  const int iters = 4000, blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<K, TYPE, DEP><<<blocks, 256>>>(out, 100, 12345u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<K, TYPE, DEP><<<blocks, 256>>>(out, iters, 12345u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // each SIMD runs blocks_per_cu waves; per wave iters*16 MFMA slots
  const double slots = (double)iters * 16 * blocks_per_cu;
  printf("%-10s K=%d dep=%d waves/SIMD=%d : %.2f ns per (MFMA + K valu) per SIMD  -> %.1f cycles at 2.0 GHz\n", name, K, DEP,
         blocks_per_cu, ms * 1e6 / slots, ms * 1e6 / slots * 2.0);
}

int main() {
  float* out; hipMalloc(&out, 256 * 4 * 256 * 4);
  for (int w = 1; w <= 2; ++w) {
    run<0, 0, 0>("none", w, out);
    run<0, 0, 1>("none", w, out);
    run<2, 0, 0>("v_and", w, out); run<4, 0, 0>("v_and", w, out); run<6, 0, 0>("v_and", w, out);
    run<2, 1, 0>("v_sub", w, out); run<4, 1, 0>("v_sub", w, out); run<6, 1, 0>("v_sub", w, out);
    run<2, 2, 0>("v_perm", w, out); run<4, 2, 0>("v_perm", w, out);
    run<2, 3, 0>("v_mul", w, out); run<4, 3, 0>("v_mul", w, out);
    run<4, 4, 0>("v_fmac", w, out);
    run<4, 1, 1>("v_sub", w, out);
  }
  return 0;
}

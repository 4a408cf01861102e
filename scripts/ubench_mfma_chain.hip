// cycles per v_mfma_f32_16x16x32_bf16: accumulators visited round-robin vs in chains of 6, operands in VGPRs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NACC>
__global__ void __launch_bounds__(256) kern(float* out, int iters, unsigned seed) {
  u32x4 a4 = {seed + threadIdx.x, seed * 3u, seed * 5u + threadIdx.x, seed * 7u};
  u32x4 b4 = {seed * 11u + threadIdx.x, seed * 13u, seed * 17u, seed * 19u + threadIdx.x};
  a4 &= 0x3f803f80u; b4 &= 0x3f803f80u;
  bf16x8 a = __builtin_bit_cast(bf16x8, a4), b = __builtin_bit_cast(bf16x8, b4);
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 6 * NACC; ++u) {
      const int k = (MODE == 0) ? (u % NACC) : (u / 6);
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[k]) : "v"(a), "v"(b));
    }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NACC>
void run(const char* name, int blocks_per_cu, float* out) {
  const int iters = 2000, blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<MODE, NACC><<<blocks, 256>>>(out, 100, 12345u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<MODE, NACC><<<blocks, 256>>>(out, iters, 12345u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 6 * NACC * blocks_per_cu;
  printf("%-28s accs=%2d waves/SIMD=%d : %.2f ns per MFMA per SIMD\n", name, NACC, blocks_per_cu, ms * 1e6 / n);
}

int main() {
  float* out; hipMalloc(&out, 256 * 4 * 256 * 4);
  for (int w = 1; w <= 2; ++w) {
    run<0, 1>("one accumulator", w, out);
    run<0, 4>("round-robin", w, out);
    run<1, 4>("chains of 6", w, out);
    run<0, 16>("round-robin", w, out);
    run<1, 16>("chains of 6", w, out);
  }
  return 0;
}

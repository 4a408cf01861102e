// issue cost of the vector instructions the expansion kernels are made of: cycles per instruction on one SIMD with 1 or
// 2 waves, 8 independent registers in turn, alone and behind one v_smfmac_f32_16x16x64_f16 per 4 instructions
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TYPE, int MFMA>
__global__ void __launch_bounds__(256) kern(float* out, int iters, unsigned seed) {
  f16x8 a; f16x16 b;
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.001f * (threadIdx.x + i + seed));
  for (int i = 0; i < 16; ++i) b[i] = (_Float16)(0.002f * (threadIdx.x + 2 * i + seed));
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float x[8], y[8];
  unsigned w[8];
  for (int i = 0; i < 8; ++i) { x[i] = 1.0f + threadIdx.x * 0.001f + i; y[i] = 0.5f + i; w[i] = i; }
  const int idx = 0xCCCC + (threadIdx.x & 3);
  const unsigned long long smask = __builtin_amdgcn_ballot_w64((threadIdx.x & 1) != 0 || seed == 7);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MFMA) acc[u & 3] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a, b, acc[u & 3], idx, 0, 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = (u * 4 + k) & 7;
        if (TYPE == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[r]) : "v"(1.0000001f));
        if (TYPE == 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w[r]) : "v"(x[r]), "v"(y[r]));
        if (TYPE == 2) asm volatile("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "+v"(w[r]) : "v"(w[(r + 1) & 7]), "v"(x[r]));
        if (TYPE == 3) asm volatile("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(w[r]) : "v"(w[(r + 1) & 7]), "v"(x[r]));
        if (TYPE == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[r]) : "v"(y[r]));
        if (TYPE == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[r]) : "v"(1.0000001f), "v"(y[r]));
        if (TYPE == 6) asm volatile("v_exp_f32 %0, %1" : "=v"(y[r]) : "v"(x[r]));
        if (TYPE == 7) asm volatile("v_rcp_f32 %0, %1" : "=v"(y[r]) : "v"(x[r]));
        if (TYPE == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&x[r & 6])) : "v"(*reinterpret_cast<double*>(&y[r & 6])));
        if (TYPE == 9) asm volatile("v_cvt_pk_f16_f32 %0, %1, 0" : "=v"(w[r]) : "v"(x[r]));
        if (TYPE == 10) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(w[r]) : "v"(x[r]), "v"(y[r]));
        if (TYPE == 11) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(w[r]) : "v"(x[r]));
        if (TYPE == 13) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[r]) : "v"(y[r]), "s"(smask));
        if (TYPE == 14) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x[r]) : "v"(w[0]), "v"(y[r]));
        if (TYPE == 15) { if (k & 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[r]) : "v"(y[r]));
                          else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[r]) : "v"(1.0000001f)); }
        if (TYPE == 16) asm volatile("v_and_b32 %0, %0, %1" : "+v"(w[r]) : "v"(w[(r + 1) & 7]));
        if (TYPE == 17) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x[r]) : "v"(y[(r + 1) & 7]), "v"(y[r]) : "vcc");
        if (TYPE == 12) asm volatile("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(y[r]) : "v"(w[r]), "v"(x[r]));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i] + y[i] + (float)w[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int TYPE, int MFMA>
void run(const char* name, int blocks_per_cu, float* out) {
  const int iters = 4000, blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<TYPE, MFMA><<<blocks, 256>>>(out, 100, 12345u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<TYPE, MFMA><<<blocks, 256>>>(out, iters, 12345u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double groups = (double)iters * 16 * blocks_per_cu;      // (smfmac +) 4 instructions, per SIMD
  printf("%-22s smfmac=%d waves/SIMD=%d : %6.2f ns per group of 4 per SIMD -> %5.1f cycles at 2.0 GHz\n", name, MFMA,
         blocks_per_cu, ms * 1e6 / groups, ms * 1e6 / groups * 2.0);
}

#define BOTH(T, name) run<T, 0>(name, w, out); run<T, 1>(name, w, out);
int main() {
  float* out; hipMalloc(&out, 256 * 4 * 256 * 4);
  for (int w = 1; w <= 2; ++w) {
    BOTH(0, "v_mul_f32") BOTH(1, "v_cvt_pk_f16_f32") BOTH(9, "v_cvt_pk_f16_f32 x,0") BOTH(2, "v_fma_mixlo_f16") BOTH(3, "v_fma_mixhi_f16")
    BOTH(12, "v_fma_mix_f32") BOTH(10, "v_cvt_pkrtz_f16_f32") BOTH(11, "v_cvt_f16_f32")
    BOTH(4, "v_cndmask_b32 vcc") BOTH(13, "v_cndmask_b32 sgpr") BOTH(14, "v_bfi_b32") BOTH(15, "mul/cndmask alternating") BOTH(16, "v_and_b32") BOTH(17, "v_cmp + v_cndmask") BOTH(5, "v_fma_f32") BOTH(8, "v_pk_mul_f32") BOTH(6, "v_exp_f32") BOTH(7, "v_rcp_f32")
  }
  return 0;
}

// Microbenchmark: FP32 VALU issue rate on gfx950 for plain v_fma_f32 vs v_pk_fma_f32, by waves/SIMD.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int CHAINS>
__global__ void k_fma(float* out, int iters, float a, float b) {
  float x[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
__global__ void k_pkfma(float* out, int iters, float a, float b) {
  float2v x[CHAINS];
  float2v av = {a, a}, bv = {b, b};
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) x[i] = float2v{float(threadIdx.x + i), float(i)};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(bv));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
__global__ void k_minmax(float* out, int iters, float a, float b) {
  float x[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define DEF_KERNEL(NAME, ASM)                                                            \
  template <int CHAINS> __global__ void NAME(float* out, int iters, float a, float b) {   \
    float x[CHAINS];                                                                       \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x + i;             \
    for (int it = 0; it < iters; ++it) {                                                   \
      _Pragma("unroll") for (int r = 0; r < 8; ++r)                                        \
        _Pragma("unroll") for (int i = 0; i < CHAINS; ++i)                                 \
          asm volatile(ASM : "+v"(x[i]) : "v"(a), "v"(b) : "vcc");                          \
    }                                                                                      \
    float s = 0;                                                                           \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) s += x[i];                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                        \
  }
DEF_KERNEL(k_min, "v_min_f32 %0, %0, %1")
DEF_KERNEL(k_max, "v_max_f32 %0, %0, %2")
DEF_KERNEL(k_mul, "v_mul_f32 %0, %0, %1")
DEF_KERNEL(k_add, "v_add_f32 %0, %0, %1")
DEF_KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
DEF_KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2")
DEF_KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %1")
DEF_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF_KERNEL(k_and, "v_and_b32 %0, %0, %1")
DEF_KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL(k_sqrt, "v_sqrt_f32 %0, %0")
DEF_KERNEL(k_rsq, "v_rsq_f32 %0, %0")
DEF_KERNEL(k_fma_neg, "v_fma_f32 %0, -%0, %1, %2")
DEF_KERNEL(k_maxabs, "v_max_f32 %0, -|%0|, %1")
DEF_KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
DEF_KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
DEF_KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8")
DEF_KERNEL(k_sub, "v_sub_f32 %0, %0, %1")
DEF_KERNEL(k_cvt, "v_cvt_f32_u32 %0, %0")
DEF_KERNEL(k_sdwa, "v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
DEF_KERNEL(k_xad, "v_xad_u32 %0, %0, %1, %2")
DEF_KERNEL(k_lshr, "v_lshrrev_b32 %0, 13, %0")
DEF_KERNEL(k_fma_lit, "v_fma_f32 %0, 2.0, %1, %0")
DEF_KERNEL(k_fmac_lit, "v_fmac_f32 %0, 0x36800000, %1")

template <typename K>
double run(K kern, int waves_per_simd, int iters, float* d) {
  // 256 CUs x 4 SIMDs; block = 256 threads (4 waves = 1 per SIMD); blocks per CU = waves_per_simd
  dim3 grid(256 * waves_per_simd), block(256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<grid, block>>>(d, 10, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<grid, block>>>(d, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4 * 2);
  const int iters = 20000;
  printf("kernel chains waves/SIMD ms  wave-instr/SIMD/us  (2.4GHz: 1200 = one per 2 clk)\n");
#define RUN(NAME, K) for (int w : {1, 2, 4, 8}) { double ms = run(K<8>, w, iters, d); printf("%-14s 8  %d  %.3f  %.1f\n", NAME, w, ms, double(iters) * 8 * 8 * w / (ms * 1e3)); }
  RUN("v_min_f32", k_min) RUN("v_max_f32", k_max) RUN("v_mul_f32", k_mul) RUN("v_add_f32", k_add) RUN("v_fmac_f32", k_fmac)
  RUN("v_med3_f32", k_med3) RUN("v_cmp_lt", k_cmp) RUN("v_cndmask", k_cndmask) RUN("v_and_b32", k_and) RUN("v_mul_lo_u32", k_mullo)
  RUN("v_sqrt_f32", k_sqrt) RUN("v_rsq_f32", k_rsq) RUN("v_fma neg", k_fma_neg) RUN("v_max -|x|", k_maxabs) RUN("v_lshl_add", k_lshl_add) RUN("v_xor", k_xor)
  RUN("v_bitop3", k_bitop3) RUN("v_sub_f32", k_sub) RUN("v_cvt_f32_u32", k_cvt) RUN("v_xor sdwa", k_sdwa) RUN("v_xad_u32", k_xad) RUN("v_lshrrev", k_lshr) RUN("v_fma 2.0", k_fma_lit) RUN("v_fmac literal", k_fmac_lit)
  for (int w : {1, 2, 3, 4, 8}) {
    double ms = run(k_fma<8>, w, iters, d);
    printf("v_fma_f32     8  %d  %.3f  %.1f\n", w, ms, double(iters) * 8 * 8 * w / (ms * 1e3));
    ms = run(k_fma<2>, w, iters, d);
    printf("v_fma_f32     2  %d  %.3f  %.1f\n", w, ms, double(iters) * 8 * 2 * w / (ms * 1e3));
    ms = run(k_pkfma<8>, w, iters, d);
    printf("v_pk_fma_f32  8  %d  %.3f  %.1f\n", w, ms, double(iters) * 8 * 8 * w / (ms * 1e3));
    ms = run(k_pkfma<2>, w, iters, d);
    printf("v_pk_fma_f32  2  %d  %.3f  %.1f\n", w, ms, double(iters) * 8 * 2 * w / (ms * 1e3));
    ms = run(k_minmax<8>, w, iters, d);
    printf("v_max3_f32    8  %d  %.3f  %.1f\n", w, ms, double(iters) * 8 * 8 * w / (ms * 1e3));
  }
  return 0;
}

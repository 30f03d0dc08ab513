// Microbenchmark: v_cndmask_b32 issue rate on gfx950 (VOP2 with vcc, VOP3 with an SGPR pair mask,
// same-register and distinct-register forms), and v_cmp writing vcc vs an SGPR pair.
// hipcc --offload-arch=gfx950 -O3 -o cndmask cndmask.hip && ./cndmask
#include <hip/hip_runtime.h>
#include <cstdio>
#define DEF(NAME, ASM)                                                                    \
  __global__ void NAME(float* out, int iters, float a, float b) {                          \
    float x[8];                                                                             \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;                   \
    unsigned long long mask = (threadIdx.x & 1) ? 0x5555555555555555ull : 0x3333333333333333ull; \
    mask = __builtin_amdgcn_readfirstlane((unsigned)mask) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(mask >> 32)) << 32); \
    asm volatile("s_mov_b64 vcc, %0" ::"s"(mask) : "vcc");                                  \
    for (int it = 0; it < iters; ++it) {                                                    \
      _Pragma("unroll") for (int r = 0; r < 8; ++r)                                         \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                       \
          asm volatile(ASM : "+v"(x[i]) : "v"(a), "v"(b), "s"(mask) : "vcc", "s10", "s11");                       \
    }                                                                                       \
    float s = 0;                                                                            \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) s += x[i];                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                         \
  }
DEF(k_vop2_vcc, "v_cndmask_b32 %0, %0, %1, vcc")
DEF(k_vop2_vcc_ab, "v_cndmask_b32 %0, %1, %2, vcc")
DEF(k_vop3_sgpr, "v_cndmask_b32 %0, %0, %1, %3")
DEF(k_vop3_sgpr_ab, "v_cndmask_b32 %0, %1, %2, %3")
DEF(k_vop3_denorm, "v_cndmask_b32 %0, 0, %1, %3")
DEF(k_cmp_sgpr, "v_cmp_lt_f32 s[10:11], %0, %1")
DEF(k_cmp_vcc, "v_cmp_lt_f32 vcc, %0, %1")
DEF(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
DEF(k_sand_vcc, "s_and_b64 vcc, %3, exec\n v_cndmask_b32 %0, %0, %1, vcc")
DEF(k_sand_sgpr, "s_and_b64 s[10:11], %3, exec\n v_cndmask_b32 %0, %0, %1, s[10:11]")
DEF(k_sand_vcc2, "s_and_b64 vcc, %3, exec\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %2, vcc")
DEF(k_cmp_sand_vcc, "v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 vcc, vcc, %3\n v_cndmask_b32 %0, %0, %2, vcc")
DEF(k_cmp_sand_sgpr, "v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 s[10:11], vcc, %3\n v_cndmask_b32 %0, %0, %2, s[10:11]")
template <typename K>
double run(K kern, int w, int iters, float* d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  kern<<<dim3(256 * w), dim3(256)>>>(d, 10, 1.0001f, 0.5f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  kern<<<dim3(256 * w), dim3(256)>>>(d, iters, 1.0001f, 0.5f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4 * 2);
  const int iters = 5000;
  printf("kernel waves/SIMD ms wave-instr/SIMD/us\n");
#define RUN(NAME, K, PER) for (int w : {1, 4, 8}) { double ms = run(K, w, iters, d); printf("%-28s %d %.3f %.1f\n", NAME, w, ms, double(iters) * 64 * PER * w / (ms * 1e3)); }
  RUN("cndmask x,x,a vcc", k_vop2_vcc, 1) RUN("cndmask x=a,b vcc", k_vop2_vcc_ab, 1) RUN("cndmask x,x,a sgpr", k_vop3_sgpr, 1)
  RUN("cndmask x=a,b sgpr", k_vop3_sgpr_ab, 1) RUN("cndmask x=0,a sgpr", k_vop3_denorm, 1)
  RUN("v_cmp -> sgpr pair", k_cmp_sgpr, 1) RUN("v_cmp -> vcc", k_cmp_vcc, 1) RUN("v_cmp + cndmask (2 instr)", k_cmp_cnd, 2)
  RUN("s_and->vcc + cndmask vcc", k_sand_vcc, 1) RUN("s_and->sgpr + cndmask sgpr", k_sand_sgpr, 1)
  RUN("s_and->vcc + 2 cndmask vcc", k_sand_vcc2, 2) RUN("cmp, s_and->vcc, cndmask", k_cmp_sand_vcc, 2) RUN("cmp, s_and->sgpr, cndmask", k_cmp_sand_sgpr, 2)
  return 0;
}

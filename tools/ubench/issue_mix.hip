// Microbenchmark: do scalar-ALU instructions issue alongside vector-ALU ones on gfx950, or do they
// take VALU issue slots?  Loop bodies of V v_fma_f32 + S s_add_u32 (independent chains), by waves/SIMD.
// hipcc --offload-arch=gfx950 -O3 -o issue_mix issue_mix.hip && ./issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>

template <int V, int S>
__global__ void k_mix(float* out, int iters, float a, float b, int sa) {
  float x[8];
  int s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; s[i] = __builtin_amdgcn_readfirstlane(sa + i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      // one round: V vector + S scalar instructions, interleaved as evenly as the counts allow
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i < V) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        if (i < S) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s[i]) : "s"(sa) : "scc");
      }
    }
  }
  float acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += x[i] + float(s[i]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// s_load_dwordx16 issue rate (L1-resident 4 KiB table), with V vector instructions per load
template <int V>
__global__ void k_smem(float* out, const float* __restrict__ tab, int iters, float a, float b) {
  typedef float f16 __attribute__((ext_vector_type(16)));
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
  const f16 __attribute__((address_space(4)))* t = (const f16 __attribute__((address_space(4)))*)tab;
  float acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f16 v = t[(it * 4 + r) & 63];
#pragma unroll
      for (int i = 0; i < V; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(a), "v"(b));
      acc += v[0] + v[15];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename F>
double timeit(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch(20000);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4 * 2);
  float* tab; hipMalloc(&tab, 4096); hipMemset(tab, 0, 4096);
  const int iters = 20000;
  printf("body(V v_fma + S s_add per round, 8 rounds/iter)  waves/SIMD  ms  vector-instr/SIMD/us  scalar-instr/SIMD/us\n");
#define RUN(V, S) for (int w : {1, 2, 4, 8}) { \
    double ms = timeit([&](int n) { k_mix<V, S><<<dim3(256 * w), dim3(256)>>>(d, n, 1.0001f, 0.5f, 3); }); \
    printf("V=%d S=%d  %d  %.3f  %.1f  %.1f\n", V, S, w, ms, double(iters) * 8 * V * w / (ms * 1e3), double(iters) * 8 * S * w / (ms * 1e3)); }
  RUN(8, 0) RUN(0, 8) RUN(8, 8) RUN(8, 4) RUN(4, 8) RUN(8, 2)
  printf("s_load_dwordx16 + V v_fma per load: waves/SIMD ms loads/SIMD/us vector-instr/SIMD/us\n");
#define RUNS(V) for (int w : {1, 2, 4, 8}) { \
    double ms = timeit([&](int n) { k_smem<V><<<dim3(256 * w), dim3(256)>>>(d, tab, n, 1.0001f, 0.5f); }); \
    printf("V=%d  %d  %.3f  %.1f  %.1f\n", V, w, ms, double(iters) * 4 * w / (ms * 1e3), double(iters) * 4 * V * w / (ms * 1e3)); }
  RUNS(0) RUNS(8) RUNS(16) RUNS(32)
  return 0;
}

// Microbenchmark 2: FMA throughput with 3 distinct VGPR sources and the shadow-test instruction mix.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

// distinct source registers: x[i] = fma(y[i], z[j], x[i])
__global__ void k_fma3(float* out, int iters, float seed) {
  float x[8], y[8], z[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; y[i] = seed + i; z[i] = seed * i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(y[(i + r) & 7]), "v"(z[(i * 3 + r) & 7]));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// VOP2 fmac with distinct sources
__global__ void k_fmac3(float* out, int iters, float seed) {
  float x[8], y[8], z[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; y[i] = seed + i; z[i] = seed * i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(y[(i + r) & 7]), "v"(z[(i * 3 + r) & 7]));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the shadow test as the compiler emits it: 10 instructions per (ray, sphere), dependent chain per ray
__global__ void k_shadow(float* out, int iters, float seed) {
  float Lx[13], Ly[13], Lz[13], m[13];
#pragma unroll
  for (int k = 0; k < 13; ++k) { Lx[k] = seed + k; Ly[k] = seed * k; Lz[k] = seed - k; m[k] = -1.f; }
  float cx = threadIdx.x * 0.01f, cy = seed, cz = 1.f - seed, r2 = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 13; ++k) {
      float nb = fmaf(cz, Lz[k], fmaf(cy, Ly[k], cx * Lx[k]));
      float lx = fmaf(-nb, Lx[k], cx), ly = fmaf(-nb, Ly[k], cy), lz = fmaf(-nb, Lz[k], cz);
      float dl = fmaf(-lz, lz, fmaf(-ly, ly, fmaf(-lx, lx, r2)));
      m[k] = __uint_as_float(__float_as_uint(m[k]) & (__float_as_uint(dl) | __float_as_uint(nb)));
      asm volatile("" : "+v"(m[k]));
    }
    asm volatile("" : "+v"(cx), "+v"(cy), "+v"(cz));
  }
  float s = 0;
#pragma unroll
  for (int k = 0; k < 13; ++k) s += m[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
double run(K kern, int waves_per_simd, int iters, float* d) {
  dim3 grid(256 * waves_per_simd), block(256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  kern<<<grid, block>>>(d, 10, 1.0001f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  kern<<<grid, block>>>(d, iters, 1.0001f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* d; CHECK(hipMalloc(&d, 256 * 8 * 256 * 4 * 2));
  printf("kernel waves/SIMD ms  wave-instr/SIMD/us\n");
  for (int w : {1, 2, 3, 4, 5, 8}) {
    double ms = run(k_fma3, w, 20000, d);
    printf("v_fma_f32 3 distinct srcs   %d  %.3f  %.1f\n", w, ms, 20000.0 * 64 * w / (ms * 1e3));
    ms = run(k_fmac3, w, 20000, d);
    printf("v_fmac_f32 distinct srcs    %d  %.3f  %.1f\n", w, ms, 20000.0 * 64 * w / (ms * 1e3));
    ms = run(k_shadow, w, 20000, d);
    printf("shadow test (10 instr/ray)  %d  %.3f  %.1f  (%.2f Gtests/s/SIMD-us...)\n", w, ms, 20000.0 * 130 * w / (ms * 1e3), 0.0);
  }
  return 0;
}

// Sustained matrix-core rate of one MI355X for the two instructions the hot kernels use, nothing but MFMAs in the loop
// (operands stay in registers): what the chip holds under its power limit, to read roofline fractions against.
//   hipcc --offload-arch=gfx950 -O3 -o ab_tmp/mfma_peak scripts/micro/mfma_peak.hip && ./ab_tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#ifndef RANDOM_DATA
#define RANDOM_DATA 1  // 0: near-constant operands (little switching power)
#endif
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS>
__global__ __launch_bounds__(512) void k_i8(int iters, int* out) {
  unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u + 12345u);
  auto nx = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (int)(RANDOM_DATA ? h : (h & 3)); };
  i32x4 a = {nx(), nx(), nx(), nx()}, b = {nx(), nx(), nx(), nx()};
  i32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[c], 0, 0, 0);
  }
  int s = 0;
  for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  if (s == 0x7fffffff) out[0] = s;
}
template <int CHAINS>
__global__ __launch_bounds__(512) void k_bf16(int iters, float* out) {
  bf16x8 a, b;
  unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u + 12345u);
  for (int r = 0; r < 8; ++r) { h ^= h << 13; h ^= h >> 17; h ^= h << 5; a[r] = (__bf16)(RANDOM_DATA ? (float)(int)(h & 0xffff) / 65536.f - 0.5f : 1.0f); b[r] = (__bf16)(RANDOM_DATA ? (float)(int)(h >> 16) / 65536.f - 0.5f : 1.0f); }
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
  }
  float s = 0;
  for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  if (s == 1.2345f) out[0] = s;
}

template <class F> double run(F launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e-3;
}

int main() {
  int* out;
  hipMalloc(&out, 64);
  const int CUS = 256, iters = 4096;
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int blocks = CUS * waves_per_simd / 2;  // 512 threads = 8 waves = 2 per SIMD; one or two blocks ... per CU
    (void)blocks;
  }
  for (int bpc = 1; bpc <= 2; ++bpc) {  // blocks of 8 waves per CU: 2 or 4 waves per SIMD
    const int grid = CUS * bpc;
    for (int ms_run = 0; ms_run < 2; ++ms_run) {  // short and long launches (the clock settles under load)
      const int it = ms_run ? iters * 8 : iters;
      double t = run([&] { hipLaunchKernelGGL(k_i8<4>, dim3(grid), dim3(512), 0, 0, it, out); }, 5);
      double ops = 2.0 * 32 * 32 * 32 * 4.0 * it * 8.0 * grid;
      printf("int8  32x32x32  %d waves/SIMD  %7.1f us per launch: %6.0f TOP/s\n", 2 * bpc, t * 1e6, ops / t / 1e12);
      t = run([&] { hipLaunchKernelGGL(k_bf16<4>, dim3(grid), dim3(512), 0, 0, it, (float*)out); }, 5);
      ops = 2.0 * 32 * 32 * 16 * 4.0 * it * 8.0 * grid;
      printf("bf16  32x32x16  %d waves/SIMD  %7.1f us per launch: %6.0f TFLOP/s\n", 2 * bpc, t * 1e6, ops / t / 1e12);
    }
  }
  return 0;
}

// A host WITHOUT Python at run time: read one layer of a packed-weights file (cube_packfile_*), copy it to device memory
// this program owns, run it through the C ABI (cube_gemm_bf16, bf16x3 operands, tanh epilogue) on a deterministic input,
// write the output planes to a file.  tests/test_gpu_c_host.py runs the same layer on the same input through the Python
// host and compares the bytes.  Built by __graft_entry__.build():
//   hipcc --offload-arch=gfx950 examples/c_host.cpp -o nlp-cube_amd/cube_amd/lib/c_host_example -L<lib> -lcubehip
// usage: c_host_example <file.cubepack> <prefix, e.g. char_pre_out> <rows M> <out file>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../include/cube_hip.h"

#define HIP_OK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
  } while (0)
#define CUBE_DO(x)                                                                 \
  do {                                                                             \
    if ((x) != CUBE_OK) { fprintf(stderr, "%s: %s\n", #x, cube_last_error()); return 3; } \
  } while (0)

static uint16_t bf16_rne(float v) {  // round-to-nearest-even, as torch's .to(bfloat16)
  uint32_t u;
  memcpy(&u, &v, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf16_f32(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float v;
  memcpy(&v, &u, 4);
  return v;
}

int main(int argc, char** argv) {
  if (argc != 5) { fprintf(stderr, "usage: %s file.cubepack prefix rows out\n", argv[0]); return 1; }
  const std::string pre = argv[2];
  const int M = atoi(argv[3]);
  cube_packfile* f = nullptr;
  CUBE_DO(cube_packfile_open(argv[1], &f));
  cube_pack_entry w, wlo, bias;
  long long K = 0, N = 0;
  CUBE_DO(cube_packfile_find(f, (pre + ".W.hi").c_str(), &w));
  CUBE_DO(cube_packfile_find(f, (pre + ".W.lo").c_str(), &wlo));
  CUBE_DO(cube_packfile_find(f, (pre + ".bias").c_str(), &bias));
  CUBE_DO(cube_packfile_int(f, (pre + ".K").c_str(), &K));
  CUBE_DO(cube_packfile_int(f, (pre + ".N").c_str(), &N));
  if (w.dtype != CUBE_PACK_BF16 || w.ndim != 2 || w.dims[0] != N || w.dims[1] != K || bias.dtype != CUBE_PACK_F32) {
    fprintf(stderr, "unexpected entry shapes\n");
    return 4;
  }
  // deterministic input in [-1, 1): x[r, c] = ((r * 131 + c * 31 + 7) mod 2001) / 1000 - 1, as hi/lo bf16 planes
  std::vector<uint16_t> xh((size_t)M * K), xl((size_t)M * K);
  for (int r = 0; r < M; ++r)
    for (int c = 0; c < K; ++c) {
      const float v = (float)((r * 131 + c * 31 + 7) % 2001) / 1000.0f - 1.0f;
      const uint16_t h = bf16_rne(v);
      xh[(size_t)r * K + c] = h;
      xl[(size_t)r * K + c] = bf16_rne(v - bf16_f32(h));
    }
  HIP_OK(hipSetDevice(0));
  CUBE_DO(cube_init());
  void *dW, *dWlo, *dB, *dX, *dXlo, *dO, *dOlo;
  const size_t xbytes = (size_t)M * K * 2, obytes = (size_t)M * N * 2;
  HIP_OK(hipMalloc(&dW, w.nbytes));
  HIP_OK(hipMalloc(&dWlo, wlo.nbytes));
  HIP_OK(hipMalloc(&dB, bias.nbytes));
  HIP_OK(hipMalloc(&dX, xbytes));
  HIP_OK(hipMalloc(&dXlo, xbytes));
  HIP_OK(hipMalloc(&dO, obytes));
  HIP_OK(hipMalloc(&dOlo, obytes));
  HIP_OK(hipMemcpy(dW, w.data, w.nbytes, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dWlo, wlo.data, wlo.nbytes, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dB, bias.data, bias.nbytes, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dX, xh.data(), xbytes, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dXlo, xl.data(), xbytes, hipMemcpyHostToDevice));
  HIP_OK(hipMemset(dO, 0, obytes));
  HIP_OK(hipMemset(dOlo, 0, obytes));
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  cube_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.A = dX; a.A_lo = dXlo; a.lda = (int)K; a.M = M; a.taps = 1; a.Kc = (int)K;
  a.W = dW; a.W_lo = dWlo; a.N = (int)N; a.bias = (const float*)dB;
  a.epilogue = CUBE_EPI_BF16_TANH; a.n_valid = (int)N;
  a.out0 = dO; a.out0_lo = dOlo; a.ldo0 = (int)N;
  CUBE_DO(cube_gemm_bf16(&a, st));
  HIP_OK(hipStreamSynchronize(st));
  std::vector<uint16_t> oh((size_t)M * N), ol((size_t)M * N);
  HIP_OK(hipMemcpy(oh.data(), dO, obytes, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(ol.data(), dOlo, obytes, hipMemcpyDeviceToHost));
  FILE* o = fopen(argv[4], "wb");
  if (!o) { fprintf(stderr, "cannot write %s\n", argv[4]); return 5; }
  fwrite(oh.data(), 2, oh.size(), o);
  fwrite(ol.data(), 2, ol.size(), o);
  fclose(o);
  printf("%s: %d x %lld -> %lld, out[0,0] = %g\n", pre.c_str(), M, K, N, bf16_f32(oh[0]) + bf16_f32(ol[0]));
  cube_packfile_close(f);
  for (void* p : {dW, dWlo, dB, dX, dXlo, dO, dOlo}) (void)hipFree(p);
  (void)hipStreamDestroy(st);
  return 0;
}

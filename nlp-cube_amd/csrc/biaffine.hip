// Biaffine arc scoring as a batched MFMA contraction over padded sentence tiles + row softmax.
//
// logits[b, i, j] = u[b, i+1, :] . h2[b, j, :]     (u = [h1;1] W from cube_gemm_bf16)
// One wave per (sentence, 32 dependents): for each 32-head column tile the wave runs D/16
// v_mfma_f32_32x32x16_bf16 with both operands read straight from HBM/L2 (each element is used
// once per wave, so LDS staging would only add a round trip).  The softmax runs over all T
// columns, padded positions included, exactly like the reference (parser.py:425).
#include "common.h"
#include "../../include/cube_hip.h"

namespace {

__device__ __forceinline__ float half_max(float v) {  // max over the 32 lanes of a half-wave
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <bool SPLIT>
__global__ __launch_bounds__(64) void biaffine_arc_kernel(const uint16_t* __restrict__ u, const uint16_t* __restrict__ u_lo,
                                                          int ld_u, const uint16_t* __restrict__ h2,
                                                          const uint16_t* __restrict__ h2_lo, int ld_h2, int col_h2,
                                                          int D, int B, int T, float* logits, float* probs) {
  const int L = T - 1;
  const int b = blockIdx.y;
  const int i0 = blockIdx.x * 32;
  const int lane = threadIdx.x;
  const int l31 = lane & 31, lh = lane >> 5;
  const int ntile = (T + 31) >> 5;
  const int ia = min(i0 + l31, L - 1);  // clamp: rows beyond L are computed on a valid row and dropped
  const size_t aoff = ((size_t)b * T + 1 + ia) * ld_u + lh * 8;
  const uint16_t* ap = u + aoff;
  float rmax[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) rmax[r] = -INFINITY;
  float* lg = logits + (size_t)b * L * T;

  for (int jt = 0; jt < ntile; ++jt) {
    const int j = jt * 32 + l31;
    const int jb = min(j, T - 1);
    const size_t boff = ((size_t)b * T + jb) * ld_h2 + col_h2 + lh * 8;
    const uint16_t* bp = h2 + boff;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = 0; k < D; k += 16) {
      const bf16x8 fa = *reinterpret_cast<const bf16x8*>(ap + k);
      const bf16x8 fb = *reinterpret_cast<const bf16x8*>(bp + k);
      if constexpr (SPLIT) {
        const bf16x8 fal = *reinterpret_cast<const bf16x8*>(u_lo + aoff + k);
        const bf16x8 fbl = *reinterpret_cast<const bf16x8*>(h2_lo + boff + k);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal, fb, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fbl, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float v = (j < T) ? acc[r] : -INFINITY;
      if (i < L && j < T) lg[(size_t)i * T + j] = v;
      rmax[r] = fmaxf(rmax[r], v);
    }
  }
  float rsum[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    rmax[r] = half_max(rmax[r]);
    rsum[r] = 0.f;
  }
  float* pr = probs + (size_t)b * L * T;
  // each lane re-reads the logits it wrote itself (same thread: program order makes them visible)
  for (int jt = 0; jt < ntile; ++jt) {
    const int j = jt * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (i < L && j < T) {
        const float e = expf(lg[(size_t)i * T + j] - rmax[r]);
        pr[(size_t)i * T + j] = e;
        rsum[r] += e;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) rsum[r] = half_sum(rsum[r]);
  for (int jt = 0; jt < ntile; ++jt) {
    const int j = jt * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (i < L && j < T) pr[(size_t)i * T + j] = pr[(size_t)i * T + j] / rsum[r];
    }
  }
}

}  // namespace

extern "C" int cube_biaffine_arc(const void* u, const void* u_lo, int ld_u, const void* h2, const void* h2_lo,
                                 int ld_h2, int col_h2, int D, int B, int T, float* logits, float* probs,
                                 void* stream) {
  CUBE_ENTRY();
  if (B <= 0 || T <= 1) return CUBE_OK;
  if (!u || !h2 || !logits || !probs) CUBE_FAIL(CUBE_ERR_ARG, "cube_biaffine_arc: null pointer");
  if (D <= 0 || D % 16 || D > ld_u || col_h2 < 0 || col_h2 + D > ld_h2) CUBE_FAIL(CUBE_ERR_ARG, "cube_biaffine_arc: bad D/columns");
  if ((u_lo == nullptr) != (h2_lo == nullptr)) CUBE_FAIL(CUBE_ERR_ARG, "cube_biaffine_arc: u_lo and h2_lo go together");
  if (ld_u % 8 || ld_h2 % 8 || col_h2 % 8 || (uintptr_t)u % 16 || (uintptr_t)h2 % 16 || (uintptr_t)u_lo % 16 ||
      (uintptr_t)h2_lo % 16)
    CUBE_FAIL(CUBE_ERR_ALIGN, "cube_biaffine_arc: operands must be 16-byte aligned");
  dim3 grid((T - 1 + 31) / 32, B), block(64);
  if (u_lo)
    hipLaunchKernelGGL(biaffine_arc_kernel<true>, grid, block, 0, (hipStream_t)stream, (const uint16_t*)u,
                       (const uint16_t*)u_lo, ld_u, (const uint16_t*)h2, (const uint16_t*)h2_lo, ld_h2, col_h2, D, B, T,
                       logits, probs);
  else
    hipLaunchKernelGGL(biaffine_arc_kernel<false>, grid, block, 0, (hipStream_t)stream, (const uint16_t*)u,
                       (const uint16_t*)nullptr, ld_u, (const uint16_t*)h2, (const uint16_t*)nullptr, ld_h2, col_h2, D,
                       B, T, logits, probs);
  CUBE_CHECK_LAUNCH("cube_biaffine_arc");
  return CUBE_OK;
}

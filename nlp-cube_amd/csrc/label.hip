// Label scoring for the decoded heads: logits[d, o] = x1[d]^T W[o] x2[head(d)] + bias[o].
//
// One wave owns 32 dependents and a group of labels.  The contraction is run transposed,
// V^T[q, d] = sum_p W[o][q][p] * x1[d][p]  (v_mfma_f32_32x32x16_bf16: A = 32 rows q of W[o],
// B = x1 of the 32 dependents), because then every accumulator register of a lane belongs to
// the lane's own dependent: the second contraction with x2 is a register-local FMA chain plus
// one half-wave exchange, and the [B*L, n_labels*D] intermediate (343 MB at B=256, L=32) is
// never materialised.  x1 fragments and the gathered x2 row stay in registers across labels;
// only W streams (L2-resident, 5 MB).
#include "common.h"
#include "../../include/cube_hip.h"

namespace {

template <int QT>  // D = 32*QT
__global__ __launch_bounds__(256) void label_kernel(const uint16_t* __restrict__ hl, int ld_hl, int col_x2,
                                                    const uint16_t* __restrict__ wl, const float* __restrict__ bias,
                                                    int n_labels, int labels_per_group,
                                                    const int32_t* __restrict__ heads, const int32_t* __restrict__ lens,
                                                    int B, int L, float* __restrict__ logits) {
  constexpr int D = 32 * QT;
  constexpr int KS = D / 16;
  const int T = L + 1;
  const int M = B * L;
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int d0 = (blockIdx.x * 4 + wid) * 32;
  if (d0 >= M) return;
  const int d = min(d0 + l31, M - 1);
  const int b = d / L, i = d - b * L;
  const int head = (i < lens[b]) ? heads[(size_t)b * L + i] : 0;
  const uint16_t* x1p = hl + ((size_t)b * T + 1 + i) * ld_hl + lh * 8;
  const uint16_t* x2p = hl + ((size_t)b * T + head) * ld_hl + col_x2 + lh * 4;

  bf16x8 fx1[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) fx1[k] = *reinterpret_cast<const bf16x8*>(x1p + k * 16);
  // x2 of this lane's dependent at q = 32*ct + 8*rq + 4*lh + (0..3): the accumulator rows it owns
  uint2 fx2[QT][4];
#pragma unroll
  for (int ct = 0; ct < QT; ++ct)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) fx2[ct][rq] = *reinterpret_cast<const uint2*>(x2p + ct * 32 + rq * 8);

  const int o_begin = blockIdx.y * labels_per_group;
  const int o_end = min(o_begin + labels_per_group, n_labels);
  for (int o = o_begin; o < o_end; ++o) {
    const uint16_t* wo = wl + (size_t)o * D * D + (size_t)l31 * D + lh * 8;
    float s = 0.f;
#pragma unroll
    for (int ct = 0; ct < QT; ++ct) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const uint16_t* wq = wo + (size_t)ct * 32 * D;
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const bf16x8 fw = *reinterpret_cast<const bf16x8*>(wq + k * 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fx1[k], acc, 0, 0, 0);
      }
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const uint2 v = fx2[ct][rq];
        s += acc[rq * 4 + 0] * __uint_as_float(v.x << 16);
        s += acc[rq * 4 + 1] * __uint_as_float(v.x & 0xffff0000u);
        s += acc[rq * 4 + 2] * __uint_as_float(v.y << 16);
        s += acc[rq * 4 + 3] * __uint_as_float(v.y & 0xffff0000u);
      }
    }
    s += __shfl_xor(s, 32);
    if (lh == 0 && d0 + l31 < M) logits[(size_t)(d0 + l31) * n_labels + o] = s + bias[o];
  }
}

}  // namespace

extern "C" int cube_label_score(const void* hl, int ld_hl, int col_x2, int D, const void* wl, const float* bias,
                                int n_labels, const int32_t* heads, const int32_t* lens, int B, int L, float* logits,
                                int32_t* labels, void* stream) {
  if (B <= 0 || L <= 0 || n_labels <= 0) return CUBE_OK;
  if (!hl || !wl || !bias || !heads || !lens || !logits) CUBE_FAIL(CUBE_ERR_ARG, "cube_label_score: null pointer");
  if (D <= 0 || D % 32 || D > 256 || D > col_x2 || col_x2 + D > ld_hl) CUBE_FAIL(CUBE_ERR_ARG, "cube_label_score: bad D/columns");
  if (ld_hl % 8 || col_x2 % 8 || (uintptr_t)hl % 16 || (uintptr_t)wl % 16)
    CUBE_FAIL(CUBE_ERR_ALIGN, "cube_label_score: operands must be 16-byte aligned");
  const int M = B * L;
  const int row_blocks = (M + 127) / 128;
  // enough label groups to give every SIMD of the chip a wave
  int groups = (1024 + row_blocks * 4 - 1) / (row_blocks * 4);
  if (groups < 1) groups = 1;
  if (groups > n_labels) groups = n_labels;
  const int per = (n_labels + groups - 1) / groups;
  groups = (n_labels + per - 1) / per;
  dim3 grid(row_blocks, groups), block(256);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH(Q)                                                                                                   \
  case Q:                                                                                                           \
    hipLaunchKernelGGL(label_kernel<Q>, grid, block, 0, st, (const uint16_t*)hl, ld_hl, col_x2, (const uint16_t*)wl, \
                       bias, n_labels, per, heads, lens, B, L, logits);                                             \
    break;
  switch (D / 32) {
    LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7) LAUNCH(8)
    default: CUBE_FAIL(CUBE_ERR_ARG, "cube_label_score: D=%d", D);
  }
#undef LAUNCH
  CUBE_CHECK_LAUNCH("cube_label_score");
  if (labels) return cube_argmax_rows(logits, n_labels, 0, n_labels, M, labels, stream);
  return CUBE_OK;
}

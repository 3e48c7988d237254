// Label scoring for the decoded heads: logits[d, o] = x1[d]^T W[o] x2[head(d)] + bias[o].
//
// One wave owns 32 dependents and a group of labels.  The contraction is run transposed,
// V^T[q, d] = sum_p W[o][q][p] * x1[d][p]  (v_mfma_f32_32x32x16_bf16: A = 32 rows q of W[o],
// B = x1 of the 32 dependents), because then every accumulator register of a lane belongs to
// the lane's own dependent: the second contraction with x2 is a register-local FMA chain plus
// one half-wave exchange, and the [B*L, n_labels*D] intermediate (343 MB at B=256, L=32) is
// never materialised.  x1 fragments stay in registers across labels, the gathered x2 rows sit
// in LDS as fp32 in accumulator order (16-byte conflict-free reads); only W streams
// (L2-resident).  SPLIT ("bf16x3"): x1 and W are hi/lo bf16 pairs (3 MFMA passes), x2 = hi + lo.
#include "common.h"
#include "../../include/cube_hip.h"

namespace {

template <int QT, bool SPLIT>  // D = 32*QT
__global__ __launch_bounds__(256) void label_kernel(const uint16_t* __restrict__ hl, const uint16_t* __restrict__ hl_lo,
                                                    int ld_hl, int col_x2, const uint16_t* __restrict__ wl,
                                                    const uint16_t* __restrict__ wl_lo, const float* __restrict__ bias,
                                                    int n_labels, int labels_per_group,
                                                    const int32_t* __restrict__ heads, const int32_t* __restrict__ lens,
                                                    int B, int L, float* __restrict__ logits) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int D = 32 * QT;
  constexpr int KS = D / 16;
  constexpr int X2_WAVE = QT * 4 * 2 * 32 * 16;  // bytes of one wave's x2 image
  const int T = L + 1;
  const int M = B * L;
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int d0 = (blockIdx.x * 4 + wid) * 32;
  if (d0 >= M) return;  // whole wave; no block-level barrier is used below
  const int d = min(d0 + l31, M - 1);
  const int b = d / L, i = d - b * L;
  const int head = (i < lens[b]) ? heads[(size_t)b * L + i] : 0;
  const size_t x1off = ((size_t)b * T + 1 + i) * ld_hl + lh * 8;
  const size_t x2off = ((size_t)b * T + head) * ld_hl + col_x2 + lh * 4;

  bf16x8 fx1[SPLIT ? 2 : 1][KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    fx1[0][k] = *reinterpret_cast<const bf16x8*>(hl + x1off + k * 16);
    if constexpr (SPLIT) fx1[1][k] = *reinterpret_cast<const bf16x8*>(hl_lo + x1off + k * 16);
  }
  // x2 of this lane's dependent at q = 32*ct + 8*rq + 4*lh + (0..3) -> LDS [ct][rq][lh][l31] float4
  float4* x2s = reinterpret_cast<float4*>(smem + wid * X2_WAVE);
#pragma unroll
  for (int ct = 0; ct < QT; ++ct)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const uint2 v = *reinterpret_cast<const uint2*>(hl + x2off + ct * 32 + rq * 8);
      float4 f = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                             __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
      if constexpr (SPLIT) {
        const uint2 w = *reinterpret_cast<const uint2*>(hl_lo + x2off + ct * 32 + rq * 8);
        f.x += __uint_as_float(w.x << 16);
        f.y += __uint_as_float(w.x & 0xffff0000u);
        f.z += __uint_as_float(w.y << 16);
        f.w += __uint_as_float(w.y & 0xffff0000u);
      }
      x2s[((ct * 4 + rq) * 2 + lh) * 32 + l31] = f;
    }
  // each lane reads back only what it wrote itself: no barrier needed

  const int o_begin = blockIdx.y * labels_per_group;
  const int o_end = min(o_begin + labels_per_group, n_labels);
  for (int o = o_begin; o < o_end; ++o) {
    const size_t woff = (size_t)o * D * D + (size_t)l31 * D + lh * 8;
    float s = 0.f;
#pragma unroll
    for (int ct = 0; ct < QT; ++ct) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const size_t wq = woff + (size_t)ct * 32 * D;
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const bf16x8 fw = *reinterpret_cast<const bf16x8*>(wl + wq + k * 16);
        if constexpr (SPLIT) {
          const bf16x8 fwl = *reinterpret_cast<const bf16x8*>(wl_lo + wq + k * 16);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwl, fx1[0][k], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fx1[1][k], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fx1[0][k], acc, 0, 0, 0);
      }
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const float4 f = x2s[((ct * 4 + rq) * 2 + lh) * 32 + l31];
        s += acc[rq * 4 + 0] * f.x;
        s += acc[rq * 4 + 1] * f.y;
        s += acc[rq * 4 + 2] * f.z;
        s += acc[rq * 4 + 3] * f.w;
      }
    }
    s += __shfl_xor(s, 32);
    if (lh == 0 && d0 + l31 < M) logits[(size_t)(d0 + l31) * n_labels + o] = s + bias[o];
  }
}

template <int QT>
int launch_label(bool split, dim3 grid, hipStream_t st, const uint16_t* hl, const uint16_t* hl_lo, int ld_hl, int col_x2,
                 const uint16_t* wl, const uint16_t* wl_lo, const float* bias, int n_labels, int per,
                 const int32_t* heads, const int32_t* lens, int B, int L, float* logits) {
  const int lds = 4 * QT * 4 * 2 * 32 * 16;
  if (split) {
    static bool set = false;
    if (!set && lds > 65536) {
      if (hipFuncSetAttribute((const void*)label_kernel<QT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        CUBE_FAIL(CUBE_ERR_LAUNCH, "cube_label_score: cannot reserve %d bytes of LDS", lds);
      set = true;
    }
    hipLaunchKernelGGL((label_kernel<QT, true>), grid, dim3(256), lds, st, hl, hl_lo, ld_hl, col_x2, wl, wl_lo, bias,
                       n_labels, per, heads, lens, B, L, logits);
  } else {
    static bool set = false;
    if (!set && lds > 65536) {
      if (hipFuncSetAttribute((const void*)label_kernel<QT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        CUBE_FAIL(CUBE_ERR_LAUNCH, "cube_label_score: cannot reserve %d bytes of LDS", lds);
      set = true;
    }
    hipLaunchKernelGGL((label_kernel<QT, false>), grid, dim3(256), lds, st, hl, hl_lo, ld_hl, col_x2, wl, wl_lo, bias,
                       n_labels, per, heads, lens, B, L, logits);
  }
  return CUBE_OK;
}

}  // namespace

extern "C" int cube_label_score(const void* hl, const void* hl_lo, int ld_hl, int col_x2, int D, const void* wl,
                                const void* wl_lo, const float* bias, int n_labels, const int32_t* heads,
                                const int32_t* lens, int B, int L, float* logits, int32_t* labels, void* stream) {
  if (B <= 0 || L <= 0 || n_labels <= 0) return CUBE_OK;
  if (!hl || !wl || !bias || !heads || !lens || !logits) CUBE_FAIL(CUBE_ERR_ARG, "cube_label_score: null pointer");
  if ((hl_lo == nullptr) != (wl_lo == nullptr)) CUBE_FAIL(CUBE_ERR_ARG, "cube_label_score: hl_lo and wl_lo go together");
  if (D <= 0 || D % 32 || D > 256 || D > col_x2 || col_x2 + D > ld_hl) CUBE_FAIL(CUBE_ERR_ARG, "cube_label_score: bad D/columns");
  if (ld_hl % 8 || col_x2 % 8 || (uintptr_t)hl % 16 || (uintptr_t)wl % 16 || (uintptr_t)hl_lo % 16 || (uintptr_t)wl_lo % 16)
    CUBE_FAIL(CUBE_ERR_ALIGN, "cube_label_score: operands must be 16-byte aligned");
  const int M = B * L;
  const int row_blocks = (M + 127) / 128;
  // enough label groups to give every SIMD of the chip a wave
  int groups = (1024 + row_blocks * 4 - 1) / (row_blocks * 4);
  if (groups < 1) groups = 1;
  if (groups > n_labels) groups = n_labels;
  const int per = (n_labels + groups - 1) / groups;
  groups = (n_labels + per - 1) / per;
  dim3 grid(row_blocks, groups);
  hipStream_t st = (hipStream_t)stream;
  int rc = CUBE_OK;
#define LAUNCH(Q)                                                                                                      \
  case Q:                                                                                                              \
    rc = launch_label<Q>(hl_lo != nullptr, grid, st, (const uint16_t*)hl, (const uint16_t*)hl_lo, ld_hl, col_x2,      \
                         (const uint16_t*)wl, (const uint16_t*)wl_lo, bias, n_labels, per, heads, lens, B, L, logits); \
    break;
  switch (D / 32) {
    LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7) LAUNCH(8)
    default: CUBE_FAIL(CUBE_ERR_ARG, "cube_label_score: D=%d", D);
  }
#undef LAUNCH
  if (rc != CUBE_OK) return rc;
  CUBE_CHECK_LAUNCH("cube_label_score");
  if (labels) return cube_argmax_rows(logits, n_labels, 0, n_labels, M, labels, stream);
  return CUBE_OK;
}

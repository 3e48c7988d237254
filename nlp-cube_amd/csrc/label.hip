// Label scoring for the decoded heads: logits[d, o] = x1[d]^T W[o] x2[head(d)] + bias[o].
//
// One workgroup = 4 waves = 128 dependents and a group of labels; one wave owns 32 dependents.
// The contraction is run transposed, V^T[q, d] = sum_p W[o][q][p] * x1[d][p]
// (v_mfma_f32_32x32x16_bf16: A = 32 rows q of W[o], B = x1 of the 32 dependents), because then
// every accumulator register of a lane belongs to the lane's own dependent: the second
// contraction with x2 is a register-local FMA chain plus one half-wave exchange, and the
// [B*L, n_labels*D] intermediate (343 MB at B=256, L=32) is never materialised.
// x1 fragments stay in registers across labels; the gathered x2 rows sit in LDS (bf16 planes in
// accumulator order).  W[o] streams through LDS in 32-row chunks by LDS-DMA, double-buffered and
// SHARED by the four waves (each wave streaming its own copy from L2 was L2-bandwidth bound);
// chunk rows are padded to an odd number of 16-byte slots so the ds_read_b128 fragment reads are
// conflict-free.  SPLIT ("bf16x3"): x1 and W are hi/lo bf16 pairs (3 MFMA passes), x2 = hi + lo.
#include "common.h"
#include "../../include/cube_hip.h"

namespace {

__device__ __forceinline__ void glds16(const uint16_t* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int QT, bool SPLIT> struct LabelCfg {
  static constexpr int D = 32 * QT, KS = D / 16, NPL = SPLIT ? 2 : 1;
  static constexpr int WROW = D * 2 + 16;                                 // padded chunk row (bytes), odd # of 16-B slots
  static constexpr int WPLANE = (32 * WROW + 1023) / 1024 * 1024;          // one plane of one 32-row chunk, whole DMA pieces
  static constexpr int WBUF = NPL * WPLANE;
  static constexpr int PIECES = WBUF / 1024, PW = (PIECES + 3) / 4;        // 1-KiB pieces per chunk / per wave
  static constexpr int LDS = 2 * WBUF;
};

template <int QT, bool SPLIT>
__global__ __launch_bounds__(256) void label_kernel(const uint16_t* __restrict__ hl, const uint16_t* __restrict__ hl_lo,
                                                    int ld_hl, int col_x2, const uint16_t* __restrict__ wl,
                                                    const uint16_t* __restrict__ wl_lo, const float* __restrict__ bias,
                                                    int n_labels, int labels_per_group,
                                                    const int32_t* __restrict__ heads, const int32_t* __restrict__ lens,
                                                    int B, int L, float* __restrict__ logits) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using C = LabelCfg<QT, SPLIT>;
  constexpr int D = C::D, KS = C::KS;
  const int T = L + 1;
  const int M = B * L;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int d0 = (blockIdx.x * 4 + wid) * 32;  // may be >= M for the last waves: they compute on a clamped row, store nothing
  const int d = min(d0 + l31, M - 1);
  const int b = d / L, i = d - b * L;
  const int head = (i < lens[b]) ? heads[(size_t)b * L + i] : 0;
  const size_t x1off = ((size_t)b * T + 1 + i) * ld_hl + lh * 8;
  const size_t x2off = ((size_t)b * T + head) * ld_hl + col_x2 + lh * 4;

  // ---- W chunk DMA: this lane's source offset (elements, inside one 32-row chunk of one plane)
  // for each piece its wave issues.  LDS byte o of a plane = row o / WROW, 16-byte slot (o % WROW) / 16.
  int w_src[C::PW];
#pragma unroll
  for (int k = 0; k < C::PW; ++k) {
    const int piece = wid + 4 * k;
    const int o = (piece % (C::WPLANE / 1024)) * 1024 + lane * 16;
    const int row = o / C::WROW, slot = (o % C::WROW) >> 4;
    w_src[k] = (row < 32 && slot < D / 8) ? row * D + slot * 8 : 0;  // padding slots re-read element 0 (never used)
  }
  auto issue_chunk = [&](int o, int ct, int buf) {
    const size_t base = (size_t)o * D * D + (size_t)ct * 32 * D;
#pragma unroll
    for (int k = 0; k < C::PW; ++k) {
      const int piece = wid + 4 * k;
      if (piece < C::PIECES) {
        const bool lo_plane = SPLIT && piece >= C::WPLANE / 1024;
        const uint16_t* src = (lo_plane ? wl_lo : wl) + base + w_src[k];
        glds16(src, smem + buf * C::WBUF + piece * 1024);
      }
    }
  };

  const int o_begin = blockIdx.y * labels_per_group;
  const int o_end = min(o_begin + labels_per_group, n_labels);
  const int n_chunks = (o_end - o_begin) * QT;
  if (n_chunks > 0) issue_chunk(o_begin, 0, 0);

  bf16x8 fx1[SPLIT ? 2 : 1][KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    fx1[0][k] = *reinterpret_cast<const bf16x8*>(hl + x1off + k * 16);
    if constexpr (SPLIT) fx1[1][k] = *reinterpret_cast<const bf16x8*>(hl_lo + x1off + k * 16);
  }
  // x2 of this lane's dependent at q = 32*ct + 8*rq + 4*lh + (0..3), fp32 in registers (one wave per
  // SIMD: 512 VGPRs are available, x1 fragments + x2 take 2 x 16*QT of them)
  float x2r[QT][16];
#pragma unroll
  for (int ct = 0; ct < QT; ++ct)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const uint2 v = *reinterpret_cast<const uint2*>(hl + x2off + ct * 32 + rq * 8);
      x2r[ct][rq * 4 + 0] = __uint_as_float(v.x << 16);
      x2r[ct][rq * 4 + 1] = __uint_as_float(v.x & 0xffff0000u);
      x2r[ct][rq * 4 + 2] = __uint_as_float(v.y << 16);
      x2r[ct][rq * 4 + 3] = __uint_as_float(v.y & 0xffff0000u);
      if constexpr (SPLIT) {
        const uint2 w = *reinterpret_cast<const uint2*>(hl_lo + x2off + ct * 32 + rq * 8);
        x2r[ct][rq * 4 + 0] += __uint_as_float(w.x << 16);
        x2r[ct][rq * 4 + 1] += __uint_as_float(w.x & 0xffff0000u);
        x2r[ct][rq * 4 + 2] += __uint_as_float(w.y << 16);
        x2r[ct][rq * 4 + 3] += __uint_as_float(w.y & 0xffff0000u);
      }
    }
  __syncthreads();  // chunk 0 landed (vmcnt(0) + barrier)

  const int a_rd = l31 * C::WROW + lh * 16;  // fragment read offset inside a chunk plane (k-step adds 32 bytes)
  for (int o = o_begin; o < o_end; ++o) {
    float s = 0.f;
#pragma unroll
    for (int ct = 0; ct < QT; ++ct) {
      const int n = (o - o_begin) * QT + ct;
      if (n + 1 < n_chunks) issue_chunk(ct + 1 < QT ? o : o + 1, ct + 1 < QT ? ct + 1 : 0, (n + 1) & 1);
      const unsigned char* wb = smem + (n & 1) * C::WBUF + a_rd;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const bf16x8 fw = *reinterpret_cast<const bf16x8*>(wb + k * 32);
        if constexpr (SPLIT) {
          const bf16x8 fwl = *reinterpret_cast<const bf16x8*>(wb + C::WPLANE + k * 32);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwl, fx1[0][k], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fx1[1][k], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fx1[0][k], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[r] * x2r[ct][r];
      __syncthreads();  // next chunk landed; everyone is done reading this one before it is overwritten
    }
    s += __shfl_xor(s, 32);
    if (lh == 0 && d0 + l31 < M) logits[(size_t)(d0 + l31) * n_labels + o] = s + bias[o];
  }
}

template <int QT>
int launch_label(bool split, dim3 grid, hipStream_t st, const uint16_t* hl, const uint16_t* hl_lo, int ld_hl, int col_x2,
                 const uint16_t* wl, const uint16_t* wl_lo, const float* bias, int n_labels, int per,
                 const int32_t* heads, const int32_t* lens, int B, int L, float* logits) {
  if (split) {
    constexpr int lds = LabelCfg<QT, true>::LDS;
    static CubeLdsReservation reserved;
    if (!reserved.ensure((const void*)label_kernel<QT, true>, lds))
      CUBE_FAIL(CUBE_ERR_LAUNCH, "cube_label_score: cannot reserve %d bytes of LDS", lds);
    hipLaunchKernelGGL((label_kernel<QT, true>), grid, dim3(256), lds, st, hl, hl_lo, ld_hl, col_x2, wl, wl_lo, bias,
                       n_labels, per, heads, lens, B, L, logits);
  } else {
    constexpr int lds = LabelCfg<QT, false>::LDS;
    static CubeLdsReservation reserved;
    if (!reserved.ensure((const void*)label_kernel<QT, false>, lds))
      CUBE_FAIL(CUBE_ERR_LAUNCH, "cube_label_score: cannot reserve %d bytes of LDS", lds);
    hipLaunchKernelGGL((label_kernel<QT, false>), grid, dim3(256), lds, st, hl, hl_lo, ld_hl, col_x2, wl, wl_lo, bias,
                       n_labels, per, heads, lens, B, L, logits);
  }
  return CUBE_OK;
}

}  // namespace

extern "C" int cube_label_score(const void* hl, const void* hl_lo, int ld_hl, int col_x2, int D, const void* wl,
                                const void* wl_lo, const float* bias, int n_labels, const int32_t* heads,
                                const int32_t* lens, int B, int L, float* logits, int32_t* labels, void* stream) {
  CUBE_ENTRY();
  if (B <= 0 || L <= 0 || n_labels <= 0) return CUBE_OK;
  if (!hl || !wl || !bias || !heads || !lens || !logits) CUBE_FAIL(CUBE_ERR_ARG, "cube_label_score: null pointer");
  if ((hl_lo == nullptr) != (wl_lo == nullptr)) CUBE_FAIL(CUBE_ERR_ARG, "cube_label_score: hl_lo and wl_lo go together");
  if (D <= 0 || D % 32 || D > 256 || D > col_x2 || col_x2 + D > ld_hl) CUBE_FAIL(CUBE_ERR_ARG, "cube_label_score: bad D/columns");
  if (ld_hl % 8 || col_x2 % 8 || (uintptr_t)hl % 16 || (uintptr_t)wl % 16 || (uintptr_t)hl_lo % 16 || (uintptr_t)wl_lo % 16)
    CUBE_FAIL(CUBE_ERR_ALIGN, "cube_label_score: operands must be 16-byte aligned");
  const int M = B * L;
  const int row_blocks = (M + 127) / 128;
  // Label groups: enough for a wave on every second SIMD of the chip, no more.  Alone the kernel is fastest with a wave on
  // EVERY SIMD (1024 waves: 122 us; two waves per SIMD 10 % slower: the x1 / x2 set-up is per workgroup), but it runs beside
  // the character-CNN GEMMs of the next batches, whose tiles need a whole CU each (159.8 KB of LDS, 464 of 512 registers per
  // SIMD): spread over all 256 CUs at one wave per SIMD it kept GEMM tiles off the whole chip while using a quarter of it.
  // A/B in the pipelined bench on one box: 1024 waves 170.8k sentences/s, 2048 172.0k, 512 175.7k, 256 175.7k.
#ifndef CUBE_LABEL_WAVES
#define CUBE_LABEL_WAVES 512
#endif
  int groups = (CUBE_LABEL_WAVES + row_blocks * 4 - 1) / (row_blocks * 4);
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

// Persistent recurrent-step kernel of one bidirectional LSTM layer (gfx950).
//
// One workgroup owns (direction, 16*RT sentences) for the whole sequence: it walks the T steps
// itself, so the serial dependency never leaves the CU.  Wave w owns the hidden units
// [16w, 16w+16): its four 16x16 accumulator tiles are the i,f,g,o pre-activations of the SAME
// (sentence, hidden unit) in the SAME lane/register, so the gate nonlinearities and the cell
// update are register-local (no shuffles, no LDS); only h_t crosses waves, through a
// double-buffered bf16 LDS image that is the A operand of the next step's MFMAs
// (v_mfma_f32_16x16x32_bf16, fp32 accumulate).  The input projections (all timesteps at once)
// come from cube_gemm_bf16 and are loaded into the accumulators across the step barrier.
// A step is bound by streaming the workgroup's W_hh fragments from L2 (they do not fit the
// register budget of a 13-wave workgroup), not by the MFMAs: with RT = 2 row tiles per workgroup
// every fragment serves 32 sentences, the step takes about as long, and the layer holds half as
// many CUs -- which matters because the kernel runs beside the GEMMs of the next batch, whose
// 160-KiB-LDS tiles cannot share a CU with anything.
// SPLIT ("bf16x3"): h and W_hh are hi/lo bf16 pairs, each product is hi*hi + hi*lo + lo*hi.
#include "common.h"
#include "../../include/cube_hip.h"

namespace {

constexpr int TILE = 16;  // sentences per row tile (= MFMA M)

template <int KSTEPS, bool SPLIT, int RT>
__global__ void bilstm_kernel(const float* __restrict__ gx, const uint16_t* __restrict__ whh,
                              const uint16_t* __restrict__ whh_lo, int B, int T, int H, uint16_t* __restrict__ out,
                              uint16_t* __restrict__ out_lo, int ld_out, int col0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int KH = KSTEPS * 32;
  constexpr int HROW = KH * 2 + 16;  // bytes per h row (padded: conflict-free ds_read_b128)
  const int U = (H + 15) >> 4;
  const int G = U * 64;
  const int dir = blockIdx.y;
  constexpr int ROWS = TILE * RT;  // sentences per workgroup
  const int b0 = blockIdx.x * ROWS;
  const int lane = threadIdx.x & 63;
  const int u = threadIdx.x >> 6;  // hidden-unit group of this wave
  const int l15 = lane & 15, lq = lane >> 4;

  constexpr int HIMG = ROWS * HROW;            // one h image (one plane of one time step)
  constexpr int NP = SPLIT ? 2 : 1;
  // zero all h images (h_{-1} = 0, K padding = 0); layout [step parity][plane][row][k]
  for (int i = threadIdx.x * 4; i < 2 * NP * HIMG; i += blockDim.x * 4) *reinterpret_cast<uint32_t*>(smem + i) = 0u;
  __syncthreads();

  // fragment-major weights: [dir][u][gate][k-step][lane][8]: every wave-level load is 1 KiB contiguous
  const size_t woff = (((size_t)dir * U + u) * 4 * KSTEPS) * 512 + lane * 8;
  const uint16_t* wbase = whh + woff;
  const uint16_t* wbase_lo = SPLIT ? whh_lo + woff : nullptr;
  float c[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[rt][r] = 0.f;
  const int hid = u * 16 + l15;
  const size_t gx_ld = (size_t)2 * G;
  const float* gx_col = gx + (size_t)dir * G + u * 64 + l15;

  // the accumulators start every step as the input projections of that step: they are loaded as soon
  // as the gate math of the previous step has consumed the registers (no separate prefetch copy:
  // with two row tiles that copy would not fit the 128-VGPR budget), in flight across the barrier
  f32x4 acc[RT][4];
  auto load_gx = [&](int t) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int b = b0 + rt * TILE + lq * 4 + r;
        if (b < B) {
          const float* p = gx_col + ((size_t)b * T + t) * gx_ld;
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[rt][g][r] = p[g * 16];
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[rt][g][r] = 0.f;
        }
      }
  };

  load_gx(dir ? T - 1 : 0);
  int cur = 0;
  for (int s = 0; s < T; ++s) {
    const int t = dir ? (T - 1 - s) : s;
    const unsigned char* hb = smem + cur * NP * HIMG + l15 * HROW + lq * 16;
    // W_hh is re-streamed from L2 every step: 7 k-steps x 4 gates of B fragments are 112 VGPRs per
    // wave, more than the 128-register budget of a 13-wave workgroup leaves.  Laundering the
    // pointer keeps hipcc from hoisting (and then spilling) the loads out of the time loop.
    const uint16_t* wb = wbase;
    const uint16_t* wl = wbase_lo;
    asm volatile("" : "+v"(wb));
    if constexpr (SPLIT) asm volatile("" : "+v"(wl));
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      bf16x8 a[RT], al[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        a[rt] = *reinterpret_cast<const bf16x8*>(hb + rt * TILE * HROW + ks * 64);
        if constexpr (SPLIT) al[rt] = *reinterpret_cast<const bf16x8*>(hb + HIMG + rt * TILE * HROW + ks * 64);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const bf16x8 w = *reinterpret_cast<const bf16x8*>(wb + (g * KSTEPS + ks) * 512);
        if constexpr (SPLIT) {
          const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(wl + (g * KSTEPS + ks) * 512);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            acc[rt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt], w, acc[rt][g], 0, 0, 0);
            acc[rt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt], w2, acc[rt][g], 0, 0, 0);
          }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          acc[rt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt], w, acc[rt][g], 0, 0, 0);
      }
    }
    unsigned char* hn = smem + (cur ^ 1) * NP * HIMG;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ig = sigmoidf_(acc[rt][0][r]);
        const float fg = sigmoidf_(acc[rt][1][r]);
        const float gg = tanhf_(acc[rt][2][r]);
        const float og = sigmoidf_(acc[rt][3][r]);
        c[rt][r] = fg * c[rt][r] + ig * gg;
        const float hf = og * tanhf_(c[rt][r]);
        uint16_t hv, hl = 0;
        if constexpr (SPLIT) split_bf16(hf, hv, hl); else hv = f32_to_bf16(hf);
        const int row = rt * TILE + lq * 4 + r;
        *reinterpret_cast<uint16_t*>(hn + row * HROW + hid * 2) = hv;
        if constexpr (SPLIT) *reinterpret_cast<uint16_t*>(hn + HIMG + row * HROW + hid * 2) = hl;
        const int b = b0 + row;
        if (b < B && hid < H) {
          const size_t o = ((size_t)b * T + t) * ld_out + col0 + dir * H + hid;
          out[o] = hv;
          if constexpr (SPLIT) out_lo[o] = hl;
        }
      }
    if (s + 1 < T) load_gx(dir ? t - 1 : t + 1);
    __syncthreads();
    cur ^= 1;
  }
}

}  // namespace

extern "C" int cube_bilstm_recurrent(const float* gx, const void* whh, const void* whh_lo, int B, int T, int H,
                                     void* out, void* out_lo, int ld_out, int col0, void* stream) {
  CUBE_ENTRY();
  if (B <= 0 || T <= 0) return CUBE_OK;
  if (!gx || !whh || !out) CUBE_FAIL(CUBE_ERR_ARG, "cube_bilstm_recurrent: null pointer");
  if (H <= 0 || H > 256) CUBE_FAIL(CUBE_ERR_ARG, "cube_bilstm_recurrent: rnn_size=%d unsupported (1..256)", H);
  if (col0 < 0 || col0 + 2 * H > ld_out) CUBE_FAIL(CUBE_ERR_ARG, "cube_bilstm_recurrent: bad output columns");
  if ((whh_lo == nullptr) != (out_lo == nullptr))
    CUBE_FAIL(CUBE_ERR_ARG, "cube_bilstm_recurrent: whh_lo and out_lo must be given together (bf16x3) or not at all");
  if ((uintptr_t)whh % 16 || (uintptr_t)whh_lo % 16) CUBE_FAIL(CUBE_ERR_ALIGN, "cube_bilstm_recurrent: whh must be 16-byte aligned");
  const int U = (H + 15) / 16;
  const int ksteps = (H + 31) / 32;
  const int KH = ksteps * 32;
  const bool split = whh_lo != nullptr;
  // two row tiles per workgroup once that still leaves a few workgroups per direction (see the header)
  const int rt = B >= 64 ? 2 : 1;
  const int rows = TILE * rt;
  const size_t lds = (size_t)2 * (split ? 2 : 1) * rows * (KH * 2 + 16);
  dim3 grid((B + rows - 1) / rows, 2), block(64 * U);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_RT(KS, SP, RT)                                                                                     \
  do {                                                                                                            \
    static CubeLdsReservation reserved; /* H > 224 with two row tiles in bf16x3 needs more than 64 KiB */         \
    if (!reserved.ensure((const void*)bilstm_kernel<KS, SP, RT>, lds))                                            \
      CUBE_FAIL(CUBE_ERR_LAUNCH, "cube_bilstm_recurrent: cannot reserve %zu bytes of LDS", lds);                  \
    hipLaunchKernelGGL((bilstm_kernel<KS, SP, RT>), grid, block, lds, st, gx, (const uint16_t*)whh,               \
                       (const uint16_t*)whh_lo, B, T, H, (uint16_t*)out, (uint16_t*)out_lo, ld_out, col0);        \
  } while (0)
#define LAUNCH(KS)                                                                                                \
  case KS:                                                                                                        \
    if (split) { if (rt == 2) LAUNCH_RT(KS, true, 2); else LAUNCH_RT(KS, true, 1); }                              \
    else { if (rt == 2) LAUNCH_RT(KS, false, 2); else LAUNCH_RT(KS, false, 1); }                                  \
    break;
  switch (ksteps) {
    LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7) LAUNCH(8)
    default: CUBE_FAIL(CUBE_ERR_ARG, "cube_bilstm_recurrent: ksteps=%d", ksteps);
  }
#undef LAUNCH
#undef LAUNCH_RT
  CUBE_CHECK_LAUNCH("cube_bilstm_recurrent");
  return CUBE_OK;
}

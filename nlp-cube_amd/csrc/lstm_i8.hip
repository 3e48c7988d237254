// Persistent recurrent-step kernel of one bidirectional LSTM layer with the hidden-state weights RESIDENT in
// VGPRs and LDS (gfx950), two-slice fixed point on the int8 matrix cores.
//
// One workgroup owns (direction, 16 sentences) for the whole sequence; wave u owns the hidden units
// [16u, 16u+16): its four 16x16 accumulators are the i, f, g, o pre-activations of the SAME (sentence, unit) in
// the same lane, so the gate math and the cell update are register-local and only h_t crosses waves (through a
// double-buffered LDS image that is the A operand of the next step).
//
// Arithmetic.  h = o * tanh(c) lies in (-1, 1): h = (256*h_hi + h_lo) / 32512 with two int8 slices (16 bits, the
// format of cube_gemm_i8x2).  A row n of W_hh is fixed point with its own scale, an int8 high slice and a low slice of
// WLB = 6 (default) or 4 bits:
//   w[n, k] = (2^WLB * w_hi + w_lo) * wmax[n] / (127 * 2^WLB),   w_hi in [-127, 127], w_lo in [-2^(WLB-1), 2^(WLB-1)).
// The products h_hi*w_hi, h_hi*w_lo and h_lo*w_hi are accumulated EXACTLY in int32 (v_mfma_i32_16x16x64_i8, and
// the K = 32 form for the last 32 columns); h_lo*w_lo (2^-(8+WLB) relative) is dropped:
//   acc = 256*HH + 2^(8-WLB)*HL + LH,   h . w = acc * wmax / (32512 * 127).
// Why these widths (scripts/lstm_prec_study.py, arc scores of the 256 x 32 bench batch against a float64 recurrence,
// rms relative to the largest score / heads that differ): 15-bit h x 12-bit W (round 2) 8.0e-6 / 9; 16 x 12: 7.0e-6 / 4;
// 16 x 14 (this kernel) 3.7e-6 / 4; 16 x 16: 3.1e-6 / 3 -- the bf16x3 kernels of the rest of the path sit at 3.5e-6.
//
// Where the weights live (H = 200: U = 13 waves, K padded to 224):
//   w_hi   56 VGPRs per wave: 4 gates x (3 fragments of 16 bytes + one of 8), loaded once
//   w_lo   LDS, unsigned (w_lo + 2^(WLB-1)) in fragment order.  WLB = 6: four 6-bit values in three bytes, a fragment
//          (16 values of a lane) is a ds_read_b64 + a ds_read_b32 + nine VALU ops (130 KB for the workgroup); WLB = 4: nibbles two to
//          a byte, one ds_read_b64 + six VALU ops (93 KB).  The offset is taken out again with the row sums of h_hi,
//          which one extra MFMA per k-step against an all-ones operand delivers in the accumulator layout:
//          sum_k h_hi*(w_lo + off) - off * sum_k h_hi.
// Nothing of W_hh crosses L2 after the prologue.  The input projections of a step (fp32, from cube_gemm_bf16): every lane
// loads its sixteen values of the NEXT step into registers right after the gate math (in flight across the barrier and
// the MFMA phases); the gate rows' scales are read from LDS in the gate phase instead of holding four registers.  (Round 2
// staged the projections through 53 KB of LDS by DMA: 4.29 us per step against 4.10 this way at the same slice widths,
// and the LDS is what the wider low slice needs.)
#include "common.h"
#include "../../include/cube_hip.h"

namespace {

typedef long i64_t;
constexpr int LSTM8_HSHIFT = CUBE_I8_SHIFT;  // h = (256*hi + lo) / CUBE_I8_ONE: the 16-bit planes of cube_gemm_i8x2

// The instruction scheduler must not move loads across k-step blocks: with 13 waves in the workgroup a wave has 128
// registers, 100 of them hold weights and state, and hoisted fragment loads would be paid for in scratch spills.
// Latency is hidden by the other three waves of the SIMD instead.
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

constexpr int LSTM8_ROWS = 16;  // sentences per workgroup (= MFMA M)

template <int U, int WLB>
struct Lstm8Cfg {
  static constexpr int KP = 32 * ((U + 1) / 2);        // K (hidden size) padded to 32
  static constexpr int KS64 = KP / 64;                 // K = 64 MFMA steps
  static constexpr bool TAIL = (KP % 64) != 0;         // + one K = 32 step
  static constexpr int HROW = KP + 16;                 // bytes per h row and plane (conflict-free b128 reads)
  static constexpr int HIMG = LSTM8_ROWS * HROW;       // one plane of one time step
  static constexpr int WHI_TILE = KS64 * 1024 + (TAIL ? 512 : 0);  // bytes of one (u, gate) tile of w_hi fragments
  // 16-lane groups of the tail fragment that can hold a hidden unit < 16 U (the others meet h = 0 and read group 0's bytes)
  static constexpr int TAILQ = TAIL ? (16 * U - 64 * KS64 + 7) / 8 : 0;
  static constexpr int WLO_TILE = WLB == 6 ? KS64 * 768 + TAILQ * 128   // 6 bits: [lane][12] per fragment, tail as whole bytes
                                           : KS64 * 512 + (TAIL ? 256 : 0);  // nibbles: [lane][8], tail [lane][4]
  static constexpr int H_BYTES = 4 * HIMG;             // [2 time steps][2 planes]
  static constexpr int WLO_BYTES = U * 4 * WLO_TILE;
  static constexpr int SC_BYTES = U * 64 * 4;          // the gate rows' scales (read in the gate phase: no registers)
  static constexpr size_t LDS = (size_t)H_BYTES + WLO_BYTES + SC_BYTES;
};

template <int U, int WLB>
__global__ __launch_bounds__(64 * U) void bilstm_i8_kernel(const float* __restrict__ gx, const unsigned char* __restrict__ whi,
                                                           const unsigned char* __restrict__ wlo, const float* __restrict__ wscale,
                                                           int B, int T, int H, uint16_t* __restrict__ out,
                                                           uint16_t* __restrict__ out_lo, int ld_out, int col0) {
  using C = Lstm8Cfg<U, WLB>;
  constexpr int KS64 = C::KS64;
  constexpr bool TAIL = C::TAIL;
  constexpr int HROW = C::HROW, HIMG = C::HIMG;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const s_h = smem;
  unsigned char* const s_wlo = smem + C::H_BYTES;
  float* const s_sc = reinterpret_cast<float*>(smem + C::H_BYTES + C::WLO_BYTES);

  const int G = U * 64;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * LSTM8_ROWS;
  const int lane = threadIdx.x & 63;
  const int u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: tile bases below stay in SGPRs
  const int l15 = lane & 15, lq = lane >> 4;

  // h_{-1} = 0 and the K padding stay zero
  for (int i = threadIdx.x * 4; i < C::H_BYTES; i += 64 * U * 4) *reinterpret_cast<uint32_t*>(s_h + i) = 0u;

  // ---- the weights come in once: w_hi into registers, this wave's packed w_lo slices and scales into LDS
  i32x4 wh[4][KS64 > 0 ? KS64 : 1];
  i64_t wt[4];
  {
    const unsigned char* base = whi + ((size_t)(dir * U + u) * 4) * C::WHI_TILE;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int ks = 0; ks < KS64; ++ks)
        wh[g][ks] = *reinterpret_cast<const i32x4*>(base + (size_t)g * C::WHI_TILE + ks * 1024 + lane * 16);
      if constexpr (TAIL) wt[g] = *reinterpret_cast<const i64_t*>(base + (size_t)g * C::WHI_TILE + KS64 * 1024 + lane * 8);
      else wt[g] = 0;
    }
    const unsigned char* lsrc = wlo + ((size_t)(dir * U + u) * 4) * C::WLO_TILE;
    unsigned char* ldst = s_wlo + (size_t)u * 4 * C::WLO_TILE;
    for (int i = lane * 4; i < 4 * C::WLO_TILE; i += 64 * 4)
      *reinterpret_cast<uint32_t*>(ldst + i) = *reinterpret_cast<const uint32_t*>(lsrc + i);
    s_sc[u * 64 + lane] = wscale[(size_t)dir * G + u * 64 + lane];
  }
  const float* const my_sc = s_sc + u * 64 + l15;  // + g * 16
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  const int hid = u * 16 + l15;
  // Addresses: a wave-uniform base that moves with the time step (SGPRs) + a per-lane 32-bit byte offset that does
  // not (the host checks that both tensors are below 2 GiB).  Rows past the batch read the last sentence; their
  // h is computed and never stored.
  const int gx_ld = 2 * G;
  // (sentence of row lq*4 + r) * T, the per-lane part of both offsets, is recomputed from the lane id at every use
  // (two VALU operations) instead of living in four registers: `opaque` keeps the compiler from hoisting it back.
  auto bt = [&](int r) -> uint32_t {
    int q = lq;
    asm volatile("" : "+v"(q));
    return (uint32_t)(min(b0 + q * 4 + r, B - 1) * T);
  };
  const float* const gx_tile = gx + (size_t)dir * G + u * 64;   // + t * gx_ld + g * 16
  float gxv[4][4];
  auto fetch_gx = [&](int t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const char* base = reinterpret_cast<const char*>(gx_tile + (size_t)t * gx_ld + g * 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* p = reinterpret_cast<const float*>(base + (bt(r) * (uint32_t)gx_ld + (uint32_t)l15) * 4u);
        gxv[g][r] = *p;
      }
    }
  };
  __syncthreads();
  fetch_gx(dir ? T - 1 : 0);

  const unsigned char* const my_wlo = s_wlo + (size_t)u * 4 * C::WLO_TILE;
  // the tail fragment of the 6-bit slice: whole bytes for the lane groups that can meet a hidden unit, group 0's otherwise
  const int tail_off = KS64 * 768 + ((lq < C::TAILQ ? lq : 0) * 16 + l15) * 8;
  int cur = 0;
  for (int s = 0; s < T; ++s) {
    const int t = dir ? (T - 1 - s) : s;
    const unsigned char* hb = s_h + cur * 2 * HIMG + l15 * HROW;  // hi plane of h_{t-1}; the lo plane is HIMG further
    i32x4 acc[4], accs = {0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (i32x4){0, 0, 0, 0};
    // ---- HH, and the row sums of h_hi
    {
      const i32x4 ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
#pragma unroll
      for (int ks = 0; ks < KS64; ++ks) {
        const i32x4 ah = *reinterpret_cast<const i32x4*>(hb + ks * 64 + lq * 16);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ah, wh[g][ks], acc[g], 0, 0, 0);
        accs = __builtin_amdgcn_mfma_i32_16x16x64_i8(ah, ones, accs, 0, 0, 0);
      }
      if constexpr (TAIL) {
        const i64_t ah = *reinterpret_cast<const i64_t*>(hb + KS64 * 64 + lq * 8);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_i32_16x16x32_i8(ah, wt[g], acc[g], 0, 0, 0);
        accs = __builtin_amdgcn_mfma_i32_16x16x32_i8(ah, 0x0101010101010101L, accs, 0, 0, 0);
        SCHED_FENCE();
      }
    }
    // ---- 2^WLB * HH + h_hi . (w_lo + 2^(WLB-1))
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = acc[g] << WLB;
#pragma unroll
    for (int ks = 0; ks < KS64; ++ks) {
      const i32x4 ah = *reinterpret_cast<const i32x4*>(hb + ks * 64 + lq * 16);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        i32x4 wl;
        if constexpr (WLB == 6) {
          // bytes 4m + b of the fragment are the low six bits of dword m (m < 3); bytes 12 + b are collected from the
          // top two bits of the three dwords
          // (words 0 and 1 of all lanes, then word 2 of all lanes: two aligned, conflict-free reads; a 12-byte lane
          // stride -- one ds_read_b96 -- is not 16-byte aligned and ran at half the speed)
          const unsigned char* f = my_wlo + g * C::WLO_TILE + ks * 768;
          const uint2 p = *reinterpret_cast<const uint2*>(f + lane * 8);
          const uint32_t pz = *reinterpret_cast<const uint32_t*>(f + 512 + lane * 4);
          wl = (i32x4){(int)(p.x & 0x3f3f3f3fu), (int)(p.y & 0x3f3f3f3fu), (int)(pz & 0x3f3f3f3fu),
                       (int)(((p.x >> 6) & 0x03030303u) | ((p.y >> 4) & 0x0c0c0c0cu) | ((pz >> 2) & 0x30303030u))};
        } else {
          const uint2 p = *reinterpret_cast<const uint2*>(my_wlo + g * C::WLO_TILE + ks * 512 + lane * 8);
          wl = (i32x4){(int)(p.x & 0x0f0f0f0fu), (int)((p.x >> 4) & 0x0f0f0f0fu), (int)(p.y & 0x0f0f0f0fu),
                       (int)((p.y >> 4) & 0x0f0f0f0fu)};
        }
        acc[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ah, wl, acc[g], 0, 0, 0);
      }
      SCHED_FENCE();
    }
    if constexpr (TAIL) {
      const i64_t ah = *reinterpret_cast<const i64_t*>(hb + KS64 * 64 + lq * 8);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        i64_t wl;
        if constexpr (WLB == 6) {
          wl = *reinterpret_cast<const i64_t*>(my_wlo + g * C::WLO_TILE + tail_off);
        } else {
          const uint32_t p = *reinterpret_cast<const uint32_t*>(my_wlo + g * C::WLO_TILE + KS64 * 512 + lane * 4);
          wl = (i64_t)(((unsigned long)((p >> 4) & 0x0f0f0f0fu) << 32) | (unsigned long)(p & 0x0f0f0f0fu));
        }
        acc[g] = __builtin_amdgcn_mfma_i32_16x16x32_i8(ah, wl, acc[g], 0, 0, 0);
      }
      SCHED_FENCE();
    }
    // ---- (2^WLB * HH + HL) * 2^(8 - WLB) + h_lo . w_hi
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (acc[g] - (accs << (WLB - 1))) << (LSTM8_HSHIFT - WLB);
#pragma unroll
    for (int ks = 0; ks < KS64; ++ks) {
      const i32x4 al = *reinterpret_cast<const i32x4*>(hb + HIMG + ks * 64 + lq * 16);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(al, wh[g][ks], acc[g], 0, 0, 0);
    }
    if constexpr (TAIL) {
      const i64_t al = *reinterpret_cast<const i64_t*>(hb + HIMG + KS64 * 64 + lq * 8);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_i32_16x16x32_i8(al, wt[g], acc[g], 0, 0, 0);
      SCHED_FENCE();
    }
    // ---- gates (the input projections of this step were requested at the end of the step before)
    unsigned char* hn = s_h + (cur ^ 1) * 2 * HIMG;
    const float sc0 = my_sc[0], sc1 = my_sc[16], sc2 = my_sc[32], sc3 = my_sc[48];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pi = (float)acc[0][r] * sc0 + gxv[0][r];
      const float pf = (float)acc[1][r] * sc1 + gxv[1][r];
      const float pg = fminf(fmaxf((float)acc[2][r] * sc2 + gxv[2][r], -15.f), 15.f);
      const float po = (float)acc[3][r] * sc3 + gxv[3][r];
      const float eg = __expf(-2.0f * pg);
      const float ig_gg = (1.0f - eg) * __builtin_amdgcn_rcpf((1.0f + __expf(-pi)) * (1.0f + eg));
      c[r] = sigmoidf_(pf) * c[r] + ig_gg;
      const float ec = __expf(-2.0f * fminf(fmaxf(c[r], -15.f), 15.f));
      const float hf = (1.0f - ec) * __builtin_amdgcn_rcpf((1.0f + __expf(-po)) * (1.0f + ec));
      int q = __float2int_rn(hf * (float)CUBE_I8_ONE);
      q = q > CUBE_I8_ONE ? CUBE_I8_ONE : (q < -CUBE_I8_ONE ? -CUBE_I8_ONE : q);
      const int qh = (q + (1 << (LSTM8_HSHIFT - 1))) >> LSTM8_HSHIFT;  // arithmetic shift = floor: hi in [-127, 127], lo in [-128, 127]
      const int row = lq * 4 + r;
      hn[row * HROW + hid] = (unsigned char)qh;
      hn[HIMG + row * HROW + hid] = (unsigned char)(q - (qh << LSTM8_HSHIFT));
      if (b0 + row < B && hid < H) {
        uint16_t hv, hl;
        split_bf16(hf, hv, hl);
        const size_t ob = ((size_t)t * ld_out + col0 + dir * H) * 2;  // wave-uniform
        const uint32_t oo = (bt(r) * (uint32_t)ld_out + (uint32_t)hid) * 2u;
        *reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(out) + ob + oo) = hv;
        if (out_lo) *reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(out_lo) + ob + oo) = hl;
      }
    }
    if (s + 1 < T) fetch_gx(dir ? t - 1 : t + 1);  // in flight across the barrier and the MFMA phases of the next step
    __syncthreads();
    cur ^= 1;
  }
}

template <int U, int WLB>
int launch_lstm8(const float* gx, const void* whi, const void* wlo, const float* wscale, int B, int T, int H, void* out,
                 void* out_lo, int ld_out, int col0, hipStream_t st) {
  using C = Lstm8Cfg<U, WLB>;
  constexpr size_t lds = C::LDS;
  if constexpr (lds > 160 * 1024) {
    CUBE_FAIL(CUBE_ERR_ARG, "cube_bilstm_recurrent_i8: rnn_size %d with %d-bit low slices needs %zu bytes of LDS", H, WLB, lds);
  } else {
    static CubeLdsReservation reserved;
    const void* fn = (const void*)bilstm_i8_kernel<U, WLB>;
    if (!reserved.ensure(fn, lds)) CUBE_FAIL(CUBE_ERR_LAUNCH, "cube_bilstm_recurrent_i8: cannot reserve %zu bytes of LDS", lds);
    dim3 grid((B + LSTM8_ROWS - 1) / LSTM8_ROWS, 2), block(64 * U);
    hipLaunchKernelGGL((bilstm_i8_kernel<U, WLB>), grid, block, lds, st, gx, (const unsigned char*)whi,
                       (const unsigned char*)wlo, wscale, B, T, H, (uint16_t*)out, (uint16_t*)out_lo, ld_out, col0);
    CUBE_CHECK_LAUNCH("cube_bilstm_recurrent_i8");
    return CUBE_OK;
  }
}

}  // namespace

extern "C" int cube_bilstm_lo_bits(int H) {
  // the widest low slice of W_hh whose packed form fits the LDS of a CU beside the h images (csrc/lstm_i8.hip)
  if (H <= 0 || H > 256) return 0;
  const int U = (H + 15) / 16, KP = 32 * ((U + 1) / 2), KS64 = KP / 64;
  const int tailq = KP % 64 ? (16 * U - 64 * KS64 + 7) / 8 : 0;
  const size_t lds6 = (size_t)4 * 16 * (KP + 16) + (size_t)U * 4 * (KS64 * 768 + tailq * 128) + (size_t)U * 256;  // Lstm8Cfg<U, 6>::LDS
  return lds6 <= 160 * 1024 ? 6 : 4;
}

extern "C" int cube_bilstm_recurrent_i8(const float* gx, const void* whi, const void* wlo, const float* wscale, int B, int T,
                                        int H, int lo_bits, void* out, void* out_lo, int ld_out, int col0, void* stream) {
  CUBE_ENTRY();
  if (B <= 0 || T <= 0) return CUBE_OK;
  if (!gx || !whi || !wlo || !wscale || !out) CUBE_FAIL(CUBE_ERR_ARG, "cube_bilstm_recurrent_i8: null pointer");
  if (H <= 0 || H > 256) CUBE_FAIL(CUBE_ERR_ARG, "cube_bilstm_recurrent_i8: rnn_size=%d unsupported (1..256)", H);
  if (lo_bits != 4 && lo_bits != 6) CUBE_FAIL(CUBE_ERR_ARG, "cube_bilstm_recurrent_i8: lo_bits=%d (4 or 6)", lo_bits);
  if (lo_bits == 6 && cube_bilstm_lo_bits(H) != 6)
    CUBE_FAIL(CUBE_ERR_ARG, "cube_bilstm_recurrent_i8: 6-bit low slices do not fit the LDS for rnn_size=%d (cube_bilstm_lo_bits)", H);
  if (col0 < 0 || col0 + 2 * H > ld_out) CUBE_FAIL(CUBE_ERR_ARG, "cube_bilstm_recurrent_i8: bad output columns");
  if ((uintptr_t)whi % 16 || (uintptr_t)wlo % 16) CUBE_FAIL(CUBE_ERR_ALIGN, "cube_bilstm_recurrent_i8: weights must be 16-byte aligned");
  const int U = (H + 15) / 16;
  if ((size_t)B * T * 2 * U * 64 * 4 >= ((size_t)1 << 31) || (size_t)B * T * ld_out * 2 >= ((size_t)1 << 31))
    CUBE_FAIL(CUBE_ERR_ARG, "cube_bilstm_recurrent_i8: B*T = %d is too large for 32-bit lane offsets", B * T);
  hipStream_t st = (hipStream_t)stream;
#define CASE(N) case N: return lo_bits == 6 ? launch_lstm8<N, 6>(gx, whi, wlo, wscale, B, T, H, out, out_lo, ld_out, col0, st) \
                                            : launch_lstm8<N, 4>(gx, whi, wlo, wscale, B, T, H, out, out_lo, ld_out, col0, st);
  switch (U) {
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14)
    CASE(15) CASE(16)
  }
#undef CASE
  CUBE_FAIL(CUBE_ERR_ARG, "cube_bilstm_recurrent_i8: rnn_size=%d", H);
}

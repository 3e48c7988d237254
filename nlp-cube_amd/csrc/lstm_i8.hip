// Persistent recurrent-step kernel of one bidirectional LSTM layer with the hidden-state weights RESIDENT in
// VGPRs and LDS (gfx950), two-slice fixed point on the int8 matrix cores.
//
// One workgroup owns (direction, 16 sentences) for the whole sequence; wave u owns the hidden units
// [16u, 16u+16): its four 16x16 accumulators are the i, f, g, o pre-activations of the SAME (sentence, unit) in
// the same lane, so the gate math and the cell update are register-local and only h_t crosses waves (through a
// double-buffered LDS image that is the A operand of the next step).
//
// Arithmetic.  h = o * tanh(c) lies in (-1, 1): h = (128*h_hi + h_lo) / 16256 with two int8 slices (15 bits, the
// format of cube_gemm_i8x2).  A row n of W_hh is 12-bit fixed point with its own scale:
//   w[n, k] = (16*w_hi + w_lo) * wmax[n] / 2032,   w_hi in [-127, 127] (int8), w_lo in [-8, 7] (a nibble).
// The products h_hi*w_hi, h_hi*w_lo and h_lo*w_hi are accumulated EXACTLY in int32 (v_mfma_i32_16x16x64_i8, and
// the K = 32 form for the last 32 columns); h_lo*w_lo (2^-12 relative) is dropped:
//   acc = 128*HH + 8*HL + LH,   h . w = acc * 16 * wmax / (16256 * 2032).
// Measured on the full-size model against the fp32 oracle (tests/study_lstm_int8x2.py, WLO=4): arc scores
// 2.2e-5 .. 4.3e-5, label scores 2.0e-5 .. 2.8e-5, all heads equal.
//
// Where the weights live (H = 200: U = 13 waves, K padded to 224):
//   w_hi   56 VGPRs per wave: 4 gates x (3 fragments of 16 bytes + one of 8), loaded once
//   w_lo   LDS, as unsigned nibbles w_lo + 8 packed two to a byte in fragment order (93 KB for the workgroup);
//          a fragment is one conflict-free ds_read_b64 + three VALU ops to unpack; the offset is taken out again
//          with the row sums of h_hi, which one extra MFMA per k-step against an all-ones operand delivers in the
//          accumulator layout:  sum_k h_hi*(w_lo + 8) - 8 * sum_k h_hi.
// Nothing of W_hh crosses L2 after the prologue.  The input projections of a step (fp32, from cube_gemm_bf16)
// come in by LDS-DMA during the MFMAs of the step before and are read once in the gate phase, so they hold no
// registers either (53 KB of LDS; for H > 208 they do not fit and are loaded into registers instead).
#include "common.h"
#include "../../include/cube_hip.h"

namespace {

typedef long i64_t;

// The instruction scheduler must not move loads across k-step blocks: with 13 waves in the workgroup a wave has 128
// registers, 100 of them hold weights and state, and hoisted fragment loads would be paid for in scratch spills.
// Latency is hidden by the other three waves of the SIMD instead.
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

constexpr int LSTM8_ROWS = 16;  // sentences per workgroup (= MFMA M)

template <int U>
struct Lstm8Cfg {
  static constexpr int KP = 32 * ((U + 1) / 2);        // K (hidden size) padded to 32
  static constexpr int KS64 = KP / 64;                 // K = 64 MFMA steps
  static constexpr bool TAIL = (KP % 64) != 0;         // + one K = 32 step
  static constexpr int HROW = KP + 16;                 // bytes per h row and plane (conflict-free b128 reads)
  static constexpr int HIMG = LSTM8_ROWS * HROW;       // one plane of one time step
  static constexpr int WHI_TILE = KS64 * 1024 + (TAIL ? 512 : 0);  // bytes of one (u, gate) tile of w_hi fragments
  static constexpr int WLO_TILE = KS64 * 512 + (TAIL ? 256 : 0);   // ... of packed w_lo nibbles
  static constexpr int H_BYTES = 4 * HIMG;             // [2 time steps][2 planes]
  static constexpr int WLO_BYTES = U * 4 * WLO_TILE;
  static constexpr int GX_BYTES = U * 16 * 256;        // per wave 16 (gate, row) blocks of 64 floats
  static constexpr size_t lds(bool gx_lds) { return (size_t)H_BYTES + WLO_BYTES + (gx_lds ? GX_BYTES : 0); }
};

__device__ __forceinline__ void glds4(const float* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}

template <int U, bool GXL>
__global__ __launch_bounds__(64 * U) void bilstm_i8_kernel(const float* __restrict__ gx, const unsigned char* __restrict__ whi,
                                                           const unsigned char* __restrict__ wlo, const float* __restrict__ wscale,
                                                           int B, int T, int H, uint16_t* __restrict__ out,
                                                           uint16_t* __restrict__ out_lo, int ld_out, int col0) {
  using C = Lstm8Cfg<U>;
  constexpr int KS64 = C::KS64;
  constexpr bool TAIL = C::TAIL;
  constexpr int HROW = C::HROW, HIMG = C::HIMG;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const s_h = smem;
  unsigned char* const s_wlo = smem + C::H_BYTES;
  unsigned char* const s_gx = smem + C::H_BYTES + C::WLO_BYTES;

  const int G = U * 64;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * LSTM8_ROWS;
  const int lane = threadIdx.x & 63;
  const int u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: tile bases below stay in SGPRs
  const int l15 = lane & 15, lq = lane >> 4;

  // h_{-1} = 0 and the K padding stay zero
  for (int i = threadIdx.x * 4; i < C::H_BYTES; i += 64 * U * 4) *reinterpret_cast<uint32_t*>(s_h + i) = 0u;

  // ---- the weights come in once: w_hi into registers, this wave's packed w_lo nibbles into LDS
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
  }
  float sc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) sc[g] = wscale[(size_t)dir * G + u * 64 + g * 16 + l15];
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
  unsigned char* const gx_mine = s_gx + (size_t)u * 16 * 256;
  float gxv[4][4];
  auto fetch_gx = [&](int t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const char* base = reinterpret_cast<const char*>(gx_tile + (size_t)t * gx_ld + g * 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* p = reinterpret_cast<const float*>(base + (bt(r) * (uint32_t)gx_ld + (uint32_t)l15) * 4u);
        if constexpr (GXL) glds4(p, gx_mine + (g * 4 + r) * 256);  // lands at ... + lane * 4
        else gxv[g][r] = *p;
      }
    }
  };
  __syncthreads();
  fetch_gx(dir ? T - 1 : 0);

  const i32x4 ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
  const i64_t ones8 = 0x0101010101010101L;
  const unsigned char* const my_wlo = s_wlo + (size_t)u * 4 * C::WLO_TILE;
  int cur = 0;
  for (int s = 0; s < T; ++s) {
    const int t = dir ? (T - 1 - s) : s;
    const unsigned char* hb = s_h + cur * 2 * HIMG + l15 * HROW;  // hi plane of h_{t-1}; the lo plane is HIMG further
    i32x4 acc[4], accs = {0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (i32x4){0, 0, 0, 0};
    // ---- HH, and the row sums of h_hi
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
      accs = __builtin_amdgcn_mfma_i32_16x16x32_i8(ah, ones8, accs, 0, 0, 0);
      SCHED_FENCE();
    }
    // ---- 16*HH + h_hi . (w_lo + 8)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = acc[g] << 4;
#pragma unroll
    for (int ks = 0; ks < KS64; ++ks) {
      const i32x4 ah = *reinterpret_cast<const i32x4*>(hb + ks * 64 + lq * 16);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint2 p = *reinterpret_cast<const uint2*>(my_wlo + g * C::WLO_TILE + ks * 512 + lane * 8);
        const i32x4 wl = {(int)(p.x & 0x0f0f0f0fu), (int)((p.x >> 4) & 0x0f0f0f0fu), (int)(p.y & 0x0f0f0f0fu),
                          (int)((p.y >> 4) & 0x0f0f0f0fu)};
        acc[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ah, wl, acc[g], 0, 0, 0);
      }
      SCHED_FENCE();
    }
    if constexpr (TAIL) {
      const i64_t ah = *reinterpret_cast<const i64_t*>(hb + KS64 * 64 + lq * 8);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint32_t p = *reinterpret_cast<const uint32_t*>(my_wlo + g * C::WLO_TILE + KS64 * 512 + lane * 4);
        const i64_t wl = (i64_t)(((unsigned long)((p >> 4) & 0x0f0f0f0fu) << 32) | (unsigned long)(p & 0x0f0f0f0fu));
        acc[g] = __builtin_amdgcn_mfma_i32_16x16x32_i8(ah, wl, acc[g], 0, 0, 0);
      }
      SCHED_FENCE();
    }
    // ---- (16*HH + HL) * 8 + h_lo . w_hi
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (acc[g] - (accs << 3)) << 3;
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
    // ---- gates (the input projections of this step have landed in LDS by now)
    if constexpr (GXL) {
      // all sixteen values out of LDS first, then the request for the NEXT step goes out before the gate math instead
      // of after it: the projections come from HBM and need every bit of lead they can get
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) gxv[g][r] = *reinterpret_cast<const float*>(gx_mine + (g * 4 + r) * 256 + lane * 4);
      if (s + 1 < T) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads above are done before the DMA rewrites the area
        fetch_gx(dir ? t - 1 : t + 1);
      }
    }
    unsigned char* hn = s_h + (cur ^ 1) * 2 * HIMG;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pi = (float)acc[0][r] * sc[0] + gxv[0][r];
      const float pf = (float)acc[1][r] * sc[1] + gxv[1][r];
      const float pg = fminf(fmaxf((float)acc[2][r] * sc[2] + gxv[2][r], -15.f), 15.f);
      const float po = (float)acc[3][r] * sc[3] + gxv[3][r];
      const float eg = __expf(-2.0f * pg);
      const float ig_gg = (1.0f - eg) * __builtin_amdgcn_rcpf((1.0f + __expf(-pi)) * (1.0f + eg));
      c[r] = sigmoidf_(pf) * c[r] + ig_gg;
      const float ec = __expf(-2.0f * fminf(fmaxf(c[r], -15.f), 15.f));
      const float hf = (1.0f - ec) * __builtin_amdgcn_rcpf((1.0f + __expf(-po)) * (1.0f + ec));
      int q = __float2int_rn(hf * (float)CUBE_I8_ONE);
      q = q > CUBE_I8_ONE ? CUBE_I8_ONE : (q < -CUBE_I8_ONE ? -CUBE_I8_ONE : q);
      const int qh = (q + 64) >> 7;  // arithmetic shift = floor: hi in [-127, 127], lo in [-64, 63]
      const int row = lq * 4 + r;
      hn[row * HROW + hid] = (unsigned char)qh;
      hn[HIMG + row * HROW + hid] = (unsigned char)(q - (qh << 7));
      if (b0 + row < B && hid < H) {
        uint16_t hv, hl;
        split_bf16(hf, hv, hl);
        const size_t ob = ((size_t)t * ld_out + col0 + dir * H) * 2;  // wave-uniform
        const uint32_t oo = (bt(r) * (uint32_t)ld_out + (uint32_t)hid) * 2u;
        *reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(out) + ob + oo) = hv;
        if (out_lo) *reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(out_lo) + ob + oo) = hl;
      }
    }
    if constexpr (!GXL)
      if (s + 1 < T) fetch_gx(dir ? t - 1 : t + 1);
    __syncthreads();
    cur ^= 1;
  }
}

template <int U>
int launch_lstm8(const float* gx, const void* whi, const void* wlo, const float* wscale, int B, int T, int H, void* out,
                 void* out_lo, int ld_out, int col0, hipStream_t st) {
  using C = Lstm8Cfg<U>;
  constexpr bool fits = C::lds(true) <= 160 * 1024;
  const size_t lds = C::lds(fits);
  if (lds > 160 * 1024) CUBE_FAIL(CUBE_ERR_ARG, "cube_bilstm_recurrent_i8: rnn_size needs %zu bytes of LDS", lds);
  static CubeLdsReservation reserved;
  const void* fn = (const void*)bilstm_i8_kernel<U, fits>;
  if (!reserved.ensure(fn, lds)) CUBE_FAIL(CUBE_ERR_LAUNCH, "cube_bilstm_recurrent_i8: cannot reserve %zu bytes of LDS", lds);
  dim3 grid((B + LSTM8_ROWS - 1) / LSTM8_ROWS, 2), block(64 * U);
  hipLaunchKernelGGL((bilstm_i8_kernel<U, fits>), grid, block, lds, st, gx, (const unsigned char*)whi, (const unsigned char*)wlo,
                     wscale, B, T, H, (uint16_t*)out, (uint16_t*)out_lo, ld_out, col0);
  CUBE_CHECK_LAUNCH("cube_bilstm_recurrent_i8");
  return CUBE_OK;
}

}  // namespace

extern "C" int cube_bilstm_recurrent_i8(const float* gx, const void* whi, const void* wlo4, const float* wscale, int B, int T,
                                        int H, void* out, void* out_lo, int ld_out, int col0, void* stream) {
  CUBE_ENTRY();
  if (B <= 0 || T <= 0) return CUBE_OK;
  if (!gx || !whi || !wlo4 || !wscale || !out) CUBE_FAIL(CUBE_ERR_ARG, "cube_bilstm_recurrent_i8: null pointer");
  if (H <= 0 || H > 256) CUBE_FAIL(CUBE_ERR_ARG, "cube_bilstm_recurrent_i8: rnn_size=%d unsupported (1..256)", H);
  if (col0 < 0 || col0 + 2 * H > ld_out) CUBE_FAIL(CUBE_ERR_ARG, "cube_bilstm_recurrent_i8: bad output columns");
  if ((uintptr_t)whi % 16 || (uintptr_t)wlo4 % 16) CUBE_FAIL(CUBE_ERR_ALIGN, "cube_bilstm_recurrent_i8: weights must be 16-byte aligned");
  const int U = (H + 15) / 16;
  if ((size_t)B * T * 2 * U * 64 * 4 >= ((size_t)1 << 31) || (size_t)B * T * ld_out * 2 >= ((size_t)1 << 31))
    CUBE_FAIL(CUBE_ERR_ARG, "cube_bilstm_recurrent_i8: B*T = %d is too large for 32-bit lane offsets", B * T);
  hipStream_t st = (hipStream_t)stream;
#define CASE(N) case N: return launch_lstm8<N>(gx, whi, wlo4, wscale, B, T, H, out, out_lo, ld_out, col0, st);
  switch (U) {
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14)
    CASE(15) CASE(16)
  }
#undef CASE
  CUBE_FAIL(CUBE_ERR_ARG, "cube_bilstm_recurrent_i8: rnn_size=%d", H);
}

// Lemmatizer decoder step kernels (reference: cube/networks/lemmatizer.py:135-169,
// cube/networks/modules.py:145-169).  The GEMMs of a step run in cube_gemm_bf16; these two kernels
// do what sits between them: arg-max + embedding feedback + attention + context (cube_lemma_attend)
// and the LSTM cell update (cube_lstm_cell).  Both are small row-wise kernels, one wave per word.
#include "common.h"
#include "../../include/cube_hip.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// first index of the maximum of x[0..n) (torch.argmax), one wave
__device__ __forceinline__ int wave_argmax(const float* x, int n, int lane) {
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = lane; i < n; i += 64) {
    const float v = x[i];
    if (v > best) { best = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  return bi == 0x7fffffff ? 0 : bi;
}

constexpr int ATT_MAX_S = 256;   // source positions per word (words of <= 25 characters need 27)
constexpr int ATT_MAX_P = 128;   // attention projection size
constexpr int ATT_MAX_C = 512;   // encoder channels
constexpr int ATT_MAX_TAPS = 5;

__global__ __launch_bounds__(256) void lemma_attend_kernel(cube_lemma_attend_args a) {
  __shared__ float se[4][ATT_MAX_S];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= a.B) return;  // wave-uniform; no block-level barrier below
  const float* q = a.q + (size_t)b * a.ld_q;

  // ---- previous step's outputs: arg-max character / case, feedback embedding, done flags
  uint16_t* xrow = (uint16_t*)a.x + (size_t)b * a.ld_x;
  uint16_t* xrow_lo = a.x_lo ? (uint16_t*)a.x_lo + (size_t)b * a.ld_x : nullptr;
  if (a.step > 0) {
    const int ch = wave_argmax(q + a.col_char, a.n_char, lane);
    const int cs = wave_argmax(q + a.col_case, a.n_case, lane);
    if (lane == 0) {
      a.out_char[(size_t)b * a.ld_out + a.step - 1] = ch;
      a.out_case[(size_t)b * a.ld_out + a.step - 1] = cs;
      if (ch == a.eol && a.done[b] == 0) {
        a.done[b] = 1;
        atomicAdd(a.n_done, 1);
      }
    }
    const uint16_t* eh = (const uint16_t*)a.char_emb + (size_t)ch * a.ld_emb;
    const uint16_t* el = a.char_emb_lo ? (const uint16_t*)a.char_emb_lo + (size_t)ch * a.ld_emb : nullptr;
    for (int e = lane; e < a.E; e += 64) {
      xrow[a.col_emb + e] = eh[e];
      if (xrow_lo) xrow_lo[a.col_emb + e] = el ? el[e] : (uint16_t)0;
    }
  } else {
    for (int e = lane; e < a.E; e += 64) {
      xrow[a.col_emb + e] = 0;
      if (xrow_lo) xrow_lo[a.col_emb + e] = 0;
    }
  }
  if (!a.attend) return;

  // ---- energies: e[t] = v . tanh(E_enc[b,t,:] + sum over the taps inside the word of Hq[tap,:])
  const int j0 = lane, j1 = lane + 64;
  float hq0[ATT_MAX_TAPS], hq1[ATT_MAX_TAPS];
#pragma unroll
  for (int k = 0; k < ATT_MAX_TAPS; ++k) {
    hq0[k] = (k < a.taps && j0 < a.P) ? q[k * a.ld_tap + j0] : 0.0f;
    hq1[k] = (k < a.taps && j1 < a.P) ? q[k * a.ld_tap + j1] : 0.0f;
  }
  const float v0 = j0 < a.P ? a.v[j0] : 0.0f, v1 = j1 < a.P ? a.v[j1] : 0.0f;
  const int half = a.taps / 2;
  const float* ee = a.e_enc + (size_t)b * a.S * a.ld_e;
  // four positions per iteration: their loads and wave reductions are independent and overlap
  for (int t0 = 0; t0 < a.S; t0 += 4) {
    float part[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u;
      const bool on = t < a.S;
      float p0 = (on && j0 < a.P) ? ee[(size_t)t * a.ld_e + j0] : 0.0f;
      float p1 = (on && j1 < a.P) ? ee[(size_t)t * a.ld_e + j1] : 0.0f;
#pragma unroll
      for (int k = 0; k < ATT_MAX_TAPS; ++k) {
        const int src = t + k - half;
        if (k < a.taps && src >= 0 && src < a.S) { p0 += hq0[k]; p1 += hq1[k]; }
      }
      part[u] = v0 * tanhf_(p0) + v1 * tanhf_(p1);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) part[u] += __shfl_xor(part[u], o, 64);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (t0 + u < a.S) se[wave][t0 + u] = part[u];  // every lane holds the sum; same value to the same address
  }
  // ---- soft-max over all S positions (pads included, as the reference does)
  float m = -INFINITY;
  for (int t = lane; t < a.S; t += 64) m = fmaxf(m, se[wave][t]);
  m = wave_max(m);
  float z = 0.0f;
  for (int t = lane; t < a.S; t += 64) {
    const float e = __expf(se[wave][t] - m);
    se[wave][t] = e;
    z += e;
  }
  z = 1.0f / wave_sum(z);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the exp() values written by other lanes
  __builtin_amdgcn_wave_barrier();
  // ---- context = att @ encoder_output  -> x[b, col_ctx .. col_ctx + C)
  const int c0 = lane * 8;
  if (c0 < a.C) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint16_t* eh = (const uint16_t*)a.enc + (size_t)b * a.S * a.ld_enc + c0;
    const uint16_t* el = a.enc_lo ? (const uint16_t*)a.enc_lo + (size_t)b * a.S * a.ld_enc + c0 : nullptr;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    for (int t0 = 0; t0 < a.S; t0 += 4) {   // 4 rows (8 16-byte loads) in flight per lane
      uint4 h[4], l[4];
      float w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u;
        const bool on = t < a.S;
        w[u] = on ? se[wave][t] * z : 0.0f;
        h[u] = on ? *(const uint4*)(eh + (size_t)t * a.ld_enc) : zero4;
        l[u] = (on && el) ? *(const uint4*)(el + (size_t)t * a.ld_enc) : zero4;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t hw[4] = {h[u].x, h[u].y, h[u].z, h[u].w};
        const uint32_t lw[4] = {l[u].x, l[u].y, l[u].z, l[u].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[2 * i] += w[u] * (__uint_as_float(hw[i] << 16) + __uint_as_float(lw[i] << 16));
          acc[2 * i + 1] += w[u] * (__uint_as_float(hw[i] & 0xffff0000u) + __uint_as_float(lw[i] & 0xffff0000u));
        }
      }
    }
    uint32_t ph[4], pl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint16_t h0, l0, h1, l1;
      split_bf16(acc[2 * i], h0, l0);
      split_bf16(acc[2 * i + 1], h1, l1);
      ph[i] = (uint32_t)h0 | ((uint32_t)h1 << 16);
      pl[i] = (uint32_t)l0 | ((uint32_t)l1 << 16);
    }
    *(uint4*)(xrow + a.col_ctx + c0) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
    if (xrow_lo) *(uint4*)(xrow_lo + a.col_ctx + c0) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
  }
}

__global__ __launch_bounds__(256) void lstm_cell_kernel(const float* __restrict__ gates, int ld_g, int H, float* __restrict__ c,
                                                        int B, uint16_t* h, uint16_t* h_lo, int ld_h, int col_h,
                                                        uint16_t* c16, uint16_t* c16_lo, int col_c) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, u = idx - b * H;
  const float* g = gates + (size_t)b * ld_g;
  float cn, hn;
  lstm_cell_update(g[u], g[H + u], g[2 * H + u], g[3 * H + u], c[idx], cn, hn);
  c[idx] = cn;
  uint16_t hi, lo;
  split_bf16(hn, hi, lo);
  h[(size_t)b * ld_h + col_h + u] = hi;
  if (h_lo) h_lo[(size_t)b * ld_h + col_h + u] = lo;
  if (c16) {
    split_bf16(cn, hi, lo);
    c16[(size_t)b * ld_h + col_c + u] = hi;
    if (c16_lo) c16_lo[(size_t)b * ld_h + col_c + u] = lo;
  }
}

}  // namespace

extern "C" int cube_lemma_attend(const cube_lemma_attend_args* a, void* stream) {
  CUBE_ENTRY();
  if (!a) CUBE_FAIL(CUBE_ERR_ARG, "cube_lemma_attend: null args");
  if (a->B <= 0) return CUBE_OK;
  if (!a->q || !a->x || a->ld_q <= 0 || a->ld_x <= 0 || a->E < 0 || a->col_emb < 0 || a->col_emb + a->E > a->ld_x)
    CUBE_FAIL(CUBE_ERR_ARG, "cube_lemma_attend: bad q / x");
  if (a->step < 0) CUBE_FAIL(CUBE_ERR_ARG, "cube_lemma_attend: step < 0");
  if (a->step > 0) {
    if (!a->out_char || !a->out_case || !a->done || !a->n_done || !a->char_emb || a->step > a->ld_out)
      CUBE_FAIL(CUBE_ERR_ARG, "cube_lemma_attend: step %d needs out_char/out_case [B, ld_out >= step], done, n_done, char_emb", a->step);
    if (a->n_char <= 0 || a->n_case <= 0 || a->col_char < 0 || a->col_case < 0 || a->col_char + a->n_char > a->ld_q ||
        a->col_case + a->n_case > a->ld_q || a->ld_emb < a->E)
      CUBE_FAIL(CUBE_ERR_ARG, "cube_lemma_attend: logits columns outside q / embedding table too narrow");
  }
  if (a->attend) {
    if (!a->v || !a->e_enc || !a->enc) CUBE_FAIL(CUBE_ERR_ARG, "cube_lemma_attend: null v / e_enc / enc");
    if (a->S <= 0 || a->S > ATT_MAX_S) CUBE_FAIL(CUBE_ERR_ARG, "cube_lemma_attend: S=%d outside 1..%d", a->S, ATT_MAX_S);
    if (a->P <= 0 || a->P > ATT_MAX_P || a->taps <= 0 || a->taps > ATT_MAX_TAPS || !(a->taps & 1) || a->ld_tap < a->P ||
        a->taps * a->ld_tap > a->ld_q || a->ld_e < a->P)
      CUBE_FAIL(CUBE_ERR_ARG, "cube_lemma_attend: P=%d (<= %d), taps=%d (odd, <= %d), ld_tap=%d, ld_e=%d", a->P, ATT_MAX_P, a->taps,
                ATT_MAX_TAPS, a->ld_tap, a->ld_e);
    if (a->C <= 0 || a->C > ATT_MAX_C || a->C % 8 || a->ld_enc % 8 || a->ld_enc < a->C || a->col_ctx % 8 || a->ld_x % 8 ||
        a->col_ctx < 0 || a->col_ctx + a->C > a->ld_x)
      CUBE_FAIL(CUBE_ERR_ARG, "cube_lemma_attend: C=%d (multiple of 8, <= %d), ld_enc=%d, col_ctx=%d, ld_x=%d", a->C, ATT_MAX_C,
                a->ld_enc, a->col_ctx, a->ld_x);
    if (((uintptr_t)a->enc | (uintptr_t)a->enc_lo | (uintptr_t)a->x | (uintptr_t)a->x_lo) & 15)
      CUBE_FAIL(CUBE_ERR_ALIGN, "cube_lemma_attend: enc / x not 16-byte aligned");
  }
  hipLaunchKernelGGL(lemma_attend_kernel, dim3((a->B + 3) / 4), dim3(256), 0, (hipStream_t)stream, *a);
  CUBE_CHECK_LAUNCH("cube_lemma_attend");
  return CUBE_OK;
}

extern "C" int cube_lstm_cell(const float* gates, int ld_g, int H, float* c, int B, void* h, void* h_lo, int ld_h, int col_h,
                              void* c16, void* c16_lo, int col_c, void* stream) {
  CUBE_ENTRY();
  if (B <= 0) return CUBE_OK;
  if (!gates || !c || !h || H <= 0 || ld_g < 4 * H || col_h < 0 || col_h + H > ld_h || (c16 && (col_c < 0 || col_c + H > ld_h)))
    CUBE_FAIL(CUBE_ERR_ARG, "cube_lstm_cell: bad args (H=%d ld_g=%d ld_h=%d col_h=%d col_c=%d)", H, ld_g, ld_h, col_h, col_c);
  if ((long long)B * H > 0x7fffffffLL) CUBE_FAIL(CUBE_ERR_ARG, "cube_lstm_cell: B*H too large");
  hipLaunchKernelGGL(lstm_cell_kernel, dim3((B * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, gates, ld_g, H, c, B,
                     (uint16_t*)h, (uint16_t*)h_lo, ld_h, col_h, (uint16_t*)c16, (uint16_t*)c16_lo, col_c);
  CUBE_CHECK_LAUNCH("cube_lstm_cell");
  return CUBE_OK;
}

// Fused bf16 MFMA GEMM / implicit Conv1d for gfx950.
//
// One kernel serves every dense contraction on the parser path: the gated Conv1d(k=5) stacks
// (5 shifted K-slices of the same activation rows, no im2col), the Linear layers and the LSTM
// input projections.  128x128 block tile, 4 waves (2x2), each wave 64x64 = 2x2 tiles of
// v_mfma_f32_32x32x16_bf16; BK = 32; A and B tiles register-staged into a double-buffered,
// padded (80-byte rows: conflict-free ds_read_b128) LDS image; one barrier per K-step.
//
// Precision: operands are bf16, accumulation fp32.  In SPLIT mode ("bf16x3") every operand
// is a pair of bf16 planes, x ~= hi + lo (16 significant bits), and each product is formed as
// hi*hi + hi*lo + lo*hi on the matrix cores — fp32-grade results (what the fp32 reference
// computes, to ~1e-5) at a third of the bf16 MFMA rate, still 5x the fp32 MFMA rate.
//
// Epilogues are fused: bias, tanh / relu, the tanh*sigmoid gate with residual accumulation,
// row scatter (unpack), hi/lo splitting of bf16 outputs.
#include "common.h"
#include "../../include/cube_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int ROWB = 80;                      // bytes per LDS row: 64 data + 16 pad
constexpr int PLANE_BYTES = BM * ROWB;        // one 128-row operand plane
template <bool SPLIT> constexpr int stage_bytes() { return (SPLIT ? 4 : 2) * PLANE_BYTES; }

struct GemmParams {
  const uint16_t* A; const uint16_t* A_lo; int lda; int M; int seq_len; int taps; int Kc;
  const uint16_t* W; const uint16_t* W_lo; int N; const float* bias;
  int n_valid; int accumulate;
  void* out0; uint16_t* out0_lo; int ldo0; int col0;
  const int32_t* rowmap;
  float* res; int ld_res; int res_mode;
  uint16_t* out1; uint16_t* out1_lo; int ldo1; int col1;
  uint16_t* out2; uint16_t* out2_lo; int ldo2; int col2;
};

__device__ __forceinline__ void store_bf16(uint16_t* hi, uint16_t* lo, size_t idx, float v) {
  if (lo) {
    uint16_t h, l;
    split_bf16(v, h, l);
    hi[idx] = h;
    lo[idx] = l;
  } else {
    hi[idx] = f32_to_bf16(v);
  }
}

template <int EPI, bool SPLIT>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = stage_bytes<SPLIT>();
  constexpr int NP = SPLIT ? 2 : 1;  // planes per operand
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int m0 = blockIdx.y * BM;
  const int n0 = blockIdx.x * BN;
  const int ksteps_per_tap = p.Kc / BK;
  const int nk = p.taps * ksteps_per_tap;
  const int ktot = p.taps * p.Kc;
  const int half_taps = p.taps >> 1;

  // staging: 512 16-byte chunks per operand plane, 2 per thread: chunk = tid + 256*i
  int a_pos[2];
  size_t a_off[2], b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int chunk = tid + 256 * i;
    const int row = chunk >> 2;
    const int kc8 = (chunk & 3) * 8;
    const int r = m0 + row;
    a_pos[i] = (r < p.M) ? (r % p.seq_len) : -100000;
    a_off[i] = (size_t)r * p.lda + kc8;
    b_off[i] = (size_t)(n0 + row) * ktot + kc8;
  }
  const int st_off0 = (tid >> 2) * ROWB + (tid & 3) * 16;          // chunk tid
  const int st_off1 = ((tid + 256) >> 2) * ROWB + (tid & 3) * 16;  // chunk tid+256

  uint4 ra[NP][2], rb[NP][2];
  auto load_step = [&](int s) {
    const int tap = s / ksteps_per_tap;
    const int kc0 = (s - tap * ksteps_per_tap) * BK;
    const int d = tap - half_taps;
    const long a_delta = (long)d * p.lda + kc0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pp = a_pos[i] + d;
      const bool ok = (pp >= 0 && pp < p.seq_len);
      ra[0][i] = ok ? *reinterpret_cast<const uint4*>(p.A + a_off[i] + a_delta) : make_uint4(0, 0, 0, 0);
      rb[0][i] = *reinterpret_cast<const uint4*>(p.W + b_off[i] + (size_t)s * BK);
      if constexpr (SPLIT) {
        ra[1][i] = ok ? *reinterpret_cast<const uint4*>(p.A_lo + a_off[i] + a_delta) : make_uint4(0, 0, 0, 0);
        rb[1][i] = *reinterpret_cast<const uint4*>(p.W_lo + b_off[i] + (size_t)s * BK);
      }
    }
  };
  // stage layout: [A hi][A lo][B hi][B lo] (SPLIT) or [A][B]
  auto store_step = [&](int buf) {
    unsigned char* base = smem + buf * STAGE;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      *reinterpret_cast<uint4*>(base + pl * PLANE_BYTES + st_off0) = ra[pl][0];
      *reinterpret_cast<uint4*>(base + pl * PLANE_BYTES + st_off1) = ra[pl][1];
      *reinterpret_cast<uint4*>(base + (NP + pl) * PLANE_BYTES + st_off0) = rb[pl][0];
      *reinterpret_cast<uint4*>(base + (NP + pl) * PLANE_BYTES + st_off1) = rb[pl][1];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_step(0);
  store_step(0);
  __syncthreads();

  const int l31 = lane & 31, lh = lane >> 5;
  const int a_rd = (wr * 64 + l31) * ROWB + lh * 16;
  const int b_rd = NP * PLANE_BYTES + (wc * 64 + l31) * ROWB + lh * 16;

  for (int s = 0; s < nk; ++s) {
    const int buf = s & 1;
    if (s + 1 < nk) load_step(s + 1);
    const unsigned char* base = smem + buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[NP][2], fb[NP][2];
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[pl][i] = *reinterpret_cast<const bf16x8*>(base + pl * PLANE_BYTES + a_rd + i * 32 * ROWB + ks * 32);
          fb[pl][i] = *reinterpret_cast<const bf16x8*>(base + pl * PLANE_BYTES + b_rd + i * 32 * ROWB + ks * 32);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (SPLIT) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
        }
    }
    if (s + 1 < nk) store_step(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue.  C/D map of 32x32x16: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  if constexpr (EPI == CUBE_EPI_GATE) {
    const int ch = blockIdx.x * 64 + wc * 32 + l31;
    const float bias_a = p.bias[n0 + wc * 64 + l31];
    const float bias_b = p.bias[n0 + wc * 64 + 32 + l31];
    uint16_t* out0 = reinterpret_cast<uint16_t*>(p.out0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < p.M && ch < p.n_valid) {
          const float g = tanhf_(acc[i][0][r] + bias_a) * sigmoidf_(acc[i][1][r] + bias_b);
          float* rp = p.res + (size_t)row * p.ld_res + ch;
          float rs;
          if (p.res_mode == 0) rs = g;
          else if (p.res_mode == 1) rs = *rp + g;
          else rs = *rp + 2.0f * g;
          *rp = rs;
          if (out0) store_bf16(out0, p.out0_lo, (size_t)row * p.ldo0 + p.col0 + ch, g);
          if (p.out1) store_bf16(p.out1, p.out1_lo, (size_t)row * p.ldo1 + p.col1 + ch, rs + g);
          if (p.out2) store_bf16(p.out2, p.out2_lo, (size_t)row * p.ldo2 + p.col2 + ch, rs + g);
        }
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wc * 64 + j * 32 + l31;
      if (col >= p.n_valid) continue;
      const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row >= p.M) continue;
          int orow = row;
          if (p.rowmap) {
            orow = p.rowmap[row];
            if (orow < 0) continue;
          }
          float v = acc[i][j][r] + bias;
          if constexpr (EPI == CUBE_EPI_F32) {
            float* o = reinterpret_cast<float*>(p.out0) + (size_t)orow * p.ldo0 + p.col0 + col;
            *o = p.accumulate ? (*o + v) : v;
          } else {
            if constexpr (EPI == CUBE_EPI_BF16_TANH) v = tanhf_(v);
            if constexpr (EPI == CUBE_EPI_BF16_RELU) v = fmaxf(v, 0.f);
            store_bf16(reinterpret_cast<uint16_t*>(p.out0), p.out0_lo, (size_t)orow * p.ldo0 + p.col0 + col, v);
          }
        }
      }
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int EPI>
int launch(const GemmParams& p, dim3 grid, hipStream_t st) {
  if (p.A_lo) {
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)gemm_kernel<EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              2 * stage_bytes<true>()) != hipSuccess)
        CUBE_FAIL(CUBE_ERR_LAUNCH, "cube_gemm_bf16: cannot reserve %d bytes of LDS", 2 * stage_bytes<true>());
      attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<EPI, true>), grid, dim3(256), 2 * stage_bytes<true>(), st, p);
  } else {
    hipLaunchKernelGGL((gemm_kernel<EPI, false>), grid, dim3(256), 2 * stage_bytes<false>(), st, p);
  }
  return CUBE_OK;
}

}  // namespace

extern "C" int cube_gemm_bf16(const cube_gemm_args* a, void* stream) {
  if (!a) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: null args");
  if (a->M <= 0) return CUBE_OK;
  if (!a->A || !a->W) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: null operand");
  if ((a->A_lo == nullptr) != (a->W_lo == nullptr))
    CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: A_lo and W_lo must be given together (bf16x3) or not at all (bf16)");
  if (a->taps != 1 && a->taps != 3 && a->taps != 5) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: taps=%d", a->taps);
  if (a->Kc <= 0 || a->Kc % BK) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: Kc=%d not a multiple of %d", a->Kc, BK);
  if (a->N <= 0 || a->N % BN) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: N=%d not a multiple of %d", a->N, BN);
  if (a->lda < a->Kc || a->lda % 8) CUBE_FAIL(CUBE_ERR_ALIGN, "cube_gemm_bf16: lda=%d (Kc=%d)", a->lda, a->Kc);
  if (!aligned16(a->A) || !aligned16(a->W) || !aligned16(a->A_lo) || !aligned16(a->W_lo))
    CUBE_FAIL(CUBE_ERR_ALIGN, "cube_gemm_bf16: operands must be 16-byte aligned");
  const int seq = a->taps == 1 ? a->M : a->seq_len;
  if (seq <= 0 || a->M % seq) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: M=%d not a multiple of seq_len=%d", a->M, seq);
  GemmParams p;
  p.A = (const uint16_t*)a->A; p.A_lo = (const uint16_t*)a->A_lo; p.lda = a->lda; p.M = a->M; p.seq_len = seq;
  p.taps = a->taps; p.Kc = a->Kc;
  p.W = (const uint16_t*)a->W; p.W_lo = (const uint16_t*)a->W_lo; p.N = a->N; p.bias = a->bias;
  p.n_valid = a->n_valid > 0 ? a->n_valid : (a->epilogue == CUBE_EPI_GATE ? a->N / 2 : a->N);
  p.accumulate = a->accumulate;
  p.out0 = a->out0; p.out0_lo = (uint16_t*)a->out0_lo; p.ldo0 = a->ldo0; p.col0 = a->col0; p.rowmap = a->rowmap;
  p.res = a->res; p.ld_res = a->ld_res; p.res_mode = a->res_mode;
  p.out1 = (uint16_t*)a->out1; p.out1_lo = (uint16_t*)a->out1_lo; p.ldo1 = a->ldo1; p.col1 = a->col1;
  p.out2 = (uint16_t*)a->out2; p.out2_lo = (uint16_t*)a->out2_lo; p.ldo2 = a->ldo2; p.col2 = a->col2;
  if (p.n_valid > a->N) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: n_valid=%d > N=%d", p.n_valid, a->N);
  dim3 grid(a->N / BN, (a->M + BM - 1) / BM);
  hipStream_t st = (hipStream_t)stream;
  int rc = CUBE_OK;
  switch (a->epilogue) {
    case CUBE_EPI_F32:
      if (!a->out0 || a->ldo0 < a->col0 + p.n_valid) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: bad out0/ldo0");
      rc = launch<CUBE_EPI_F32>(p, grid, st);
      break;
    case CUBE_EPI_BF16:
    case CUBE_EPI_BF16_TANH:
    case CUBE_EPI_BF16_RELU:
      if (!a->out0 || a->ldo0 < a->col0 + p.n_valid) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: bad out0/ldo0");
      if (a->epilogue == CUBE_EPI_BF16) rc = launch<CUBE_EPI_BF16>(p, grid, st);
      else if (a->epilogue == CUBE_EPI_BF16_TANH) rc = launch<CUBE_EPI_BF16_TANH>(p, grid, st);
      else rc = launch<CUBE_EPI_BF16_RELU>(p, grid, st);
      break;
    case CUBE_EPI_GATE: {
      const int half = p.n_valid;  // valid channels; N/2 >= half when padded to the 128-column tile
      if (half > a->N / 2) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: gate n_valid=%d > N/2", half);
      if (!a->res || a->ld_res < half) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: gate epilogue needs res[M, >=%d]", half);
      if (a->res_mode < 0 || a->res_mode > 2) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: res_mode=%d", a->res_mode);
      if (a->rowmap) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: rowmap is not supported by the gate epilogue");
      if (a->out0 && a->ldo0 < a->col0 + half) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: bad out0/ldo0");
      if (a->out1 && a->ldo1 < a->col1 + half) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: bad out1/ldo1");
      if (a->out2 && a->ldo2 < a->col2 + half) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: bad out2/ldo2");
      if (!a->bias) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: gate epilogue needs a bias");
      rc = launch<CUBE_EPI_GATE>(p, grid, st);
      break;
    }
    default:
      CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: unknown epilogue %d", a->epilogue);
  }
  if (rc != CUBE_OK) return rc;
  CUBE_CHECK_LAUNCH("cube_gemm_bf16");
  return CUBE_OK;
}

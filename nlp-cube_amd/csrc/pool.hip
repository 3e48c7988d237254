// Sub-word piece -> word pooling of the external-embedding stage (gfx950): the arithmetic of
// LMHelperHF._compute_we (cube/networks/lm.py:183-197) on hidden states that are already on the device, written
// straight into the K per-layer word-embedding matrices the parser takes (lm.py:199-208 cuts the 13 x 768 vector
// into 13 layers).  HBM-bound: every hidden-state value of a word's pieces is read once, every output value
// written once.
//
// The reference's loop, restated exactly (n = pieces of the word, h[j] = the K*E hidden values of piece j):
//     m = we[piece 0]                  -- a VIEW of the buffer row, not a copy
//     for z in 0 .. n-2: m += we[piece z]        -- z = 0 adds the row to itself IN PLACE
//     m = m / n
// i.e. n = 1: h[0];  n >= 2: ((h[0] + h[0]) + h[1] + ... + h[n-2]) / n -- the first piece counts twice, the last
// one is dropped; a word without pieces gets zeros.  fp32 throughout, additions in that order, one correctly rounded
// division: the kernel's results are BIT-EQUAL to NumPy's.  (The in-place doubling of the buffer row has no other
// reader: a piece belongs to one word.)
#include "common.h"
#include "../../include/cube_hip.h"

namespace {

struct PoolLayers {
  const float* h[CUBE_POOL_MAX_LAYERS];
};

// one workgroup of 64 * VEC-float lanes per (word, layer chunk); a lane owns four consecutive values
__global__ __launch_bounds__(256) void subword_pool_kernel(PoolLayers layers, size_t stride_s, size_t stride_p, int E,
                                                           const int32_t* __restrict__ word_sent,
                                                           const int32_t* __restrict__ word_pos,
                                                           const int32_t* __restrict__ word_n, int n_words,
                                                           float* __restrict__ out) {
  const int w = blockIdx.x, k = blockIdx.y;
  const int n = word_n[w];
  float* o = out + ((size_t)k * n_words + w) * E;
  const float* h = layers.h[k] + (size_t)word_sent[w] * stride_s + (size_t)word_pos[w] * stride_p;
  for (int e = threadIdx.x * 4; e < E; e += blockDim.x * 4) {
    float4 m = {0.f, 0.f, 0.f, 0.f};
    if (n > 0) {
      m = *reinterpret_cast<const float4*>(h + e);
      if (n > 1) {
        m.x += m.x; m.y += m.y; m.z += m.z; m.w += m.w;   // z = 0: the row added to itself
        for (int z = 1; z < n - 1; ++z) {
          const float4 v = *reinterpret_cast<const float4*>(h + (size_t)z * stride_p + e);
          m.x += v.x; m.y += v.y; m.z += v.z; m.w += v.w;
        }
        const float d = (float)n;
        m.x = __fdiv_rn(m.x, d); m.y = __fdiv_rn(m.y, d); m.z = __fdiv_rn(m.z, d); m.w = __fdiv_rn(m.w, d);
      }
    }
    *reinterpret_cast<float4*>(o + e) = m;
  }
}

}  // namespace

extern "C" int cube_subword_pool(const void* const* hidden, int n_layers, size_t stride_s, size_t stride_p, int E,
                                 const int32_t* word_sent, const int32_t* word_pos, const int32_t* word_n, int n_words,
                                 float* out, void* stream) {
  CUBE_ENTRY();
  if (n_words <= 0 || n_layers <= 0) return CUBE_OK;
  if (!hidden || !word_sent || !word_pos || !word_n || !out) CUBE_FAIL(CUBE_ERR_ARG, "cube_subword_pool: null pointer");
  if (n_layers > CUBE_POOL_MAX_LAYERS) CUBE_FAIL(CUBE_ERR_ARG, "cube_subword_pool: %d layers (at most %d)", n_layers, CUBE_POOL_MAX_LAYERS);
  if (E <= 0 || E % 4 || stride_p % 4 || stride_s % 4 || stride_p < (size_t)E)
    CUBE_FAIL(CUBE_ERR_ALIGN, "cube_subword_pool: E, stride_p and stride_s must be multiples of 4 and stride_p >= E");
  if (n_words > 65535 * 1024) CUBE_FAIL(CUBE_ERR_ARG, "cube_subword_pool: too many words");
  PoolLayers L;
  for (int k = 0; k < n_layers; ++k) {
    if (!hidden[k] || (uintptr_t)hidden[k] % 16) CUBE_FAIL(CUBE_ERR_ALIGN, "cube_subword_pool: layer %d is null or not 16-byte aligned", k);
    L.h[k] = (const float*)hidden[k];
  }
  if ((uintptr_t)out % 16) CUBE_FAIL(CUBE_ERR_ALIGN, "cube_subword_pool: out must be 16-byte aligned");
  const int threads = E >= 1024 ? 256 : (E >= 512 ? 128 : 64);
  hipLaunchKernelGGL(subword_pool_kernel, dim3((unsigned)n_words, (unsigned)n_layers), dim3(threads), 0, (hipStream_t)stream, L,
                     stride_s, stride_p, E, word_sent, word_pos, word_n, n_words, out);
  CUBE_CHECK_LAUNCH("cube_subword_pool");
  return CUBE_OK;
}

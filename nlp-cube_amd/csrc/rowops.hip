// Row-wise HBM-bound helpers of the parser path: embedding gathers into concat buffers,
// the WordGram masked mean, the external-embedding finish and arg-max over logits.
// All are coalesced along the row (16-byte vector accesses where the layout allows).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"
#include "../../include/cube_hip.h"

static thread_local char g_err[512] = "";
void cube_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* cube_last_error(void) { return g_err; }
extern "C" int cube_abi_version(void) { return 19; }

namespace {

// ---- gather: one wave per row, both planes in one launch; 2-byte fallback when not 16-byte alignable
__global__ __launch_bounds__(256) void gather_rows_kernel(uint16_t* dst, uint16_t* dst_lo, int ld_dst, int col0, int width,
                                                          const uint16_t* table, const uint16_t* table_lo, int ld_table,
                                                          const int32_t* idx, const int32_t* dst_rows, int n,
                                                          int vec_ok) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int i = wave; i < n; i += nwaves) {
    const int src_row = idx ? idx[i] : 0;
    const int dst_row = dst_rows ? dst_rows[i] : i;
    if (dst_row < 0) continue;
    const size_t so = (size_t)src_row * ld_table, dof = (size_t)dst_row * ld_dst + col0;
    if (vec_ok) {
      for (int c = lane * 8; c < width; c += 64 * 8) {
        *reinterpret_cast<uint4*>(dst + dof + c) = *reinterpret_cast<const uint4*>(table + so + c);
        if (dst_lo) *reinterpret_cast<uint4*>(dst_lo + dof + c) = *reinterpret_cast<const uint4*>(table_lo + so + c);
      }
    } else {
      for (int c = lane; c < width; c += 64) {
        dst[dof + c] = table[so + c];
        if (dst_lo) dst_lo[dof + c] = table_lo[so + c];
      }
    }
  }
}

// ---- masked mean over the W character positions of each word, summed over the gated layers
struct MeanLayers {
  const uint16_t* hi[CUBE_MAX_MEAN_LAYERS];
  const uint16_t* lo[CUBE_MAX_MEAN_LAYERS];
  float weight[CUBE_MAX_MEAN_LAYERS];
  int n;
};
__global__ __launch_bounds__(256) void masked_mean_kernel(MeanLayers ly, int ld_x, const int32_t* len, int N, int W,
                                                          int C, uint16_t* out, uint16_t* out_lo, int ld_out,
                                                          const int32_t* word_rows) {
  const int n = blockIdx.x;
  if (n >= N) return;
  const int l = min(max(len[n], 0), W);
  const float den = (float)max(len[n], 1);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < ly.n; ++k) {
      const size_t base = (size_t)(word_rows ? word_rows[(size_t)k * N + n] : n * W) * ld_x + c;
      float sk = 0.f;
      for (int q = 0; q < l; ++q) {
        float v = bf16_to_f32(ly.hi[k][base + (size_t)q * ld_x]);
        if (ly.lo[k]) v += bf16_to_f32(ly.lo[k][base + (size_t)q * ld_x]);
        sk += v;
      }
      s += ly.weight[k] * sk;
    }
    const float v = s / den;
    if (out_lo) {
      uint16_t h, lo;
      split_bf16(v, h, lo);
      out[(size_t)n * ld_out + c] = h;
      out_lo[(size_t)n * ld_out + c] = lo;
    } else {
      out[(size_t)n * ld_out + c] = f32_to_bf16(v);
    }
  }
}

// 8 channels (16 bytes) per thread: the loads of one word row are 512-byte contiguous runs
__global__ __launch_bounds__(256) void masked_mean_vec_kernel(MeanLayers ly, int ld_x, const int32_t* len, int N, int W,
                                                              int C, uint16_t* out, uint16_t* out_lo, int ld_out,
                                                              const int32_t* word_rows) {
  const int nchunk = C >> 3;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * nchunk) return;
  const int n = t / nchunk, c = (t - n * nchunk) << 3;
  const int l = min(max(len[n], 0), W);
  const float den = (float)max(len[n], 1);
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  for (int k = 0; k < ly.n; ++k) {
    const size_t base = (size_t)(word_rows ? word_rows[(size_t)k * N + n] : n * W) * ld_x + c;
    float sk[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sk[j] = 0.f;
    const uint16_t* hi = ly.hi[k] + base;
    const uint16_t* lo = ly.lo[k] ? ly.lo[k] + base : nullptr;
    // four positions (eight 16-byte loads) in flight per thread; summed in position order, hi then lo
    for (int q0 = 0; q0 < l; q0 += 4) {
      uint4 v[4], u[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = min(q0 + i, l - 1);
        v[i] = *reinterpret_cast<const uint4*>(hi + (size_t)q * ld_x);
        if (lo) u[i] = *reinterpret_cast<const uint4*>(lo + (size_t)q * ld_x);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (q0 + i >= l) break;
        const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sk[2 * j] += __uint_as_float(w[j] << 16);
          sk[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
        }
        if (lo) {
          const uint32_t x[4] = {u[i].x, u[i].y, u[i].z, u[i].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            sk[2 * j] += __uint_as_float(x[j] << 16);
            sk[2 * j + 1] += __uint_as_float(x[j] & 0xffff0000u);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] += ly.weight[k] * sk[j];
  }
  uint16_t h[8], lw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = s[j] / den;
    if (out_lo) split_bf16(v, h[j], lw[j]); else h[j] = f32_to_bf16(v);
  }
  const size_t o = (size_t)n * ld_out + c;
  *reinterpret_cast<uint4*>(out + o) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
  if (out_lo)
    *reinterpret_cast<uint4*>(out_lo + o) =
        make_uint4(lw[0] | (lw[1] << 16), lw[2] | (lw[3] << 16), lw[4] | (lw[5] << 16), lw[6] | (lw[7] << 16));
}

// int8x2 layers (cube_hip.h): 16 channels (16 bytes per plane) of one word in one LAYER per lane -- eight adjacent lanes take
// the (up to eight) layers of a (word, channel chunk) and the layer terms are added in layer order by lane shuffles, exactly
// the order of the loop they replace (one thread per (word, chunk) walking the layers: 131k threads for 8192 words, a
// quarter of the chip's wave slots, each a chain of dependent loads: 37 us; now .. us).  The sum over the positions of a word
// is exact in int32.
__global__ __launch_bounds__(256) void masked_mean_i8_kernel(MeanLayers ly, int ld_x, const int32_t* len, int N, int W,
                                                             int C, uint16_t* out, uint16_t* out_lo, int ld_out,
                                                             const int32_t* word_rows) {
  const int nchunk = C >> 4;
  const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;  // (word, chunk); whole groups of eight lanes leave together
  const int k = threadIdx.x & 7;                                // layer
  if (t >= N * nchunk) return;
  const int n = t / nchunk, c = (t - n * nchunk) << 4;
  const int l = min(max(len[n], 0), W);
  const float den = (float)max(len[n], 1);
  float term[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) term[j] = 0.f;
  if (k < ly.n) {
    const size_t base = (size_t)(word_rows ? word_rows[(size_t)k * N + n] : n * W) * ld_x + c;  // bytes
    const unsigned char* hi = reinterpret_cast<const unsigned char*>(ly.hi[k]) + base;
    const unsigned char* lo = reinterpret_cast<const unsigned char*>(ly.lo[k]) + base;
    int sk[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) sk[j] = 0;
    for (int q0 = 0; q0 < l; q0 += 4) {
      uint4 v[4], u[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = min(q0 + i, l - 1);
        v[i] = *reinterpret_cast<const uint4*>(hi + (size_t)q * ld_x);
        u[i] = *reinterpret_cast<const uint4*>(lo + (size_t)q * ld_x);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (q0 + i >= l) break;
        const uint32_t wh[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        const uint32_t wl[4] = {u[i].x, u[i].y, u[i].z, u[i].w};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int hq = (int)(int8_t)(wh[j >> 2] >> (8 * (j & 3)));
          const int lq = (int)(int8_t)(wl[j >> 2] >> (8 * (j & 3)));
          sk[j] += (hq << CUBE_I8_SHIFT) + lq;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) term[j] = (float)sk[j] * (1.0f / (float)CUBE_I8_ONE);
  }
  // s = fma(w_k, term_k, s) for k = 0 .. n-1 in layer order: every lane of the group runs the same chain on the shuffled terms
  float s[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) s[j] = 0.f;
  const int lane0 = (threadIdx.x & 63) & ~7;
  for (int kk = 0; kk < ly.n; ++kk) {
    const float wk = ly.weight[kk];
#pragma unroll
    for (int j = 0; j < 16; ++j) s[j] = fmaf(wk, __shfl(term[j], lane0 + kk), s[j]);
  }
  if (k >= 2) return;  // lanes 0 and 1 of the group store 8 channels each
  uint16_t h[8], lw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = (k ? s[8 + j] : s[j]) / den;
    if (out_lo) split_bf16(v, h[j], lw[j]); else h[j] = f32_to_bf16(v);
  }
  const size_t o = (size_t)n * ld_out + c + 8 * k;
  *reinterpret_cast<uint4*>(out + o) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
  if (out_lo)
    *reinterpret_cast<uint4*>(out_lo + o) =
        make_uint4(lw[0] | (lw[1] << 16), lw[2] | (lw[3] << 16), lw[4] | (lw[5] << 16), lw[6] | (lw[7] << 16));
}

__global__ __launch_bounds__(256) void ext_finish_kernel(const float* acc, int ld_acc, int B, int L, int root, int P, float K, int parts,
                                                         int part_stride,
                                                         uint16_t* da, uint16_t* da_lo, int ld_a, int col_a,
                                                         uint16_t* db, uint16_t* db_lo, int ld_b, int col_b) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * L * P) return;
  const int r = t / P, c = t - r * P;
  const int b = r / L, i = r - b * L;
  float sum = acc[(size_t)r * ld_acc + c];
  for (int j = 1; j < parts; ++j) sum += acc[(size_t)r * ld_acc + (size_t)j * part_stride + c];  // in the order of the LM layers
  const float e = tanhf_(sum / K);
  uint16_t v, vl;
  split_bf16(e, v, vl);
  const size_t orow = (size_t)b * (L + root) + i + root;
  if (da) da[orow * ld_a + col_a + c] = v;
  if (db) db[orow * ld_b + col_b + c] = v;
  if (da_lo) da_lo[orow * ld_a + col_a + c] = vl;
  if (db_lo) db_lo[orow * ld_b + col_b + c] = vl;
}

// ---- first gated character layer as table look-ups (see cube_char_conv0_lookup in the header)
// first output row past each sequence group, computed by the host: the kernel's row -> group search is then compares only (a
// loop with an integer division per group and thread was most of the kernel's time: 78.9 -> .. us per launch)
struct Conv0Ends { int end[CUBE_MAX_CONV_GROUPS]; };
template <bool I8, int NV>  // NV quads of channels per thread: the row's index arithmetic and code loads are paid once per 4 * NV channels
__global__ __launch_bounds__(256) void conv0_lookup_kernel(const float* __restrict__ T, const float* __restrict__ btab,
                                                           const int32_t* __restrict__ cc, const int32_t* __restrict__ lang_word,
                                                           const cube_conv_groups cg, const Conv0Ends ce, int taps, int half, int n_combo,
                                                           uint16_t* __restrict__ out, uint16_t* __restrict__ out_lo, int ld_out,
                                                           int tpr, int rpb, int total_rows, int zero_code) {
  const int r_local = threadIdx.x / tpr;
  const int row = blockIdx.x * rpb + r_local;
  if (r_local >= rpb || row >= total_rows) return;
  const int c0 = (threadIdx.x - r_local * tpr) * 4 * NV;
  int g = 0;
#pragma unroll
  for (int i = 0; i < CUBE_MAX_CONV_GROUPS - 1; ++i) g += (i < cg.n - 1 && row >= ce.end[i]) ? 1 : 0;  // ends are non-decreasing
  const int seq_in = cg.seq_in[g], seq_out = cg.seq_out[g];
  const int rel = row - cg.row0_out[g];
  const int j = rel / seq_out, pos = rel - j * seq_out;
  const int in0 = cg.row0_in[g] + j * seq_in + pos - (taps >> 1);
  unsigned mask = 0, ldm = 0;  // taps inside the sequence (selects the bias row) / taps whose table row is read
  const size_t F = 2 * (size_t)half;
  // all character codes first, then all table rows (16-byte loads, all in flight), then the sums in tap order
  int code[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const int pp = pos + t - (taps >> 1);
    const bool ok = t < taps && pp >= 0 && pp < seq_in;
    if (ok) mask |= 1u << t;
    code[t] = ok ? cc[in0 + t] : 0;
    // zero_code: a combination whose table rows are all zero (the padding character: zero embedding rows) adds nothing,
    // so its rows are not read.  Layer 0 computes up to 8 positions past a word's end for the layers above it, and every
    // tap of theirs that falls on padding would fetch 2 KB of zeros: 70 % of the look-ups at a mean word length of 5.
    if (ok && code[t] != zero_code) ldm |= 1u << t;
  }
  float4 va[5][NV], vb[5][NV];
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    if ((ldm >> t) & 1u) {
      const float* tp = T + ((size_t)t * n_combo + code[t]) * F + c0;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        va[t][v] = *reinterpret_cast<const float4*>(tp + 4 * v);
        vb[t][v] = *reinterpret_cast<const float4*>(tp + half + 4 * v);
      }
    }
  }
  const float* bp = btab + ((size_t)lang_word[cg.word0[g] + j] * 32 + mask) * F + c0;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c4 = c0 + 4 * v;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      if ((ldm >> t) & 1u) {
        a.x += va[t][v].x; a.y += va[t][v].y; a.z += va[t][v].z; a.w += va[t][v].w;
        b.x += vb[t][v].x; b.y += vb[t][v].y; b.z += vb[t][v].z; b.w += vb[t][v].w;
      }
    }
    const float4 ba = *reinterpret_cast<const float4*>(bp + 4 * v);
    const float4 bb = *reinterpret_cast<const float4*>(bp + half + 4 * v);
    const float gv[4] = {tanhf_(a.x + ba.x) * sigmoidf_(b.x + bb.x), tanhf_(a.y + ba.y) * sigmoidf_(b.y + bb.y),
                         tanhf_(a.z + ba.z) * sigmoidf_(b.z + bb.z), tanhf_(a.w + ba.w) * sigmoidf_(b.w + bb.w)};
    if constexpr (I8) {  // int8x2 planes (cube_hip.h): q = round(g * CUBE_I8_ONE) = 256*hi + lo
      uint32_t ph = 0, pl = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int q = __float2int_rn(gv[k] * (float)CUBE_I8_ONE);
        q = q > CUBE_I8_ONE ? CUBE_I8_ONE : (q < -CUBE_I8_ONE ? -CUBE_I8_ONE : q);
        const int hq = (q + (1 << (CUBE_I8_SHIFT - 1))) >> CUBE_I8_SHIFT;
        ph |= (uint32_t)(hq & 0xff) << (8 * k);
        pl |= (uint32_t)((q - (hq << CUBE_I8_SHIFT)) & 0xff) << (8 * k);
      }
      const size_t ob = (size_t)row * ld_out + c4;  // bytes
      *reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(out) + ob) = ph;
      *reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(out_lo) + ob) = pl;
    } else {
      uint16_t h[4], l[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (out_lo) split_bf16(gv[k], h[k], l[k]); else h[k] = f32_to_bf16(gv[k]);
      }
      const size_t o = (size_t)row * ld_out + c4;
      *reinterpret_cast<uint2*>(out + o) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
      if (out_lo) *reinterpret_cast<uint2*>(out_lo + o) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
    }
  }
}

// ---- arg-max (first maximal index, like torch.argmax / numpy.argmax): one wave per row
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* x, int ld, int col0, int n, int M, int32_t* out) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* p = x + (size_t)row * ld + col0;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < n; c += 64) {
    const float v = p[c];
    if (v > best || (v != v && best == best)) { best = v; bi = c; }  // NaN counts as maximal
  }
  if (bi == 0x7fffffff && lane < n) bi = lane;  // all -inf in this lane's slice
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    const bool take = (ob > best) || (ob == best && oi < bi) || (ob != ob && best == best) ||
                      (ob != ob && best != best && oi < bi);
    if (take) { best = ob; bi = oi; }
  }
  if (lane == 0) out[row] = bi == 0x7fffffff ? 0 : bi;
}

}  // namespace

// ---- several gathers / arg-maxes of one stage in one launch (each was a 5 us launch of its own on the stage's stream)
namespace {
struct GatherBatch {
  cube_gather_desc d[CUBE_GATHER_BATCH_MAX];
  int vec_ok[CUBE_GATHER_BATCH_MAX];
};

__global__ __launch_bounds__(256) void gather_rows_batch_kernel(GatherBatch gb) {
  const cube_gather_desc& g = gb.d[blockIdx.y];
  uint16_t* dst = (uint16_t*)g.dst;
  uint16_t* dst_lo = (uint16_t*)g.dst_lo;
  const uint16_t* table = (const uint16_t*)g.table;
  const uint16_t* table_lo = (const uint16_t*)g.table_lo;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int i = wave; i < g.n; i += nwaves) {
    const int src_row = g.idx ? g.idx[i] : 0;
    const int dst_row = g.dst_rows ? g.dst_rows[i] : i;
    if (dst_row < 0) continue;
    const size_t so = (size_t)src_row * g.ld_table, dof = (size_t)dst_row * g.ld_dst + g.col0;
    if (gb.vec_ok[blockIdx.y]) {
      for (int c = lane * 8; c < g.width; c += 64 * 8) {
        *reinterpret_cast<uint4*>(dst + dof + c) = *reinterpret_cast<const uint4*>(table + so + c);
        if (dst_lo) *reinterpret_cast<uint4*>(dst_lo + dof + c) = *reinterpret_cast<const uint4*>(table_lo + so + c);
      }
    } else {
      for (int c = lane; c < g.width; c += 64) {
        dst[dof + c] = table[so + c];
        if (dst_lo) dst_lo[dof + c] = table_lo[so + c];
      }
    }
  }
}

// one wave per row: first maximum of x[row, col0 .. col0+n)  (NaN counts as maximal, as torch.argmax)
__device__ __forceinline__ int wave_argmax(const float* p, int n, int lane) {
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < n; c += 64) {
    const float v = p[c];
    if (v > best || (v != v && best == best)) { best = v; bi = c; }
  }
  if (bi == 0x7fffffff && lane < n) bi = lane;  // all -inf in this lane's slice
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    const bool take = (ob > best) || (ob == best && oi < bi) || (ob != ob && best == best) ||
                      (ob != ob && best != best && oi < bi);
    if (take) { best = ob; bi = oi; }
  }
  return bi == 0x7fffffff ? 0 : bi;
}

__global__ __launch_bounds__(256) void tag_argmax_kernel(const float* x, int ld, int M, int col_a, int n_a, int32_t* out_a,
                                                         int col_b, int n_b, int32_t* out_b, const uint16_t* table,
                                                         const uint16_t* table_lo, int ld_table, int width, uint16_t* dst,
                                                         uint16_t* dst_lo, int ld_dst, int col_dst) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* p = x + (size_t)row * ld;
  const int ia = wave_argmax(p + col_a, n_a, lane);
  if (lane == 0) out_a[row] = ia;
  if (out_b) {
    const int ib = wave_argmax(p + col_b, n_b, lane);
    if (lane == 0) out_b[row] = ib;
  }
  if (dst) {  // the embedding of the first decision into the row's columns [col_dst, col_dst + width)
    const size_t so = (size_t)ia * ld_table, dof = (size_t)row * ld_dst + col_dst;
    for (int c = lane; c < width; c += 64) {
      dst[dof + c] = table[so + c];
      if (dst_lo) dst_lo[dof + c] = table_lo[so + c];
    }
  }
}

__global__ __launch_bounds__(256) void pack_tags_kernel(const int32_t* tags, int cap, int B, int T, int root, int32_t* out) {
  const int L = T - root;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * L) return;
  const int b = t / L, i = t - b * L;
  const int src = b * T + i + root;
  out[t] = tags[src];                              // upos
  out[(size_t)B * L + t] = tags[2 * (size_t)cap + src];      // xpos
  out[2 * (size_t)B * L + t] = tags[(size_t)cap + src];      // attrs
}
}  // namespace

extern "C" int cube_gather_rows_batch(const cube_gather_desc* descs, int n_desc, void* stream) {
  CUBE_ENTRY();
  if (n_desc <= 0) return CUBE_OK;
  if (!descs || n_desc > CUBE_GATHER_BATCH_MAX) CUBE_FAIL(CUBE_ERR_ARG, "cube_gather_rows_batch: %d gathers (1..%d)", n_desc, CUBE_GATHER_BATCH_MAX);
  GatherBatch gb;
  int n_max = 0;
  for (int k = 0; k < n_desc; ++k) {
    const cube_gather_desc& g = descs[k];
    if (g.n < 0 || g.width <= 0 || !g.dst || !g.table) CUBE_FAIL(CUBE_ERR_ARG, "cube_gather_rows_batch: gather %d: null pointer or bad size", k);
    if (g.dst_lo && !g.table_lo) CUBE_FAIL(CUBE_ERR_ARG, "cube_gather_rows_batch: gather %d: dst_lo needs table_lo", k);
    if (g.col0 < 0 || g.col0 + g.width > g.ld_dst || g.width > g.ld_table) CUBE_FAIL(CUBE_ERR_ARG, "cube_gather_rows_batch: gather %d: bad columns", k);
    gb.d[k] = g;
    if (!g.dst_lo) gb.d[k].table_lo = nullptr;
    gb.vec_ok[k] = (g.width % 8 == 0) && (g.col0 % 8 == 0) && (g.ld_dst % 8 == 0) && (g.ld_table % 8 == 0) &&
                   ((uintptr_t)g.dst % 16 == 0) && ((uintptr_t)g.table % 16 == 0) && ((uintptr_t)g.dst_lo % 16 == 0) &&
                   ((uintptr_t)g.table_lo % 16 == 0);
    if (g.n > n_max) n_max = g.n;
  }
  if (n_max == 0) return CUBE_OK;
  int blocks = (n_max + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gather_rows_batch_kernel, dim3(blocks, n_desc), dim3(256), 0, (hipStream_t)stream, gb);
  CUBE_CHECK_LAUNCH("cube_gather_rows_batch");
  return CUBE_OK;
}

extern "C" int cube_tag_argmax(const float* x, int ld, int M, int col_a, int n_a, int32_t* out_a, int col_b, int n_b,
                               int32_t* out_b, const void* table, const void* table_lo, int ld_table, int width, void* dst,
                               void* dst_lo, int ld_dst, int col_dst, void* stream) {
  CUBE_ENTRY();
  if (M <= 0) return CUBE_OK;
  if (!x || !out_a || n_a <= 0 || col_a < 0 || col_a + n_a > ld) CUBE_FAIL(CUBE_ERR_ARG, "cube_tag_argmax: bad first range");
  if (out_b && (n_b <= 0 || col_b < 0 || col_b + n_b > ld)) CUBE_FAIL(CUBE_ERR_ARG, "cube_tag_argmax: bad second range");
  if (dst && (!table || width <= 0 || width > ld_table || col_dst < 0 || col_dst + width > ld_dst || (dst_lo && !table_lo)))
    CUBE_FAIL(CUBE_ERR_ARG, "cube_tag_argmax: bad embedding arguments");
  hipLaunchKernelGGL(tag_argmax_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ld, M, col_a, n_a, out_a, col_b,
                     n_b, out_b, (const uint16_t*)table, (const uint16_t*)(dst_lo ? table_lo : nullptr), ld_table, width,
                     (uint16_t*)dst, (uint16_t*)dst_lo, ld_dst, col_dst);
  CUBE_CHECK_LAUNCH("cube_tag_argmax");
  return CUBE_OK;
}

extern "C" int cube_pack_tags(const int32_t* tags, int cap, int B, int T, int root, int32_t* out, void* stream) {
  CUBE_ENTRY();
  if (B <= 0 || T - root <= 0) return CUBE_OK;
  if (!tags || !out || root < 0 || root > 1 || cap < B * T) CUBE_FAIL(CUBE_ERR_ARG, "cube_pack_tags: bad arguments");
  const int n = B * (T - root);
  hipLaunchKernelGGL(pack_tags_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, tags, cap, B, T, root, out);
  CUBE_CHECK_LAUNCH("cube_pack_tags");
  return CUBE_OK;
}

// results -> pinned host memory by a kernel's own stores (one dword per lane: 256-byte PCIe writes per wavefront)
__global__ __launch_bounds__(256) void copy_words_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

extern "C" int cube_copy_to_pinned(const void* src, void* dst_pinned, size_t bytes, void* stream) {
  CUBE_ENTRY();
  if (bytes == 0) return CUBE_OK;
  if (!src || !dst_pinned || bytes % 4 != 0 || (uintptr_t)src % 4 != 0 || (uintptr_t)dst_pinned % 4 != 0 || bytes > ((size_t)1 << 30))
    CUBE_FAIL(CUBE_ERR_ARG, "cube_copy_to_pinned: needs 4-byte aligned pointers and a multiple of 4 bytes (at most 1 GiB)");
  void* dptr = nullptr;
  if (hipHostGetDevicePointer(&dptr, dst_pinned, 0) != hipSuccess || !dptr) {
    (void)hipGetLastError();
    CUBE_FAIL(CUBE_ERR_ARG, "cube_copy_to_pinned: the destination is not pinned (page-locked, device-visible) host memory");
  }
  const size_t n = bytes / 4;
  hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint32_t*)src, (uint32_t*)dptr, n);
  CUBE_CHECK_LAUNCH("cube_copy_to_pinned");
  return CUBE_OK;
}

extern "C" int cube_gather_rows(void* dst, void* dst_lo, int ld_dst, int col0, int width, const void* table,
                                const void* table_lo, int ld_table, const int32_t* idx, const int32_t* dst_rows, int n,
                                void* stream) {
  CUBE_ENTRY();
  if (n <= 0 || width <= 0) return CUBE_OK;
  if (!dst || !table) CUBE_FAIL(CUBE_ERR_ARG, "cube_gather_rows: null pointer");
  if (dst_lo && !table_lo) CUBE_FAIL(CUBE_ERR_ARG, "cube_gather_rows: dst_lo needs table_lo");
  if (col0 < 0 || col0 + width > ld_dst || width > ld_table) CUBE_FAIL(CUBE_ERR_ARG, "cube_gather_rows: bad columns");
  const int vec_ok = (width % 8 == 0) && (col0 % 8 == 0) && (ld_dst % 8 == 0) && (ld_table % 8 == 0) &&
                     ((uintptr_t)dst % 16 == 0) && ((uintptr_t)table % 16 == 0) && ((uintptr_t)dst_lo % 16 == 0) &&
                     ((uintptr_t)table_lo % 16 == 0);
  const int waves_per_block = 4;
  int blocks = (n + waves_per_block - 1) / waves_per_block;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (uint16_t*)dst, (uint16_t*)dst_lo,
                     ld_dst, col0, width, (const uint16_t*)table, (const uint16_t*)(dst_lo ? table_lo : nullptr), ld_table, idx,
                     dst_rows, n, vec_ok);
  CUBE_CHECK_LAUNCH("cube_gather_rows");
  return CUBE_OK;
}

static int masked_mean_entry(bool i8, const void* const* x_hi, const void* const* x_lo, const float* weight, int n_layers,
                             int ld_x, const int32_t* len, int N, int W, int C, void* out, void* out_lo, int ld_out,
                             const int32_t* word_rows, void* stream) {
  if (N <= 0) return CUBE_OK;
  if (!x_hi || !len || !out || W <= 0 || C <= 0 || C > ld_x || C > ld_out) CUBE_FAIL(CUBE_ERR_ARG, "cube_masked_mean: bad args");
  if (n_layers <= 0 || n_layers > CUBE_MAX_MEAN_LAYERS) CUBE_FAIL(CUBE_ERR_ARG, "cube_masked_mean: n_layers=%d (max %d)", n_layers, CUBE_MAX_MEAN_LAYERS);
  MeanLayers ly;
  ly.n = n_layers;
  for (int k = 0; k < CUBE_MAX_MEAN_LAYERS; ++k) {
    ly.hi[k] = k < n_layers ? (const uint16_t*)x_hi[k] : nullptr;
    ly.lo[k] = (k < n_layers && x_lo) ? (const uint16_t*)x_lo[k] : nullptr;
    ly.weight[k] = (k < n_layers && weight) ? weight[k] : 1.0f;
    if (k < n_layers && !ly.hi[k]) CUBE_FAIL(CUBE_ERR_ARG, "cube_masked_mean: null layer %d", k);
  }
  bool vec = C % 8 == 0 && ld_x % 8 == 0 && ld_out % 8 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)out_lo % 16 == 0;
  for (int k = 0; k < n_layers; ++k) vec = vec && (uintptr_t)ly.hi[k] % 16 == 0 && (uintptr_t)ly.lo[k] % 16 == 0;
  if (i8) {
    if (!x_lo || !vec || C % 16 || ld_x % 16) CUBE_FAIL(CUBE_ERR_ALIGN, "cube_masked_mean_i8: both planes, C and ld_x multiples of 16, 16-byte aligned pointers");
    for (int k = 0; k < n_layers; ++k) if (!ly.lo[k]) CUBE_FAIL(CUBE_ERR_ARG, "cube_masked_mean_i8: null lo plane %d", k);
    hipLaunchKernelGGL(masked_mean_i8_kernel, dim3((N * (C / 16) * 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, ly, ld_x,
                       len, N, W, C, (uint16_t*)out, (uint16_t*)out_lo, ld_out, word_rows);
  } else if (vec)
    hipLaunchKernelGGL(masked_mean_vec_kernel, dim3((N * (C / 8) + 255) / 256), dim3(256), 0, (hipStream_t)stream, ly, ld_x,
                       len, N, W, C, (uint16_t*)out, (uint16_t*)out_lo, ld_out, word_rows);
  else
    hipLaunchKernelGGL(masked_mean_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, ly, ld_x, len, N, W, C,
                       (uint16_t*)out, (uint16_t*)out_lo, ld_out, word_rows);
  CUBE_CHECK_LAUNCH("cube_masked_mean");
  return CUBE_OK;
}

extern "C" int cube_masked_mean(const void* const* x_hi, const void* const* x_lo, const float* weight, int n_layers,
                                int ld_x, const int32_t* len, int N, int W, int C, void* out, void* out_lo, int ld_out,
                                const int32_t* word_rows, void* stream) {
  CUBE_ENTRY();
  return masked_mean_entry(false, x_hi, x_lo, weight, n_layers, ld_x, len, N, W, C, out, out_lo, ld_out, word_rows, stream);
}
extern "C" int cube_masked_mean_i8(const void* const* x_hi, const void* const* x_lo, const float* weight, int n_layers,
                                   int ld_x, const int32_t* len, int N, int W, int C, void* out, void* out_lo, int ld_out,
                                   const int32_t* word_rows, void* stream) {
  CUBE_ENTRY();
  return masked_mean_entry(true, x_hi, x_lo, weight, n_layers, ld_x, len, N, W, C, out, out_lo, ld_out, word_rows, stream);
}

extern "C" int cube_ext_finish(const float* acc, int ld_acc, int B, int L, int root, int P, int K, int parts, int part_stride,
                               void* dst_a, void* dst_a_lo, int ld_a, int col_a, void* dst_b, void* dst_b_lo, int ld_b, int col_b,
                               void* stream) {
  CUBE_ENTRY();
  if (B <= 0 || L <= 0 || P <= 0) return CUBE_OK;
  if (!acc || K <= 0 || P > ld_acc || (root != 0 && root != 1)) CUBE_FAIL(CUBE_ERR_ARG, "cube_ext_finish: bad args");
  if (parts < 1 || (parts > 1 && (part_stride < P || (long)(parts - 1) * part_stride + P > ld_acc)))
    CUBE_FAIL(CUBE_ERR_ARG, "cube_ext_finish: %d parts of stride %d do not fit rows of %d", parts, part_stride, ld_acc);
  if ((dst_a && col_a + P > ld_a) || (dst_b && col_b + P > ld_b)) CUBE_FAIL(CUBE_ERR_ARG, "cube_ext_finish: bad columns");
  const int total = B * L * P;
  hipLaunchKernelGGL(ext_finish_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, acc, ld_acc, B, L,
                     root, P, (float)K, parts, part_stride, (uint16_t*)dst_a, (uint16_t*)dst_a_lo, ld_a, col_a, (uint16_t*)dst_b,
                     (uint16_t*)dst_b_lo, ld_b, col_b);
  CUBE_CHECK_LAUNCH("cube_ext_finish");
  return CUBE_OK;
}

static int conv0_lookup_entry(bool i8, const float* T, const float* btab, const int32_t* cc, const int32_t* lang_word,
                              const cube_conv_groups* groups, int taps, int half, int n_combo, int zero_code, void* out,
                              void* out_lo, int ld_out, void* stream) {
  if (i8 && !out_lo) CUBE_FAIL(CUBE_ERR_ARG, "cube_char_conv0_lookup_i8: needs both planes");
  if (!T || !btab || !cc || !lang_word || !groups || !out) CUBE_FAIL(CUBE_ERR_ARG, "cube_char_conv0_lookup: null pointer");
  if ((taps != 1 && taps != 3 && taps != 5) || half <= 0 || half % 4 || half > 1024 || n_combo <= 0 || ld_out < half || ld_out % 4)
    CUBE_FAIL(CUBE_ERR_ARG, "cube_char_conv0_lookup: bad taps / half / ld_out");
  if ((uintptr_t)T % 16 || (uintptr_t)btab % 16 || (uintptr_t)out % 8 || (uintptr_t)out_lo % 8)
    CUBE_FAIL(CUBE_ERR_ALIGN, "cube_char_conv0_lookup: tables must be 16-byte aligned, outputs 8-byte aligned");
  const cube_conv_groups& cg = *groups;
  if (cg.n <= 0 || cg.n > CUBE_MAX_CONV_GROUPS) CUBE_FAIL(CUBE_ERR_ARG, "cube_char_conv0_lookup: %d sequence groups", cg.n);
  int total = 0;
  Conv0Ends ends = {};
  for (int g = 0; g < cg.n; ++g) {
    if (cg.seq_in[g] <= 0 || cg.n_rows[g] <= 0 || cg.n_rows[g] % cg.seq_in[g] || cg.seq_out[g] <= 0 ||
        cg.seq_out[g] > cg.seq_in[g] || cg.row0_out[g] != total || cg.row0_in[g] < 0 || cg.word0[g] < 0)
      CUBE_FAIL(CUBE_ERR_ARG, "cube_char_conv0_lookup: bad sequence group %d (output rows must be contiguous)", g);
    total += cg.n_rows[g] / cg.seq_in[g] * cg.seq_out[g];
    ends.end[g] = total;
  }
  // two quads of channels per thread when the row is wide enough for half a wave of such threads (65.1 -> 57.1 us per launch;
  // four quads: 82.9, the 40 table loads in flight cost the occupancy)
#ifndef CUBE_CONV0_NV
#define CUBE_CONV0_NV 2
#endif
  const int nv = (CUBE_CONV0_NV == 2 && half % 8 == 0 && half / 8 >= 32) ? 2 : 1;
  const int tpr = half / (4 * nv);
  const int rpb = tpr >= 256 ? 1 : 256 / tpr;
  if (tpr > 256) CUBE_FAIL(CUBE_ERR_ARG, "cube_char_conv0_lookup: half=%d too wide", half);
  const dim3 grid((total + rpb - 1) / rpb), block(256);
#define CONV0_LAUNCH(I8_, NV_)                                                                                          \
  hipLaunchKernelGGL((conv0_lookup_kernel<I8_, NV_>), grid, block, 0, (hipStream_t)stream, T, btab, cc, lang_word, cg, ends, \
                     taps, half, n_combo, (uint16_t*)out, (uint16_t*)out_lo, ld_out, tpr, rpb, total, zero_code)
  if (i8) { if (nv == 2) CONV0_LAUNCH(true, 2); else CONV0_LAUNCH(true, 1); }
  else { if (nv == 2) CONV0_LAUNCH(false, 2); else CONV0_LAUNCH(false, 1); }
#undef CONV0_LAUNCH
  CUBE_CHECK_LAUNCH("cube_char_conv0_lookup");
  return CUBE_OK;
}

extern "C" int cube_char_conv0_lookup(const float* T, const float* btab, const int32_t* cc, const int32_t* lang_word,
                                      const cube_conv_groups* groups, int taps, int half, int n_combo, int zero_code, void* out,
                                      void* out_lo, int ld_out, void* stream) {
  CUBE_ENTRY();
  return conv0_lookup_entry(false, T, btab, cc, lang_word, groups, taps, half, n_combo, zero_code, out, out_lo, ld_out, stream);
}
extern "C" int cube_char_conv0_lookup_i8(const float* T, const float* btab, const int32_t* cc, const int32_t* lang_word,
                                         const cube_conv_groups* groups, int taps, int half, int n_combo, int zero_code, void* out,
                                         void* out_lo, int ld_out, void* stream) {
  CUBE_ENTRY();
  return conv0_lookup_entry(true, T, btab, cc, lang_word, groups, taps, half, n_combo, zero_code, out, out_lo, ld_out, stream);
}

extern "C" int cube_argmax_rows(const float* x, int ld, int col0, int n, int M, int32_t* out, void* stream) {
  CUBE_ENTRY();
  if (M <= 0) return CUBE_OK;
  if (!x || !out || n <= 0 || col0 < 0 || col0 + n > ld) CUBE_FAIL(CUBE_ERR_ARG, "cube_argmax_rows: bad args");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ld, col0, n, M, out);
  CUBE_CHECK_LAUNCH("cube_argmax_rows");
  return CUBE_OK;
}

// ---- fp32 -> bf16 hi (+ lo) planes: the external embeddings arrive from the host as fp32 rows
namespace {
__global__ __launch_bounds__(256) void split_f32_kernel(const float4* __restrict__ x, size_t n4, uint2* __restrict__ hi,
                                                        uint2* __restrict__ lo) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = x[i];
    uint16_t h[4], l[4];
    split_bf16(v.x, h[0], l[0]);
    split_bf16(v.y, h[1], l[1]);
    split_bf16(v.z, h[2], l[2]);
    split_bf16(v.w, h[3], l[3]);
    hi[i] = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    if (lo) lo[i] = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
  }
}
}  // namespace

extern "C" int cube_split_f32(const float* x, size_t n, void* hi, void* lo, void* stream) {
  CUBE_ENTRY();
  if (n == 0) return CUBE_OK;
  if (!x || !hi) CUBE_FAIL(CUBE_ERR_ARG, "cube_split_f32: null pointer");
  if (n % 4 || (uintptr_t)x % 16 || (uintptr_t)hi % 8 || (uintptr_t)lo % 8)
    CUBE_FAIL(CUBE_ERR_ALIGN, "cube_split_f32: n must be a multiple of 4, x 16-byte and hi / lo 8-byte aligned");
  const size_t n4 = n / 4;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)x, n4, (uint2*)hi,
                     (uint2*)lo);
  CUBE_CHECK_LAUNCH("cube_split_f32");
  return CUBE_OK;
}

// ---- bf16 hi/lo rows -> int8x2 planes with one scale per column (input of a cube_gemm_i8x2 layer whose activations are not
// gate outputs: the first word-level gated convolution reads [word emb | char vector | lang emb | ext], parser.py:149-158)
__global__ __launch_bounds__(256) void quant_i8x2_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ x_lo, int ld_x,
                                                         int rows, int cols8, const float* __restrict__ inv_scale,
                                                         unsigned char* __restrict__ q, unsigned char* __restrict__ q_lo, int ld_q) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * cols8) return;
  const int r = idx / cols8, c = (idx - r * cols8) * 8;
  const uint4 h = *reinterpret_cast<const uint4*>(x + (size_t)r * ld_x + c);
  uint4 l = {0u, 0u, 0u, 0u};
  if (x_lo) l = *reinterpret_cast<const uint4*>(x_lo + (size_t)r * ld_x + c);
  const float4 s0 = *reinterpret_cast<const float4*>(inv_scale + c), s1 = *reinterpret_cast<const float4*>(inv_scale + c + 4);
  const uint32_t hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
  const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
  uint32_t t[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const uint32_t hb = (e & 1) ? (hw[e >> 1] & 0xffff0000u) : (hw[e >> 1] << 16);
    const uint32_t lb = (e & 1) ? (lw[e >> 1] & 0xffff0000u) : (lw[e >> 1] << 16);
    float v = (__uint_as_float(hb) + __uint_as_float(lb)) * sc[e];   // hi + lo is exact in fp32 (16 significant bits)
    v = fminf(fmaxf(v, -1.0f), 1.0f);
    // q = round-to-nearest-even(v * CUBE_I8_ONE): the low 16 bits of this sum are q + 128 (gemm.hip, int8x2 gate epilogue)
    t[e] = __float_as_uint(fmaf(v, (float)CUBE_I8_ONE, 12582912.0f + 128.0f));
  }
  static_assert(CUBE_I8_SHIFT == 8, "byte extraction written for 8 + 8 bits");
  uint32_t dh[2], dl[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const uint32_t p01 = __builtin_amdgcn_perm(t[4 * k + 1], t[4 * k], 0x05010400u), p23 = __builtin_amdgcn_perm(t[4 * k + 3], t[4 * k + 2], 0x05010400u);
    dl[k] = __builtin_amdgcn_perm(p23, p01, 0x05040100u) ^ 0x80808080u;
    dh[k] = __builtin_amdgcn_perm(p23, p01, 0x07060302u);
  }
  *reinterpret_cast<uint2*>(q + (size_t)r * ld_q + c) = uint2{dh[0], dh[1]};
  *reinterpret_cast<uint2*>(q_lo + (size_t)r * ld_q + c) = uint2{dl[0], dl[1]};
}

extern "C" int cube_quant_i8x2(const void* x, const void* x_lo, int ld_x, int rows, int cols, const float* inv_scale, void* q,
                               void* q_lo, int ld_q, void* stream) {
  CUBE_ENTRY();
  if (rows <= 0 || cols <= 0) return CUBE_OK;
  if (!x || !inv_scale || !q || !q_lo) CUBE_FAIL(CUBE_ERR_ARG, "cube_quant_i8x2: null pointer");
  if (cols % 8 || ld_x < cols || ld_q < cols || ld_x % 8 || ld_q % 8) CUBE_FAIL(CUBE_ERR_ARG, "cube_quant_i8x2: cols=%d ld_x=%d ld_q=%d (multiples of 8)", cols, ld_x, ld_q);
  if ((uintptr_t)x % 16 || (uintptr_t)x_lo % 16 || (uintptr_t)inv_scale % 16 || (uintptr_t)q % 8 || (uintptr_t)q_lo % 8)
    CUBE_FAIL(CUBE_ERR_ALIGN, "cube_quant_i8x2: x / inv_scale 16-byte aligned, q 8-byte aligned");
  if ((long long)rows * (cols / 8) > 0x7fffffffLL) CUBE_FAIL(CUBE_ERR_ARG, "cube_quant_i8x2: too large");
  const int total = rows * (cols / 8);
  hipLaunchKernelGGL(quant_i8x2_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                     (const uint16_t*)x_lo, ld_x, rows, cols / 8, inv_scale, (unsigned char*)q, (unsigned char*)q_lo, ld_q);
  CUBE_CHECK_LAUNCH("cube_quant_i8x2");
  return CUBE_OK;
}

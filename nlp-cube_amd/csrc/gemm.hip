// Fused bf16 MFMA GEMM / implicit Conv1d for gfx950.
//
// One kernel serves every dense contraction on the parser path: the gated Conv1d(k=5) stacks
// (5 shifted K-slices of the same activation rows, no im2col), the Linear layers and the LSTM
// input projections.  Two tile shapes (256x256 / 8 waves for the big convolutions, 128x128 /
// 4 waves otherwise) over v_mfma_f32_32x32x16_bf16; BK = 32; operands go global -> LDS by
// LDS-DMA (global_load_lds_dwordx4) into double-buffered images with 64-byte rows whose 16-byte
// chunks are XOR-swizzled by (row>>2)&3 (conflict-free ds_read_b128 fragment reads; the swizzle
// rides on the DMA source address); one barrier per K-step.  The activation rows of a tile are
// staged once per 32-channel chunk and shared by the taps.
//
// Precision: operands are bf16, accumulation fp32.  In SPLIT mode ("bf16x3") every operand
// is a pair of bf16 planes, x ~= hi + lo (16 significant bits), and each product is formed as
// hi*hi + hi*lo + lo*hi on the matrix cores — fp32-grade results (what the fp32 reference
// computes, to ~1e-5) at a third of the bf16 MFMA rate, still 5x the fp32 MFMA rate.
//
// Epilogues are fused: bias, tanh / relu, the tanh*sigmoid gate with residual accumulation,
// row scatter (unpack), hi/lo splitting of bf16 outputs.
#include <type_traits>

#include "common.h"
#include "../../include/cube_hip.h"

namespace {

constexpr int BK = 32;
constexpr int ROWB = 64;  // bytes per LDS row (32 bf16); 16-byte chunks XOR-swizzled by (row>>2)&3

// Tile configurations.  Big: 256x256, 8 waves (2x4), 128x64 per wave — half the LDS fill per MFMA
// of the small one.  Small: 128x128, 8 waves (4x2), 32x64 per wave, 2 blocks/CU — for the GEMMs
// with few rows.
// SR = activation slab rows in LDS (a multiple of the 16 rows one LDS-DMA piece fills).  A tile is
// BM OUTPUT rows; the slab holds the input rows they read (+ halo).  For plain sequences that is
// BM + 4 rows; with ragged groups (seq_out < seq_in: the last positions of every sequence are read
// as neighbours but produce no output) the input span of a tile is up to seq_in/seq_out longer, so
// the slab is made as large as the 160 KiB of LDS allow and a tile takes fewer rows if it must.
struct CfgBig { static constexpr int BM = 256, BN = 256, WM = 2, WN = 4, SR = 352, NB = 2; };
// 8 waves (4x2, 32x64 per wave): these grids are about one block per CU, and two waves per SIMD cover
// each other's barrier / LDS latency (4 waves of 64x64 were 2..25 % slower, most on the short-K GEMMs)
struct CfgSmall { static constexpr int BM = 128, BN = 128, WM = 4, WN = 2, SR = 176, NB = 2; };
// Shorter variant of the big tile (8 waves, 96x64 per wave).  A launch of T equal tiles on 256 CUs
// at one block per CU takes ceil(T/256) rounds: when the last round of 256-row tiles would be mostly
// empty, more, shorter tiles finish sooner (see `launch_split`; a 128-row variant lost what it
// gained to its lower per-tile efficiency and was dropped).
struct CfgBig192 { static constexpr int BM = 192, BN = 256, WM = 2, WN = 4, SR = 272, NB = 2; };
// NB = weight-tile stages in LDS.  With three, the weight tile of step s+2 is in flight while step s
// runs: the end-of-step wait is a COUNTED vmcnt (that prefetch stays outstanding across a raw
// s_barrier) instead of the full drain a two-stage ring needs.  Measured: +12 % on the few-row
// convolutions below; nothing on the 192/256-row tiles (a 192-row, three-stage variant ran at the
// speed of the two-stage one: at the clock the chip holds under this load those are MFMA-bound, not
// latency-bound), which therefore keep two stages and the longer activation slab.
// few-row convolutions whose 128-row tiles would be a little over one block per CU: 160-row tiles on
// 10 waves (5x2), one block per CU; four weight stages = k-steps in pairs between barriers like CfgI8 below
// (a k-step of this tile is 12 MFMAs per wave: +2 % sentences/s over three stages with one barrier per step)
struct CfgSmall160 { static constexpr int BM = 160, BN = 128, WM = 5, WN = 2, SR = 176, NB = 4; };
// int8x2 operands (cube_gemm_i8x2): two int32 accumulator sets per output (the slice products differ
// by 2^7), so a wave keeps 64x64; 256x128 / 8 waves (4x2).  Tall and narrow: the activation slab is
// staged once per 5 taps, the weight tile once per tap, so rows are the cheaper side to extend.
// Measured: three weight stages 134.4k sentences/s, two 132.9k; a 128x128 tile on 4 waves at two blocks
// per CU 128.2k, 128x64 per wave on 4 waves (accumulators in AGPRs, one wave per SIMD) 121.1k (both dropped).
// NB = 4: k-steps run in PAIRS between barriers (a k-step of this tile is only 24 MFMAs per wave, so
// the per-step barrier, first fragment reads and DMA drain weigh twice what they do in the bf16 tile).
// Step s reads weight stage s % 4; its DMAs for step s + 2 go to stage (s + 2) % 4, last read at step
// s - 2, i.e. before the previous pair's barrier; the full-drain barrier follows every odd step, so
// everything issued in a pair has landed before the next pair.  The activation slab of chunk kc + 1
// goes out on taps 1..3 of chunk kc only: tap 0 may share its pair with the last tap of chunk kc - 1
// (still reading that buffer), tap 4 with the first tap of chunk kc + 1 (already reading the new one).
struct CfgI8 { static constexpr int BM = 256, BN = 128, WM = 4, WN = 2, SR = 352, NB = 4; };

constexpr int MAX_TAPS = 5;
#ifndef CUBE_AOFF_SELECT
#define CUBE_AOFF_SELECT 1
#endif
#ifndef CUBE_EPI_PREFETCH
#define CUBE_EPI_PREFETCH 1
#endif
#ifndef CUBE_I8_AD
#define CUBE_I8_AD 2  // units the int8x2 tile's activation fragments run ahead of their MFMAs (A/B: scripts/ab_build.py)
#endif
// -DCUBE_GEMM_PROF (scripts/ab_build.py + scripts/gemm_phase_prof.py): wall-clock stamps (10 ns units) per workgroup of the
// int8x2 character-CNN tile -- entry, prologue done, k-loop done, end -- read back with cube_gemm_prof()
#ifdef CUBE_GEMM_PROF
__device__ long long g_gemm_prof[4 * 4096];
#define PROF_STAMP(name) const long long name = wall_clock64()
#else
#define PROF_STAMP(name)
#endif
constexpr bool EPI_PREFETCH = CUBE_EPI_PREFETCH != 0;  // (A/B switch of the int8x2 gate epilogue's first loads, scripts/ab_build.py)
template <class C> constexpr int slab_rows() { return C::SR; }
// LDS: [A slab x2 buffers][B tile x NB stages][one zero row][per tile row: bias-table offset, output
// row, tap mask, slab row]; every operand buffer holds NP planes
template <class C, bool SPLIT> constexpr int lds_bytes() {
  return (SPLIT ? 2 : 1) * (2 * slab_rows<C>() + C::NB * C::BN) * ROWB + ROWB + 4 * C::BM * 4;
}
// output rows one tile of a sequence group takes: all BM unless the input rows they read would
// not fit the slab.  r rows touch at most (r-2)/seq_out + 2 sequences of seq_in rows each.
__host__ __device__ inline int tile_rows(int BM, int SR, int seq_in, int seq_out) {
  if (seq_in == seq_out) return BM;
  const int q = (SR - (MAX_TAPS - 1)) / seq_in;  // whole sequences the slab can hold
  const int r = (q - 1) * seq_out + 1;
  return r < BM ? r : BM;
}

struct GemmParams {
  const uint16_t* A; const uint16_t* A_lo; int lda; int M; int seq_len; int taps; int Kc;
  const uint16_t* W; const uint16_t* W_lo; int N; const float* bias;
  int n_valid; int accumulate;
  void* out0; uint16_t* out0_lo; int ldo0; int col0;
  const int32_t* rowmap;
  float* res; int ld_res; int res_mode;
  uint16_t* out1; uint16_t* out1_lo; int ldo1; int col1;
  uint16_t* out2; uint16_t* out2_lo; int ldo2; int col2;
  int m_tiles; int n_tiles;  // set by launch_cfg
  int vec_out;               // gate epilogue without residual: 16-byte row-contiguous stores are legal
  int batch; long bs_a; long bs_w; int bs_bias; int bs_col0;  // batched Linear layers: problem z = blockIdx.y (strides in elements)
  int out0_i8;               // bf16 operands, gate epilogue with residual: out0 = int8x2 planes of the gate value (ldo0, col0 in bytes)
  const float* bias_tab; const int32_t* row_group;  // row-dependent bias (broadcast channels folded in)
  const float* w_scale;      // int8x2 only: per packed column, value of one unit of 128*hh + (hl + lh)
};

// Ragged sequence groups (host struct cube_conv_groups passed by value): group g owns input rows
// [row0_in, row0_in + n_rows) as sequences of seq_in rows, writes the first seq_out positions of
// each sequence to rows row0_out + j*seq_out + pos, and its sequences are words word0, word0+1, ...
typedef cube_conv_groups ConvGroups;
// per group, computed on the host for the tile shape of the launch: first row tile, output rows per tile, output
// rows of the group (the kernel's tile -> group search is then a scalar scan without integer divisions)
struct ConvTiles { int tile0[CUBE_MAX_CONV_GROUPS + 1]; int bm[CUBE_MAX_CONV_GROUPS]; int n_out[CUBE_MAX_CONV_GROUPS]; };

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

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 2) & 3)) << 4); }

// LDS-DMA: 64 lanes x 16 bytes land at (wave-uniform) l + lane*16; the per-lane SOURCE address
// carries the swizzle (cdna guide: linear destination + permuted source + permuted read).
__device__ __forceinline__ void glds16(const uint16_t* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// K loop order: channel chunk kc (32 channels) outer, tap inner; k-step s = kc*taps + tap reads
// W[n][s*32 .. s*32+32).  For each kc the block stages the A rows its output rows read (+ taps/2 halo)
// x 32 channels ONCE (the "slab") and the taps read it at shifted rows: the activations cross
// HBM/L2 -> LDS once per output tile instead of once per tap.  Rows that a tap would take from a
// neighbouring sequence (Conv1d zero padding) are redirected, per lane, to a zero row in LDS.
// Both operands go global -> LDS by LDS-DMA (no staging registers, no ds_write), one k-step ahead.
//
// I8 ("int8x2", cube_gemm_i8x2): the same kernel over int8 planes.  A 64-byte LDS row then holds 64
// channels instead of 32 and every address below is unchanged when counted in 2-byte units (the host
// passes lda/2 and Kc/2); x = 128*hi + lo, the products hi*hi and hi*lo + lo*hi accumulate exactly
// in two int32 sets (v_mfma_i32_32x32x32_i8, twice the bf16 rate) and are combined in the epilogue.
// RES8 (int8x2 only): the epilogue carries the fp32 residual stream and the x + res side outputs (word-level stack); a
// template parameter so that the character-CNN launches keep their lean epilogue and their own kernel name in traces
template <int EPI, bool SPLIT, class C, bool I8 = false, bool RES8 = false>
__global__ __launch_bounds__(64 * C::WM * C::WN) void gemm_kernel(const GemmParams p_in, const ConvGroups cg, const ConvTiles ct) {
  GemmParams p = p_in;
  if (p.batch > 1) {  // same-shape problems side by side in grid.y (uniform: scalar arithmetic)
    const long z = blockIdx.y;
    p.A += z * p.bs_a;
    p.W += z * p.bs_w;
    if (p.A_lo) { p.A_lo += z * p.bs_a; p.W_lo += z * p.bs_w; }
    if (p.bias) p.bias += z * p.bs_bias;
    p.col0 += (int)z * p.bs_col0;
  }
  static_assert(!I8 || (SPLIT && EPI == CUBE_EPI_GATE), "int8x2 is built for the gated convolutions only");
  using acc_t = typename std::conditional<I8, i32x16, f32x16>::type;
  using frag_t = typename std::conditional<I8, i32x4, bf16x8>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  PROF_STAMP(prof_t0);
  constexpr int BM = C::BM, BN = C::BN;
  constexpr int NW = C::WM * C::WN;
  constexpr int TM = BM / C::WM / 32, TN = BN / C::WN / 32;  // 32x32 MFMA tiles per wave
  constexpr int NP = SPLIT ? 2 : 1;                          // planes per operand
  constexpr int SR = slab_rows<C>();
  constexpr int A_PLANE = SR * ROWB, B_PLANE = BN * ROWB;
  constexpr int A_BUF = NP * A_PLANE, B_BUF = NP * B_PLANE;
  constexpr int B_BASE = 2 * A_BUF;
  constexpr int NB = C::NB, PD = NB == 4 ? 2 : NB - 1;       // weight-tile stages, prefetch distance (k-steps)
  constexpr bool PAIRS = NB == 4;                            // two k-steps per barrier (see CfgI8)
  constexpr int ZERO_OFF = 2 * A_BUF + NB * B_BUF;
  constexpr int A_PIECES = SR / 16, B_PIECES = BN / 16;      // 1-KiB DMA pieces per plane
  constexpr int PA = (A_PIECES + NW - 1) / NW, PB = (B_PIECES + NW - 1) / NW;
  constexpr bool B_EVEN = B_PIECES % NW == 0;                // every wave issues PB weight pieces per tile
  static_assert(TN == 2, "the gate epilogue pairs the two 32-column tiles of a wave");
  static_assert(NB == 2 || NB == 3 || NB == 4, "two or three weight stages, or four with paired k-steps");
  static_assert(B_EVEN || B_PIECES % NW > 0, "");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid / C::WN, wc = wid % C::WN;
  // 1-D grid, XCD-aware: workgroups go round-robin over the 8 XCDs (each with its own L2), so the
  // N-tiles that share one row tile (= the same activation slab) are given consecutive slots of
  // the SAME XCD and the slab crosses HBM -> L2 once.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mt = (slot / p.n_tiles) * 8 + xcd;
  if (mt >= p.m_tiles) return;  // whole block: before any barrier
  // the row tile's sequence group (one implicit group = the whole matrix when cg.n == 0): the tile
  // takes output rows [ro0, ro0 + bm) of the group; in0 = the input row its first output row reads
  int row_lo = 0, seq = p.seq_len, seq_out = p.seq_len, row0_out = 0, word0 = 0;
  int n_out = p.M, ro0 = mt * BM, bm = BM;
  if (cg.n > 0) {
    // tile0 is non-decreasing: g = how many groups start at or before this tile.  Constant indices, so the table comes in
    // with two wide scalar loads and the search is 31 scalar compares (a data-dependent loop reads it one dependent
    // scalar load at a time: about a microsecond of every workgroup's prologue)
    int g = 0;
#pragma unroll
    for (int i = 1; i < CUBE_MAX_CONV_GROUPS; ++i) g += (i < cg.n && mt >= ct.tile0[i]) ? 1 : 0;
    const int t0 = ct.tile0[g];
    bm = ct.bm[g];
    n_out = ct.n_out[g];
    row_lo = cg.row0_in[g]; seq = cg.seq_in[g]; seq_out = cg.seq_out[g];
    row0_out = cg.row0_out[g]; word0 = cg.word0[g];
    ro0 = (mt - t0) * bm;
  }
  const int m0 = ro0;  // first output row of the tile (plain sequences: output row = input row)
  const int in0 = row_lo + (ro0 / seq_out) * seq + ro0 % seq_out;
  const int n0 = (slot % p.n_tiles) * BN;
  const int taps = p.taps;
  const int half_taps = taps >> 1;
  const int nkc = p.Kc / BK;
  const int nk = nkc * taps;
  const int ktot = taps * p.Kc;

  // ---- DMA source offsets (elements) of this lane for the pieces its wave issues
  size_t a_src[PA], b_src[PB];
  {
    const int prow = lane >> 2, ppos = lane & 3;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int row = 16 * (wid + NW * i) + prow;           // slab row
      long g = (long)in0 - half_taps + row;                 // global activation row, clamped: rows
      g = g < 0 ? 0 : (g >= p.M ? p.M - 1 : g);             // outside [0, M) are never read unmasked
      a_src[i] = (size_t)g * p.lda + ((ppos ^ ((row >> 2) & 3)) << 3);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int row = 16 * (wid + NW * i) + prow;
      b_src[i] = (size_t)(n0 + row) * ktot + ((ppos ^ ((row >> 2) & 3)) << 3);
    }
  }

  // ---- per-lane fragment rows and tap validity (bit t of vmask[i]: tap t reads a real row)
  const int l31 = lane & 31, lh = lane >> 5;
  int a_row[TM];
  unsigned vmask[TM];
  int b_rd[TN][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int j = 0; j < TN; ++j) b_rd[j][ks] = lds_off(wc * (BN / C::WN) + j * 32 + l31, 2 * ks + lh);

  acc_t acc[TM][TN];
  acc_t acc2[I8 ? TM : 1][TN];  // int8x2: the cross terms hi*lo + lo*hi
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[i][j][r] = 0;
        if constexpr (I8) acc2[i][j][r] = 0;
      }

#define ISSUE_A_PIECE(KC, BUF, I)                                                             \
  {                                                                                           \
    const int q_ = wid + NW * (I);                                                            \
    if (q_ < A_PIECES) {                                                                      \
      unsigned char* l_ = smem + (BUF) * A_BUF + q_ * 1024;                                   \
      glds16(p.A + a_src[I] + (KC) * BK, l_);                                                 \
      if constexpr (SPLIT) glds16(p.A_lo + a_src[I] + (KC) * BK, l_ + A_PLANE);               \
    }                                                                                         \
  }
#define ISSUE_B_PIECE(S, BUF, I)                                                              \
  if (B_EVEN || wid + NW * (I) < B_PIECES) {                                                  \
    unsigned char* l_ = smem + B_BASE + (BUF) * B_BUF + (wid + NW * (I)) * 1024;              \
    glds16(p.W + b_src[I] + (size_t)(S) * BK, l_);                                            \
    if constexpr (SPLIT) glds16(p.W_lo + b_src[I] + (size_t)(S) * BK, l_ + B_PLANE);          \
  }
#define ISSUE_A(KC, BUF) _Pragma("unroll") for (int i_ = 0; i_ < PA; ++i_) ISSUE_A_PIECE(KC, BUF, i_)
#define ISSUE_B(S, BUF) _Pragma("unroll") for (int i_ = 0; i_ < PB; ++i_) ISSUE_B_PIECE(S, BUF, i_)

  ISSUE_A(0, 0)
  ISSUE_B(0, 0)
  if constexpr (NB >= 3) {
    if (nk > 1) ISSUE_B(1, 1)
  }
  if (tid < ROWB / 4) reinterpret_cast<uint32_t*>(smem + ZERO_OFF)[tid] = 0u;
  // (after the first DMAs are out: their flight covers this)  per tile row, computed once (one division per row): which taps fall inside the row's sequence,
  // where its output goes (-1: nowhere), which slab row it reads and the offset of its bias vector
  // in bias_tab.  Read after the first barrier below.
  constexpr int ROWOFF_OFF = ZERO_OFF + ROWB, OROW_OFF = ROWOFF_OFF + BM * 4, MASK_OFF = OROW_OFF + BM * 4,
                AROW_OFF = MASK_OFF + BM * 4;
  if (tid < BM) {
    const int ro = ro0 + tid;
    unsigned off = 0, m = 0;
    int orow = -1, arow = 0;
    if (tid < bm && ro < n_out) {
      const int j = ro / seq_out, pos = ro - j * seq_out;
      for (int t = 0; t < taps; ++t) {
        const int pp = pos + t - half_taps;
        if (pp >= 0 && pp < seq) m |= 1u << t;
      }
      orow = row0_out + ro;
      arow = row_lo + j * seq + pos - in0;
      if (p.bias_tab) off = ((unsigned)p.row_group[cg.n > 0 ? word0 + j : ro] * 32u + m) * (unsigned)p.N;
    }
    reinterpret_cast<uint32_t*>(smem + ROWOFF_OFF)[tid] = off;
    reinterpret_cast<int32_t*>(smem + OROW_OFF)[tid] = orow;
    reinterpret_cast<uint32_t*>(smem + MASK_OFF)[tid] = m;
    reinterpret_cast<int32_t*>(smem + AROW_OFF)[tid] = arow;
  }

  __syncthreads();  // drains the first DMAs (vmcnt(0)) and publishes the row tables
  PROF_STAMP(prof_t1);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int lr = wr * (BM / C::WM) + i * 32 + l31;  // row inside the block tile
    vmask[i] = reinterpret_cast<const uint32_t*>(smem + MASK_OFF)[lr];
    a_row[i] = reinterpret_cast<const int32_t*>(smem + AROW_OFF)[lr];  // slab row of tap 0 is a_row + 0: the slab starts half_taps rows early
  }
  // LDS-DMA instructions this wave issues per weight tile (what a counted wait leaves in flight)
  const bool b_full = B_EVEN || wid + NW * (PB - 1) < B_PIECES;
  // int8x2 gate epilogue: what its first group of channels needs -- the weight scales and, for the lane's row in each of
  // the wave's row tiles, the bias-table values -- is loaded under the MFMAs of the LAST k-step (pre_*), so that the
  // epilogue starts computing at once and the loads of the other three groups arrive behind that math.  (Loaded at the top of
  // the epilogue, every wave of the tile sat out one L2 round trip with the matrix cores idle: 143 -> 137 us per launch when
  // the loads of both row tiles were at least issued together, -> ... with the first group ahead.)
  float4 pre_ws[2] = {}, pre_b[I8 ? TM : 1][2] = {};
  const float* bsrc_row[I8 ? TM : 1] = {};
  if constexpr (I8) {
    const int wcol0_ = n0 + wc * (BN / C::WN);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int lr = wr * (BM / C::WM) + i * 32 + l31;
      bsrc_row[i] = p.bias_tab ? p.bias_tab + (size_t)reinterpret_cast<const uint32_t*>(smem + ROWOFF_OFF)[lr] + wcol0_ : p.bias + wcol0_;
    }
  }

  int kc = 0, tap = 0, stg = 0;  // stg = weight stage of step s
  for (int s = 0; s < nk; ++s) {
    const bool last_tap = (tap == taps - 1);
    if constexpr (I8 && EPI_PREFETCH) {
      if (s == nk - 1) {
        const int wcol0_ = n0 + wc * (BN / C::WN);
        pre_ws[0] = *reinterpret_cast<const float4*>(p.w_scale + wcol0_ + 4 * lh);
        pre_ws[1] = *reinterpret_cast<const float4*>(p.w_scale + wcol0_ + 32 + 4 * lh);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          pre_b[i][0] = *reinterpret_cast<const float4*>(bsrc_row[i] + 4 * lh);
          pre_b[i][1] = *reinterpret_cast<const float4*>(bsrc_row[i] + 32 + 4 * lh);
        }
      }
    }
    // operands of later steps: one LDS-DMA piece after each MFMA group below (a piece costs the wave
    // 60-180 issue cycles; spread out, the queued MFMAs cover it, bunched at the top they do not)
    const bool pre = s + PD < nk;  // a weight tile to prefetch in this step
    const int stg_pre = stg + PD >= NB ? stg + PD - NB : stg + PD;
    constexpr int PPU = (PA + PB + 2 * TM - 1) / (2 * TM);  // pieces per MFMA group
    const unsigned char* bbase = smem + B_BASE + stg * B_BUF;
    // fragment reads run one (k-substep, row tile) unit ahead of the MFMAs that consume them
    int aoff[TM], alo[TM];  // LDS byte offsets of this lane's A fragment (k-substep 0) and hi->lo distance
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = a_row[i] + tap;  // slab row of this lane's output row shifted by the tap
#if CUBE_AOFF_SELECT
      // branch-free: the compiler turns the conditional form into two divergent branches at the top of every k-step
      const int okm = -(int)((vmask[i] >> tap) & 1u);
      aoff[i] = ZERO_OFF + (((kc & 1) * A_BUF + lds_off(row, lh) - ZERO_OFF) & okm);
      alo[i] = A_PLANE & okm;
#else
      const bool ok = (vmask[i] >> tap) & 1u;
      aoff[i] = ok ? (kc & 1) * A_BUF + lds_off(row, lh) : ZERO_OFF;
      alo[i] = ok ? A_PLANE : 0;
#endif
    }
    // AD = how many units the activation fragments run ahead.  The int8x2 tile's units are short (6 MFMAs)
    // and two waves' reads queue in front of each other: two units ahead, and the second k-substep's
    // weight fragments right behind the first's (137.5k vs 135.5-137.1k sentences/s).  The same in the bf16
    // tiles: 133.3k (dropped).
    constexpr int AD = I8 ? CUBE_I8_AD : 1, AR = AD + 1;
    frag_t fb[2][NP][TN];
    frag_t fa[AR][NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[0][pl][j] = *reinterpret_cast<const frag_t*>(bbase + pl * B_PLANE + b_rd[j][0]);
#pragma unroll
    for (int u0 = 0; u0 < AD && u0 < 2 * TM; ++u0) {
      const int o = aoff[u0 % TM] ^ (u0 / TM ? 32 : 0);
      fa[u0][0] = *reinterpret_cast<const frag_t*>(smem + o);
      if constexpr (SPLIT) fa[u0][1] = *reinterpret_cast<const frag_t*>(smem + o + alo[u0 % TM]);
    }
    if constexpr (AD > 1) {
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[1][pl][j] = *reinterpret_cast<const frag_t*>(bbase + pl * B_PLANE + b_rd[j][1]);
    }
#pragma unroll
    for (int u = 0; u < 2 * TM; ++u) {
      const int ks = u / TM, i = u % TM;
      if (u + AD < 2 * TM) {  // prefetch a later unit's A fragments
        const int ks2 = (u + AD) / TM, i2 = (u + AD) % TM;
        // chunk = (2*ks + lh) ^ swz: the k-substep toggles bit 1 of the chunk index = offset ^ 32
        const int o = aoff[i2] ^ (ks2 ? 32 : 0);
        fa[(u + AD) % AR][0] = *reinterpret_cast<const frag_t*>(smem + o);
        if constexpr (SPLIT) fa[(u + AD) % AR][1] = *reinterpret_cast<const frag_t*>(smem + o + alo[i2]);
      }
      if (AD == 1 && u + 1 == TM) {  // once, the second k-substep's B
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            fb[1][pl][j] = *reinterpret_cast<const frag_t*>(bbase + pl * B_PLANE + b_rd[j][1]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (I8) {
          // weights as the A operand: the accumulators hold the TRANSPOSED tile (lane = activation row, registers =
          // channels in runs of four), which the epilogue packs into dwords without crossing lanes
          acc2[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[ks][0][j], fa[u % AR][1], acc2[i][j], 0, 0, 0);
          acc2[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[ks][1][j], fa[u % AR][0], acc2[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[ks][0][j], fa[u % AR][0], acc[i][j], 0, 0, 0);
        } else {
          if constexpr (SPLIT) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u % AR][1], fb[ks][0][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u % AR][0], fb[ks][1][j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u % AR][0], fb[ks][0][j], acc[i][j], 0, 0, 0);
        }
      }
      // DMA slots q of this unit: the weight pieces of step s+PD and the activation pieces of the
      // NEXT channel chunk.  The slab buffer of chunk kc+1 is free during all taps of chunk kc, so
      // its pieces are spread over the taps (piece a goes out on tap a % taps) instead of bunching
      // on the last tap.  Two stages: weights first (they must land within this step).  Three
      // stages: activations first -- the counted wait at the end of the step then leaves exactly
      // this step's weight pieces in flight and everything older (the slab pieces, the weight tile
      // of step s+1) has landed.
#pragma unroll
      for (int q = u * PPU; q < (u + 1) * PPU && q < PA + PB; ++q) {
        const int qb = NB != 3 ? q : q - PA, qa = NB != 3 ? q - PB : q;
        if (qb >= 0 && qb < PB) {
          if (pre) ISSUE_B_PIECE(s + PD, stg_pre, qb)
        } else if (qa >= 0 && qa < PA) {
          if (kc + 1 < nkc && (!PAIRS || (tap >= 1 && tap <= taps - 2))) {
            const int a = PAIRS ? qa * (taps - 2) + tap - 1 : qa * taps + tap;  // wave-uniform
#pragma unroll
            for (int ai = 0; ai < PA; ++ai)
              if (a == ai) ISSUE_A_PIECE(kc + 1, (kc + 1) & 1, ai)
          }
        }
      }
    }
    if constexpr (NB == 2) {
      __syncthreads();  // also drains this step's LDS-DMA (vmcnt(0))
    } else if constexpr (PAIRS) {
      if ((s & 1) || s == nk - 1) __syncthreads();
    } else {
      // every wave waits for its own older DMAs, then the barrier: after it all pieces of step s+1
      // (from all waves) are in LDS, and all reads of stage `stg` (restaged next step) are done
      if (pre) {
        if (b_full) wait_vm<NP * PB>(); else wait_vm<NP * (PB - 1)>();
      } else {
        wait_vm<0>();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if (last_tap) { tap = 0; ++kc; } else { ++tap; }
    stg = stg + 1 == NB ? 0 : stg + 1;
  }
#undef ISSUE_A
#undef ISSUE_B
#undef ISSUE_A_PIECE
#undef ISSUE_B_PIECE

  PROF_STAMP(prof_t2);
  // ---- epilogue.  C/D map of 32x32x16: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int wrow0 = m0 + wr * (BM / C::WM);
  const int wcol0 = n0 + wc * (BN / C::WN);
  if constexpr (EPI == CUBE_EPI_GATE) {
    const int ch = wcol0 / 2 + l31;
    const float bias_a = p.bias[wcol0 + l31];
    const float bias_b = p.bias[wcol0 + 32 + l31];
    const float* __restrict__ btab = p.bias_tab;
    // offset (floats) of row r's bias vector inside bias_tab
    const uint32_t* rowoff = reinterpret_cast<const uint32_t*>(smem + ROWOFF_OFF);
    const int32_t* orow_lds = reinterpret_cast<const int32_t*>(smem + OROW_OFF);
    auto tab_off = [&](int row) -> size_t { return (size_t)rowoff[row - m0] + wcol0 + l31; };
    float* __restrict__ res = p.res;
    uint16_t* __restrict__ out0 = reinterpret_cast<uint16_t*>(p.out0);
    uint16_t* __restrict__ out0_lo = p.out0_lo;
    const bool ch_ok = ch < p.n_valid;
    const int res_mode = p.res_mode;
    if constexpr (I8) {
      // int8x2 planes out: pre-activation = (256*hh + cross) * w_scale[col] + bias, the gate value g in
      // (-1, 1) as q = round(g * CUBE_I8_ONE) = 256*hi + lo with hi in [-127, 127], lo in [-128, 127].
      // Transposed accumulators (see the MFMA calls): this lane owns activation row i*32 + l31 and, of the
      // wave's 32 channels, c = 8*g + 4*lh + e for register r = 4*g + e -- four consecutive channels per g,
      // i.e. one dword of each int8 plane.  The dwords go through wave-private LDS rows (32 bytes per
      // activation row and plane) so that the global stores are 16 bytes per lane.
      constexpr int SROW8 = 48;
      unsigned char* stg = smem + wid * (TM * 2 * 32 * SROW8);  // both row tiles of the wave are staged, then stored
      unsigned char* out8 = reinterpret_cast<unsigned char*>(p.out0);
      unsigned char* out8_lo = reinterpret_cast<unsigned char*>(p.out0_lo);
      const int srow = lane >> 1, schunk = lane & 1;
      const int chb = wcol0 / 2;  // first of the wave's 32 channels
      // group 0 came in under the last k-step (pre_*); groups 1..3 are requested here, all at once, and used in order
      float4 ws_a[4], ws_b[4], ba_all[TM][4], bb_all[TM][4];
      if constexpr (!EPI_PREFETCH) {
        pre_ws[0] = *reinterpret_cast<const float4*>(p.w_scale + wcol0 + 4 * lh);
        pre_ws[1] = *reinterpret_cast<const float4*>(p.w_scale + wcol0 + 32 + 4 * lh);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          pre_b[i][0] = *reinterpret_cast<const float4*>(bsrc_row[i] + 4 * lh);
          pre_b[i][1] = *reinterpret_cast<const float4*>(bsrc_row[i] + 32 + 4 * lh);
        }
      }
      ws_a[0] = pre_ws[0];
      ws_b[0] = pre_ws[1];
#pragma unroll
      for (int g = 1; g < 4; ++g) {
        ws_a[g] = *reinterpret_cast<const float4*>(p.w_scale + wcol0 + 8 * g + 4 * lh);
        ws_b[g] = *reinterpret_cast<const float4*>(p.w_scale + wcol0 + 32 + 8 * g + 4 * lh);
      }
      int orow_i[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        orow_i[i] = orow_lds[wrow0 - m0 + i * 32 + l31];  // this lane's output row (-1: none)
        ba_all[i][0] = pre_b[i][0];
        bb_all[i][0] = pre_b[i][1];
#pragma unroll
        for (int g = 1; g < 4; ++g) {
          ba_all[i][g] = *reinterpret_cast<const float4*>(bsrc_row[i] + 8 * g + 4 * lh);
          bb_all[i][g] = *reinterpret_cast<const float4*>(bsrc_row[i] + 32 + 8 * g + 4 * lh);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float wa[4] = {ws_a[g].x, ws_a[g].y, ws_a[g].z, ws_a[g].w}, wb[4] = {ws_b[g].x, ws_b[g].y, ws_b[g].z, ws_b[g].w};
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int orow_l = orow_i[i];
          const float4 *ba = ba_all[i], *bb = bb_all[i];
          const float xa[4] = {ba[g].x, ba[g].y, ba[g].z, ba[g].w}, xb[4] = {bb[g].x, bb[g].y, bb[g].z, bb[g].w};
          float gq[4];
          uint32_t tq[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            const float ca = fmaf((float)acc[i][0][r], (float)(1 << CUBE_I8_SHIFT), (float)acc2[i][0][r]) * wa[e] + xa[e];
            const float cb = fmaf((float)acc[i][1][r], (float)(1 << CUBE_I8_SHIFT), (float)acc2[i][1][r]) * wb[e] + xb[e];
            const float gv = tanhf_(ca) * sigmoidf_(cb);  // in [-1, 1]: |q| <= CUBE_I8_ONE without a clamp
            gq[e] = gv;
            // q = round-to-nearest-even(gv * CUBE_I8_ONE) in ONE fma: 1.5 * 2^23 + 128 + gv * 32512 has unit spacing, so the low
            // 16 bits of its bit pattern are t = q + 128 in two's complement; hi = t >> 8 is byte 1, lo = q - 256 * hi =
            // (t & 255) - 128 is byte 0 with its top bit flipped (mul + round + convert + clamp + shifts: 11 operations before)
            static_assert(CUBE_I8_SHIFT == 8, "the byte extraction below is written for 8 + 8 bits");
            tq[e] = __float_as_uint(fmaf(gv, (float)CUBE_I8_ONE, 12582912.0f + 128.0f));
          }
          // bytes 0 / 1 of the four values -> one dword each (v_perm_b32: selector 0-3 = bytes of the second operand, 4-7 = first)
          const uint32_t p01 = __builtin_amdgcn_perm(tq[1], tq[0], 0x05010400u), p23 = __builtin_amdgcn_perm(tq[3], tq[2], 0x05010400u);
          const uint32_t dl = __builtin_amdgcn_perm(p23, p01, 0x05040100u) ^ 0x80808080u;
          const uint32_t dh = __builtin_amdgcn_perm(p23, p01, 0x07060302u);
          unsigned char* stg_i = stg + i * (2 * 32 * SROW8);
          *reinterpret_cast<uint32_t*>(stg_i + l31 * SROW8 + 8 * g + 4 * lh) = dh;
          *reinterpret_cast<uint32_t*>(stg_i + 32 * SROW8 + l31 * SROW8 + 8 * g + 4 * lh) = dl;
          if constexpr (RES8) {  // word-level stack: fp32 residual stream (res_mode 0: = g, 1: += g) and the x + res side outputs
            const int c0 = chb + 8 * g + 4 * lh;  // four consecutive channels of this lane's row
            if (orow_l >= 0 && c0 < p.n_valid) {
              float* rp = res + (size_t)orow_l * p.ld_res + c0;
              float4 rs = res_mode == 0 ? float4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const float4*>(rp);
              rs.x += gq[0]; rs.y += gq[1]; rs.z += gq[2]; rs.w += gq[3];
              *reinterpret_cast<float4*>(rp) = rs;
              if (p.out1 || p.out2) {
                const float xs[4] = {rs.x + gq[0], rs.y + gq[1], rs.z + gq[2], rs.w + gq[3]};
                uint16_t h4[4], l4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split_bf16(xs[e], h4[e], l4[e]);
                const uint2 ph = {(uint32_t)h4[0] | ((uint32_t)h4[1] << 16), (uint32_t)h4[2] | ((uint32_t)h4[3] << 16)};
                const uint2 pl = {(uint32_t)l4[0] | ((uint32_t)l4[1] << 16), (uint32_t)l4[2] | ((uint32_t)l4[3] << 16)};
                if (p.out1) {
                  const size_t o = (size_t)orow_l * p.ldo1 + p.col1 + c0;
                  *reinterpret_cast<uint2*>(p.out1 + o) = ph;
                  if (p.out1_lo) *reinterpret_cast<uint2*>(p.out1_lo + o) = pl;
                }
                if (p.out2) {
                  const size_t o = (size_t)orow_l * p.ldo2 + p.col2 + c0;
                  *reinterpret_cast<uint2*>(p.out2 + o) = ph;
                  if (p.out2_lo) *reinterpret_cast<uint2*>(p.out2_lo + o) = pl;
                }
              }
            }
          }
        }
      }
      if (out8) {  // (uniform) int8x2 planes of the gate values for the next layer
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int cbase = wcol0 / 2 + schunk * 16;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const unsigned char* stg_i = stg + i * (2 * 32 * SROW8);
          const int row = orow_lds[wrow0 - m0 + i * 32 + srow];  // -1 = none
          if (row >= 0 && cbase < p.n_valid) {
            const size_t o = (size_t)row * p.ldo0 + p.col0 + cbase;  // bytes
            *reinterpret_cast<uint4*>(out8 + o) = *reinterpret_cast<const uint4*>(stg_i + srow * SROW8 + schunk * 16);
            *reinterpret_cast<uint4*>(out8_lo + o) = *reinterpret_cast<const uint4*>(stg_i + 32 * SROW8 + srow * SROW8 + schunk * 16);
          }
        }
      }
#ifdef CUBE_GEMM_PROF
      if (!RES8 && tid == 0 && blockIdx.x < 4096) {
        long long* q_ = g_gemm_prof + 4 * blockIdx.x;
        q_[0] = prof_t0; q_[1] = prof_t1; q_[2] = prof_t2; q_[3] = wall_clock64();
      }
#endif
      return;
    } else if (p.vec_out) {
      // no residual stream (character CNN): transpose each 32x32 gate tile through wave-private LDS
      // (the operand images are dead after the last barrier) and store 16 bytes per lane, 64-byte
      // row segments, instead of 2-byte scattered stores.
      constexpr int SROW = 80;  // bytes per staged row: 16-byte aligned, the two half-waves hit disjoint banks
      unsigned char* stg = smem + wid * (NP * 32 * SROW);
      const int srow = lane >> 2, schunk = lane & 3;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float ba[16], bb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (btab) {
            const size_t o = tab_off(row);
            ba[r] = btab[o];
            bb[r] = btab[o + 32];
          } else {
            ba[r] = bias_a;
            bb[r] = bias_b;
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lr = (r & 3) + 8 * (r >> 2) + 4 * lh;
          const float g = tanhf_(acc[i][0][r] + ba[r]) * sigmoidf_(acc[i][1][r] + bb[r]);
          uint16_t h, l = 0;
          if constexpr (SPLIT) split_bf16(g, h, l); else h = f32_to_bf16(g);
          *reinterpret_cast<uint16_t*>(stg + lr * SROW + l31 * 2) = h;
          if constexpr (SPLIT) *reinterpret_cast<uint16_t*>(stg + 32 * SROW + lr * SROW + l31 * 2) = l;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int cbase = wcol0 / 2 + schunk * 8;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int lr = srow + 16 * hh;
          const int row = orow_lds[wrow0 - m0 + i * 32 + lr];  // output row (ragged groups re-lay rows), -1 = none
          if (row >= 0 && cbase < p.n_valid) {
            const size_t o = (size_t)row * p.ldo0 + p.col0 + cbase;
            *reinterpret_cast<uint4*>(out0 + o) = *reinterpret_cast<const uint4*>(stg + lr * SROW + schunk * 16);
            if constexpr (SPLIT)
              *reinterpret_cast<uint4*>(out0_lo + o) = *reinterpret_cast<const uint4*>(stg + 32 * SROW + lr * SROW + schunk * 16);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      // all residual loads of this 32-row tile first (independent, in flight together), then the math
      float rv[16], ba[16], bb[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int orow = orow_lds[row - m0];
        rv[r] = (res && res_mode != 0 && ch_ok && orow >= 0) ? res[(size_t)orow * p.ld_res + ch] : 0.f;
        if (btab) {
          const size_t o = tab_off(row);
          ba[r] = btab[o];
          bb[r] = btab[o + 32];
        } else {
          ba[r] = bias_a;
          bb[r] = bias_b;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = orow_lds[wrow0 - m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
        if (row >= 0 && ch_ok) {
          const float g = tanhf_(acc[i][0][r] + ba[r]) * sigmoidf_(acc[i][1][r] + bb[r]);
          if (out0 && res_mode != 3) {
            if (p.out0_i8) {  // the next layer reads int8x2 planes (cube_gemm_i8x2): q = round(g * CUBE_I8_ONE) = 256*hi + lo
              int q = __float2int_rn(g * (float)CUBE_I8_ONE);
              q = q > CUBE_I8_ONE ? CUBE_I8_ONE : (q < -CUBE_I8_ONE ? -CUBE_I8_ONE : q);
              const int h = (q + (1 << (CUBE_I8_SHIFT - 1))) >> CUBE_I8_SHIFT;
              const size_t o = (size_t)row * p.ldo0 + p.col0 + ch;
              reinterpret_cast<signed char*>(out0)[o] = (signed char)h;
              reinterpret_cast<signed char*>(out0_lo)[o] = (signed char)(q - (h << CUBE_I8_SHIFT));
            } else {
              store_bf16(out0, out0_lo, (size_t)row * p.ldo0 + p.col0 + ch, g);
            }
          }
          if (!res) continue;  // plain gated layer: no residual stream
          const float rs = rv[r] + (res_mode == 2 ? 2.0f * g : g);
          res[(size_t)row * p.ld_res + ch] = rs;
          if (out0 && res_mode == 3) store_bf16(out0, out0_lo, (size_t)row * p.ldo0 + p.col0 + ch, rs);
          if (p.out1) store_bf16(p.out1, p.out1_lo, (size_t)row * p.ldo1 + p.col1 + ch, rs + g);
          if (p.out2) store_bf16(p.out2, p.out2_lo, (size_t)row * p.ldo2 + p.col2 + ch, rs + g);
        }
      }
    }
  } else if constexpr (EPI == CUBE_EPI_LSTM) {
    // The four gate pre-activations of a hidden unit sit in two NEIGHBOURING lanes (cube_hip.h: columns packed
    // [i f i f ..] in the wave's first 32-column block, [g o g o ..] in its second): the even lane holds i and g of
    // its 16 rows, the odd lane f and o of the same rows.  They swap half of them -- the even lane takes rows r < 8,
    // the odd lane rows r >= 8 -- and each then runs eight cell updates with everything in registers.
    static_assert(TN == 2, "the LSTM epilogue pairs the two 32-column blocks of a wave");
    const bool odd = l31 & 1;
    const int unit = (wcol0 >> 6) * 16 + (l31 >> 1);
    const bool unit_ok = unit < p.n_valid;
    const float b0 = p.bias ? p.bias[wcol0 + l31] : 0.f, b1 = p.bias ? p.bias[wcol0 + 32 + l31] : 0.f;
    float* __restrict__ cst = p.res;
    uint16_t* __restrict__ h_hi = reinterpret_cast<uint16_t*>(p.out0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float mine0[8], mine1[8], oth0[8], oth1[8], cold[8];
      int rows[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = odd ? k + 8 : k;
        const int row = wrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        rows[k] = (row < p.M && unit_ok) ? row : -1;
        cold[k] = rows[k] >= 0 ? cst[(size_t)row * p.ld_res + unit] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float v0k = acc[i][0][k] + b0, v0k8 = acc[i][0][k + 8] + b0;
        const float v1k = acc[i][1][k] + b1, v1k8 = acc[i][1][k + 8] + b1;
        mine0[k] = odd ? v0k8 : v0k;   // what this lane keeps: its own gates of the rows it updates
        mine1[k] = odd ? v1k8 : v1k;
        oth0[k] = __shfl_xor(odd ? v0k : v0k8, 1, 64);   // ... and the neighbour's gates of those rows
        oth1[k] = __shfl_xor(odd ? v1k : v1k8, 1, 64);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (rows[k] < 0) continue;
        float cn, hn;
        if (odd) lstm_cell_update(oth0[k], mine0[k], oth1[k], mine1[k], cold[k], cn, hn);
        else lstm_cell_update(mine0[k], oth0[k], mine1[k], oth1[k], cold[k], cn, hn);
        cst[(size_t)rows[k] * p.ld_res + unit] = cn;
        store_bf16(h_hi, p.out0_lo, (size_t)rows[k] * p.ldo0 + p.col0 + unit, hn);
        if (p.out1) store_bf16(p.out1, p.out1_lo, (size_t)rows[k] * p.ldo1 + p.col1 + unit, cn);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = wcol0 + j * 32 + l31;
      if (col >= p.n_valid) continue;
      const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
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

// row tiles of a launch with tile height BM / slab SR (-1: a sequence does not fit the slab); `ct`: the per-group table
inline int count_row_tiles(int M, const ConvGroups& cg, int BM, int SR, ConvTiles* ct = nullptr) {
  if (cg.n <= 0) return (M + BM - 1) / BM;
  int tiles = 0;
  for (int g = 0; g < cg.n; ++g) {
    const int bm = tile_rows(BM, SR, cg.seq_in[g], cg.seq_out[g]);
    if (bm < 1) return -1;
    const int n_out = cg.n_rows[g] / cg.seq_in[g] * cg.seq_out[g];
    if (ct) { ct->tile0[g] = tiles; ct->bm[g] = bm; ct->n_out[g] = n_out; }
    tiles += (n_out + bm - 1) / bm;
  }
  if (ct) ct->tile0[cg.n] = tiles;
  return tiles;
}

constexpr int NUM_CUS = 256;

template <int EPI, bool SPLIT, class C, bool I8 = false, bool RES8 = false>
int launch_cfg(GemmParams p, const ConvGroups& cg, hipStream_t st) {
  constexpr int lds = lds_bytes<C, SPLIT>();
  static CubeLdsReservation reserved;
  if (!reserved.ensure((const void*)gemm_kernel<EPI, SPLIT, C, I8, RES8>, lds))
    CUBE_FAIL(CUBE_ERR_LAUNCH, "cube_gemm_bf16: cannot reserve %d bytes of LDS", lds);
  p.n_tiles = p.N / C::BN;
  ConvTiles ct = {};
  p.m_tiles = count_row_tiles(p.M, cg, C::BM, C::SR, &ct);
  if (p.m_tiles < 0) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: a sequence group does not fit the activation slab");
  dim3 grid(p.n_tiles * ((p.m_tiles + 7) / 8 * 8), p.batch > 1 ? p.batch : 1);
  hipLaunchKernelGGL((gemm_kernel<EPI, SPLIT, C, I8, RES8>), grid, dim3(64 * C::WM * C::WN), lds, st, p, cg, ct);
  return CUBE_OK;
}

template <int EPI, bool SPLIT>
int launch_split(const GemmParams& p, const ConvGroups& cg, hipStream_t st) {
  // the big tiles need enough row tiles to fill the 256 CUs
  const long nb = p.batch > 1 ? p.batch : 1;  // problems in the grid
  const bool big = (p.N % CfgBig::BN == 0) && ((long)(p.M / CfgBig::BM) * (p.N / CfgBig::BN) * nb >= 192);
  if (!big) {
    if constexpr (EPI == CUBE_EPI_GATE) {
      // a convolution whose 128-row tiles are a little more than one block per CU: 160-row tiles, one round
      const long nt = p.N / CfgSmall160::BN;
      const long t160 = count_row_tiles(p.M, cg, CfgSmall160::BM, CfgSmall160::SR);
      if (p.taps > 1 && t160 > 0 && t160 * nt <= NUM_CUS &&
          (long)count_row_tiles(p.M, cg, CfgSmall::BM, CfgSmall::SR) * nt > NUM_CUS * 3 / 4)
        return launch_cfg<EPI, SPLIT, CfgSmall160>(p, cg, st);
    }
    return launch_cfg<EPI, SPLIT, CfgSmall>(p, cg, st);
  }
  // one block per CU: time ~ rounds x tile height (x a small penalty for the shorter tiles, whose
  // weight-tile traffic and fragment reads are spread over fewer rows)
  const long nt = p.N / CfgBig::BN;
  auto cost = [&](int BM, int SR, double penalty) {
    const int t = count_row_tiles(p.M, cg, BM, SR);
    return t < 0 ? 1e30 : (double)((t * nt * nb + NUM_CUS - 1) / NUM_CUS) * BM * penalty;
  };
  // The gate epilogue of the 256-row tile does not fit the registers (576 bytes of scratch per lane; with the residual
  // stream and the side outputs of the word-level stack a launch took 1.17 ms where the 192-row tile takes 0.3): the
  // gated convolutions take the 192-row tile whenever the big tiles are chosen.
  if (EPI == CUBE_EPI_GATE || cost(CfgBig192::BM, CfgBig192::SR, 1.04) < cost(CfgBig::BM, CfgBig::SR, 1.0))
    return launch_cfg<EPI, SPLIT, CfgBig192>(p, cg, st);
  return launch_cfg<EPI, SPLIT, CfgBig>(p, cg, st);
}

template <int EPI>
int launch(const GemmParams& p, const ConvGroups& cg, hipStream_t st) {
  return p.A_lo ? launch_split<EPI, true>(p, cg, st) : launch_split<EPI, false>(p, cg, st);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// both entry points: w_scale != NULL selects int8x2 operands
int gemm_entry(const cube_gemm_args* a, const float* w_scale, void* stream) {
  const bool i8 = w_scale != nullptr;
  const int BKC = i8 ? 2 * BK : BK;  // channels per 64-byte LDS row
  const char* fn = i8 ? "cube_gemm_i8x2" : "cube_gemm_bf16";
  if (!a) CUBE_FAIL(CUBE_ERR_ARG, "%s: null args", fn);
  if (a->M <= 0) return CUBE_OK;
  if (!a->A || !a->W) CUBE_FAIL(CUBE_ERR_ARG, "%s: null operand", fn);
  if ((a->A_lo == nullptr) != (a->W_lo == nullptr))
    CUBE_FAIL(CUBE_ERR_ARG, "%s: A_lo and W_lo must be given together (bf16x3) or not at all (bf16)", fn);
  if (a->taps != 1 && a->taps != 3 && a->taps != 5) CUBE_FAIL(CUBE_ERR_ARG, "%s: taps=%d", fn, a->taps);
  if (i8 && a->taps != 5) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_i8x2: taps=%d (the int8x2 tile is built for Conv1d(k=5))", a->taps);
  if (a->Kc <= 0 || a->Kc % BKC) CUBE_FAIL(CUBE_ERR_ARG, "%s: Kc=%d not a multiple of %d", fn, a->Kc, BKC);
  if (a->N <= 0 || a->N % 128) CUBE_FAIL(CUBE_ERR_ARG, "%s: N=%d not a multiple of 128", fn, a->N);
  if (a->lda < a->Kc || a->lda % (i8 ? 16 : 8)) CUBE_FAIL(CUBE_ERR_ALIGN, "%s: lda=%d (Kc=%d)", fn, a->lda, a->Kc);
  if (i8 && (a->epilogue != CUBE_EPI_GATE || !a->A_lo || !a->W_lo || (a->out0 == nullptr) != (a->out0_lo == nullptr) || a->out0_int8))
    CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_i8x2: int8x2 needs both planes of A and W (and of out0, if given) and the gate epilogue");
  if (i8 && a->res) {
    if (a->res_mode != 0 && a->res_mode != 1) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_i8x2: res_mode=%d (0 or 1)", a->res_mode);
    if (a->ld_res % 4 || !aligned16(a->res) || a->ldo1 % 4 || a->col1 % 4 || a->ldo2 % 4 || a->col2 % 4 ||
        ((uintptr_t)a->out1 | (uintptr_t)a->out1_lo | (uintptr_t)a->out2 | (uintptr_t)a->out2_lo) % 8)
      CUBE_FAIL(CUBE_ERR_ALIGN, "cube_gemm_i8x2: res / out1 / out2 rows are accessed four channels at a time (ld, col multiples of 4)");
  }
  if (!i8 && a->out0_int8 && (a->epilogue != CUBE_EPI_GATE || !a->res || a->res_mode == 3 || !a->out0 || !a->out0_lo))
    CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_bf16: out0_int8 needs the gate epilogue with a residual stream (res_mode 0..2) and both out0 planes");
  if (!aligned16(a->A) || !aligned16(a->W) || !aligned16(a->A_lo) || !aligned16(a->W_lo))
    CUBE_FAIL(CUBE_ERR_ALIGN, "%s: operands must be 16-byte aligned", fn);
  ConvGroups cg;
  cg.n = 0;
  if (a->groups) {
    cg = *a->groups;
    if (cg.n <= 0 || cg.n > CUBE_MAX_CONV_GROUPS) CUBE_FAIL(CUBE_ERR_ARG, "%s: %d sequence groups (1..%d)", fn, cg.n, CUBE_MAX_CONV_GROUPS);
    if (a->epilogue != CUBE_EPI_GATE || a->res || a->rowmap)
      CUBE_FAIL(CUBE_ERR_ARG, "%s: sequence groups need the gate epilogue without residual stream / rowmap", fn);
    for (int g = 0; g < cg.n; ++g) {
      if (cg.seq_in[g] <= 0 || cg.n_rows[g] <= 0 || cg.n_rows[g] % cg.seq_in[g] || cg.seq_out[g] <= 0 ||
          cg.seq_out[g] > cg.seq_in[g] || cg.row0_in[g] < 0 || cg.row0_in[g] + cg.n_rows[g] > a->M || cg.row0_out[g] < 0 ||
          cg.word0[g] < 0)
        CUBE_FAIL(CUBE_ERR_ARG, "%s: bad sequence group %d", fn, g);
    }
  }
  const int seq = (a->taps == 1 || a->groups) ? a->M : a->seq_len;
  if (seq <= 0 || a->M % seq) CUBE_FAIL(CUBE_ERR_ARG, "%s: M=%d not a multiple of seq_len=%d", fn, a->M, seq);
  GemmParams p;
  p.vec_out = 0; p.m_tiles = p.n_tiles = 0;
  p.out0_i8 = a->out0_int8;
  p.batch = a->batch > 1 ? a->batch : 1; p.bs_a = (long)a->stride_A; p.bs_w = (long)a->stride_W; p.bs_bias = a->stride_bias;
  p.bs_col0 = a->stride_col0;
  if (p.batch > 1) {
    if (i8 || a->epilogue == CUBE_EPI_GATE || a->taps != 1 || a->rowmap || a->groups || a->accumulate || p.batch > 65535)
      CUBE_FAIL(CUBE_ERR_ARG, "%s: batch=%d is for plain Linear layers (taps 1, no gate epilogue / rowmap / groups / accumulate)", fn, a->batch);
    if (a->stride_A % 8 || a->stride_W % 8 || a->stride_A < 0 || a->stride_W < 0 || a->stride_bias < 0 || a->stride_col0 < 0)
      CUBE_FAIL(CUBE_ERR_ALIGN, "%s: batch strides of A and W must be non-negative multiples of 8 elements", fn);
    if (a->ldo0 < a->col0 + (p.batch - 1) * a->stride_col0 + (a->n_valid > 0 ? a->n_valid : a->N))
      CUBE_FAIL(CUBE_ERR_ARG, "%s: ldo0=%d does not hold %d problems at column stride %d", fn, a->ldo0, p.batch, a->stride_col0);
  }
  p.bias_tab = a->bias_tab; p.row_group = a->row_group;
  if ((a->bias_tab == nullptr) != (a->row_group == nullptr))
    CUBE_FAIL(CUBE_ERR_ARG, "%s: bias_tab and row_group go together", fn);
  if (a->bias_tab && (a->epilogue != CUBE_EPI_GATE || a->taps > 5 || a->n_groups <= 0))
    CUBE_FAIL(CUBE_ERR_ARG, "%s: bias_tab needs the gate epilogue, taps <= 5 and n_groups > 0", fn);
  p.w_scale = w_scale;
  // int8 planes: the kernel counts in 2-byte units (its addressing is then the bf16 one, byte for byte)
  p.A = (const uint16_t*)a->A; p.A_lo = (const uint16_t*)a->A_lo; p.lda = i8 ? a->lda / 2 : a->lda; p.M = a->M; p.seq_len = seq;
  p.taps = a->taps; p.Kc = i8 ? a->Kc / 2 : a->Kc;
  p.W = (const uint16_t*)a->W; p.W_lo = (const uint16_t*)a->W_lo; p.N = a->N; p.bias = a->bias;
  p.n_valid = a->n_valid > 0 ? a->n_valid : (a->epilogue == CUBE_EPI_GATE ? a->N / 2 : a->N);
  p.accumulate = a->accumulate;
  p.out0 = a->out0; p.out0_lo = (uint16_t*)a->out0_lo; p.ldo0 = a->ldo0; p.col0 = a->col0; p.rowmap = a->rowmap;
  p.res = a->res; p.ld_res = a->ld_res; p.res_mode = a->res_mode;
  p.out1 = (uint16_t*)a->out1; p.out1_lo = (uint16_t*)a->out1_lo; p.ldo1 = a->ldo1; p.col1 = a->col1;
  p.out2 = (uint16_t*)a->out2; p.out2_lo = (uint16_t*)a->out2_lo; p.ldo2 = a->ldo2; p.col2 = a->col2;
  if (p.n_valid > a->N) CUBE_FAIL(CUBE_ERR_ARG, "%s: n_valid=%d > N=%d", fn, p.n_valid, a->N);
  hipStream_t st = (hipStream_t)stream;
  int rc = CUBE_OK;
  switch (a->epilogue) {
    case CUBE_EPI_F32:
      if (!a->out0 || a->ldo0 < a->col0 + p.n_valid) CUBE_FAIL(CUBE_ERR_ARG, "%s: bad out0/ldo0", fn);
      rc = launch<CUBE_EPI_F32>(p, cg, st);
      break;
    case CUBE_EPI_BF16:
    case CUBE_EPI_BF16_TANH:
    case CUBE_EPI_BF16_RELU:
      if (!a->out0 || a->ldo0 < a->col0 + p.n_valid) CUBE_FAIL(CUBE_ERR_ARG, "%s: bad out0/ldo0", fn);
      if (a->epilogue == CUBE_EPI_BF16) rc = launch<CUBE_EPI_BF16>(p, cg, st);
      else if (a->epilogue == CUBE_EPI_BF16_TANH) rc = launch<CUBE_EPI_BF16_TANH>(p, cg, st);
      else rc = launch<CUBE_EPI_BF16_RELU>(p, cg, st);
      break;
    case CUBE_EPI_LSTM:
      // n_valid = hidden units; N >= 4 * units rounded up to whole 64-column wave ranges (16 units each)
      if (a->n_valid <= 0 || (a->n_valid + 15) / 16 * 64 > a->N) CUBE_FAIL(CUBE_ERR_ARG, "%s: LSTM epilogue: n_valid=%d units need N >= %d", fn, a->n_valid, (a->n_valid + 15) / 16 * 64);
      if (a->taps != 1 || a->rowmap || a->groups || a->accumulate || p.batch > 1 || a->bias_tab)
        CUBE_FAIL(CUBE_ERR_ARG, "%s: LSTM epilogue is for one plain Linear problem (taps 1, no rowmap / groups / accumulate / batch)", fn);
      if (!a->res || a->ld_res < a->n_valid) CUBE_FAIL(CUBE_ERR_ARG, "%s: LSTM epilogue needs the cell state res[M, >= %d]", fn, a->n_valid);
      if (!a->out0 || a->ldo0 < a->col0 + a->n_valid) CUBE_FAIL(CUBE_ERR_ARG, "%s: bad out0/ldo0", fn);
      if (a->out1 && a->ldo1 < a->col1 + a->n_valid) CUBE_FAIL(CUBE_ERR_ARG, "%s: bad out1/ldo1", fn);
      {  // h (and the bf16 copy of c) must not alias the columns of A that other workgroups still read
        const char* a0 = (const char*)a->A;
        const char* a1 = a0 + 2 * ((size_t)(a->M - 1) * a->lda + a->Kc);
        auto clash = [&](const void* out, int col, int ld) {
          if (!out) return false;
          const char* o0 = (const char*)out + 2 * (size_t)col;
          const char* o1 = o0 + 2 * ((size_t)(a->M - 1) * ld + a->n_valid);
          if (!(o0 < a1 && a0 < o1)) return false;   // disjoint address ranges
          if (ld != a->lda) return true;              // interleaved with another row pitch: refuse
          const ptrdiff_t d = (o0 - a0) / 2;          // same pitch: the column range of the output against [0, Kc)
          const ptrdiff_t c0 = ((d % ld) + ld) % ld;
          return c0 < a->Kc || c0 + a->n_valid > ld;
        };
        if (clash(a->out0, a->col0, a->ldo0) || clash(a->out0_lo, a->col0, a->ldo0) || clash(a->out1, a->col1, a->ldo1) ||
            clash(a->out1_lo, a->col1, a->ldo1))
          CUBE_FAIL(CUBE_ERR_ARG, "%s: LSTM epilogue: out0 / out1 overlap the columns of A this launch reads (use a second buffer)", fn);
      }
      rc = launch<CUBE_EPI_LSTM>(p, cg, st);
      break;
    case CUBE_EPI_GATE: {
      const int half = p.n_valid;  // valid channels; N/2 >= half when padded to the 128-column tile
      if (half > a->N / 2) CUBE_FAIL(CUBE_ERR_ARG, "%s: gate n_valid=%d > N/2", fn, half);
      if (a->res && a->ld_res < half) CUBE_FAIL(CUBE_ERR_ARG, "%s: gate epilogue needs res[M, >=%d]", fn, half);
      if (!a->res && (a->out1 || a->out2)) CUBE_FAIL(CUBE_ERR_ARG, "%s: out1/out2 (x + res) need res", fn);
      if (!a->res && !a->out0) CUBE_FAIL(CUBE_ERR_ARG, "%s: gate epilogue without any output", fn);
      if (a->res_mode < 0 || a->res_mode > 3 || (a->res_mode == 3 && !a->res))
        CUBE_FAIL(CUBE_ERR_ARG, "%s: res_mode=%d", fn, a->res_mode);
      if (a->rowmap) CUBE_FAIL(CUBE_ERR_ARG, "%s: rowmap is not supported by the gate epilogue", fn);
      if (a->out0 && a->ldo0 < a->col0 + half) CUBE_FAIL(CUBE_ERR_ARG, "%s: bad out0/ldo0", fn);
      if (a->out1 && a->ldo1 < a->col1 + half) CUBE_FAIL(CUBE_ERR_ARG, "%s: bad out1/ldo1", fn);
      if (a->out2 && a->ldo2 < a->col2 + half) CUBE_FAIL(CUBE_ERR_ARG, "%s: bad out2/ldo2", fn);
      if (!a->bias) CUBE_FAIL(CUBE_ERR_ARG, "%s: gate epilogue needs a bias", fn);
      p.vec_out = !a->res && a->out0 && half % 8 == 0 && a->ldo0 % 8 == 0 && a->col0 % 8 == 0 && aligned16(a->out0) &&
                  aligned16(a->out0_lo);
      if (i8) {
        if (half % 16 || (a->out0 && (a->ldo0 % 16 || a->col0 % 16 || !aligned16(a->out0) || !aligned16(a->out0_lo))))
          CUBE_FAIL(CUBE_ERR_ALIGN, "cube_gemm_i8x2: out0 planes need 16-byte aligned rows (half, ldo0, col0 multiples of 16)");
        if (!aligned16(w_scale) || !aligned16(a->bias) || !aligned16(a->bias_tab))
          CUBE_FAIL(CUBE_ERR_ALIGN, "cube_gemm_i8x2: w_scale, bias and bias_tab must be 16-byte aligned (read four floats at a time)");
        rc = a->res ? launch_cfg<CUBE_EPI_GATE, true, CfgI8, true, true>(p, cg, st)
                    : launch_cfg<CUBE_EPI_GATE, true, CfgI8, true, false>(p, cg, st);
        break;
      }
      rc = launch<CUBE_EPI_GATE>(p, cg, st);
      break;
    }
    default:
      CUBE_FAIL(CUBE_ERR_ARG, "%s: unknown epilogue %d", fn, a->epilogue);
  }
  if (rc != CUBE_OK) return rc;
  CUBE_CHECK_LAUNCH(fn);
  return CUBE_OK;
}

}  // namespace

extern "C" int cube_gemm_bf16(const cube_gemm_args* a, void* stream) {
  CUBE_ENTRY();
  return gemm_entry(a, nullptr, stream);
}

extern "C" int cube_gemm_i8x2(const cube_gemm_args* a, const float* w_scale, void* stream) {
  CUBE_ENTRY();
  if (!w_scale) CUBE_FAIL(CUBE_ERR_ARG, "cube_gemm_i8x2: null w_scale");
  return gemm_entry(a, w_scale, stream);
}

#ifdef CUBE_GEMM_PROF
extern "C" int cube_gemm_prof(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_prof), sizeof(long long) * 4 * 4096) == hipSuccess ? 0 : 1;
}
#endif

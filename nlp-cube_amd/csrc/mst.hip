// Chu-Liu-Edmonds single-root head decoding, one wavefront per task, score matrix in LDS.
//
// Restates the reference decoder (cube/networks/utils.py:400-577) with the same fp64
// arithmetic and the same tie-breaks (first maximal index in the current node order; the
// cycle whose smallest node comes last), so heads are identical wherever the reference's own
// comparisons are not decided by the last bits of a log-sum.
//
//   mst_first_kernel   one wave per sentence: builds S (fp64, n x n, odd leading dimension ->
//                      conflict-free row-per-lane scans) from the padded-softmax probabilities,
//                      runs the contraction loop, and either emits the heads (exactly one root)
//                      or appends one task per candidate root to a device-side task list.
//   mst_root_kernel    one wave per (sentence, candidate root): the reference's `set_root` +
//                      CLE + log-probability score.  Grid-strides over the device-side task
//                      count: no host round trip.  Two launches: the first runs, per sentence, the
//                      candidate with the largest score BOUND (every word takes its best head; or all
//                      candidates when there are at most MST_BOUND_MIN_CAND), the second the other
//                      candidates whose bound is not below that tree's exact score.  The skipped ones
//                      cannot be the reference's maximum, so the chosen heads are unchanged.
//   mst_select_kernel  first strictly-best candidate per sentence (utils.py:548-555), or
//                      status 1 when every candidate scores -inf (the reference asserts).
//
// Contraction keeps node ids (= LDS rows/columns) in place: a logical order list gives the
// reference's index order (non-cycle nodes in order, contracted node last), the contracted node
// reuses the slot of the cycle's first node, and every level appends a byte record
// [m, c, cycle nodes, their heads, best-cycle-member per outside head, best-cycle-member per
// outside dependent] that the expansion pass replays backwards.  Cycles are found with 8 rounds
// of pointer doubling instead of a recursive Tarjan.
#include "common.h"
#include "../../include/cube_hip.h"

namespace {

// A task is ONE wavefront, and a workgroup holds several independent tasks (below): all synchronisation is
// wave-level.  The workgroup-scope fence makes the wave's LDS and global writes (the score matrix lives in the
// workspace for long sentences) visible to its own later reads; no s_barrier, which would tie tasks of different
// lengths together.
#define MST_SYNC()                                            \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");    \
    __builtin_amdgcn_wave_barrier();                          \
  } while (0)
// Tasks (waves) per workgroup.  The decode runs beside the 160-KiB-LDS GEMM tiles of the next batches: with one task
// per workgroup the 256 first-pass workgroups land on 256 different CUs, 11 KB of LDS each, and no CU can take a GEMM
// tile until they are done.  Packed, the same tasks sit on B / 14 CUs (L = 32) and leave the rest of the chip alone.
constexpr int MST_MAX_WAVES = 16;
// the rooted re-runs need ~160 registers: 12 waves per workgroup (three per SIMD, 168 registers each) keep them out of scratch
// memory -- with 16 the compiler had 128 and spilled 40 bytes per lane, a private-segment allocation at first dispatch
constexpr int MST_ROOT_MAX_WAVES = 12;
constexpr int MST_ROOT_MAX_BLOCKS = 96;  // CUs the rooted re-runs may spread over when tasks are packed
constexpr int MST_BOUND_MIN_CAND = 4;    // up to this many candidate roots of a sentence all run in the first round (mst_first_kernel)

constexpr int MST_MAX_N = 130;   // L <= 129: S (n*n fp64) + records fit the 160 KiB LDS of one CU
constexpr int MST_MAX_N_GLOBAL = 255;  // longer sentences, node ids still bytes: S lives in the workspace (HBM/L2), slow but exact
constexpr int MST_MAX_N_LONG = 2048;   // beyond 254 words: 16-bit node ids, S and the contraction records in the workspace
constexpr size_t MST_LONG_ROOT_BYTES = (size_t)2 << 30;  // workspace the rooted re-runs of such a batch may take
constexpr int MST_GLOBAL_ROOT_BLOCKS = 3072;  // workgroups (= private score matrices) of the root kernel in that mode: 12 per CU
// (its contraction records live in the workspace too, so a task holds 6 KB of LDS and the ~150 VGPRs of the kernel set the limit)
// The rooted re-runs (one task per candidate root, ~17 per sentence when most first trees are not
// single-rooted) switch to workspace-resident score matrices earlier: from this size on an LDS-resident
// matrix leaves one or two tasks per CU, and a task is bound by its own dependent-access latency, not
// by where S lives (129 nodes: 1088 tasks in 5 rounds of 3 ms with S in LDS, all at once in 5.5 ms
// with S in the workspace; the first pass, one task per sentence, stays in LDS)
constexpr int MST_ROOT_GLOBAL_N = 96;  // workgroups (= private score matrices) of the root kernel in that mode

template <typename IdT>  // node ids: bytes up to 254 words, 16 bits beyond
struct MstLdsT {
  double* S; int ld;
  double* hval; double* dval; double* cs; double* bval;
  IdT *order, *pos, *tree, *p0, *p1, *mn0, *mn1, *flag, *incyc, *alive, *clist, *rankn, *dlist, *tie;
  IdT* rec; int rec_cap;
};

__host__ __device__ inline int mst_np(int n) { return (n + 7) & ~7; }
__host__ __device__ inline size_t mst_rec_cap(int n) { return (size_t)n * n + 3 * (size_t)n + 16; }
// LDS of one task; `id` = bytes per node id; 2-byte ids keep their contraction records in the workspace as well
__host__ __device__ inline size_t mst_lds_bytes(int n, bool global_s = false, int id = 1, bool global_rec = false) {
  const int ld = n | 1;
  const int np = mst_np(n);
  return (global_s ? 0 : (size_t)n * ld * 8) + (size_t)4 * np * 8 + (size_t)14 * np * id + (global_rec ? 0 : id * mst_rec_cap(n));
}

template <typename IdT>
__device__ inline MstLdsT<IdT> mst_carve(unsigned char* smem, int n, double* global_s = nullptr, IdT* global_rec = nullptr) {
  MstLdsT<IdT> m;
  const int np = mst_np(n);
  m.ld = n | 1;
  m.S = global_s ? global_s : reinterpret_cast<double*>(smem);
  unsigned char* q = smem + (global_s ? 0 : (size_t)n * m.ld * 8);
  m.hval = reinterpret_cast<double*>(q); q += np * 8;
  m.dval = reinterpret_cast<double*>(q); q += np * 8;
  m.cs = reinterpret_cast<double*>(q); q += np * 8;
  m.bval = reinterpret_cast<double*>(q); q += np * 8;
  IdT** arr[14] = {&m.order, &m.pos, &m.tree, &m.p0, &m.p1, &m.mn0, &m.mn1, &m.flag, &m.incyc, &m.alive, &m.clist, &m.rankn, &m.dlist, &m.tie};
  for (int i = 0; i < 14; ++i) { *arr[i] = reinterpret_cast<IdT*>(q); q += np * sizeof(IdT); }
  m.rec = global_rec ? global_rec : reinterpret_cast<IdT*>(q);
  m.rec_cap = (int)mst_rec_cap(n);
  return m;
}

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

// First maximum of one row over the current nodes in logical order, exactly like numpy.argmax (NaN counts as
// maximal, the first one wins).  The two dependent reads per entry (order list, then the score) are issued eight
// entries at a time: a one-entry-at-a-time loop pays the full latency per entry.  `bv` = the maximum, `tie` = some
// other entry equals it (or it is NaN): see cle_run, step 1.
template <typename IdT>
__device__ __forceinline__ int scan_row_exact(const MstLdsT<IdT>& m, const double* row, int cur, double& bv, int& tie) {
  double best = row[m.order[0]];
  int bi = m.order[0], t = 0;
  for (int l0 = 1; l0 < cur; l0 += 8) {
    int j[8];
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) j[u] = m.order[min(l0 + u, cur - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = row[j[u]];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (l0 + u < cur) {
        if (v[u] > best || (v[u] != v[u] && best == best)) { best = v[u]; bi = j[u]; t = 0; }
        else if (v[u] == best) t = 1;
      }
  }
  bv = best;
  tie = (t || best != best) ? 1 : 0;
  return bi;
}

// The same for a matrix without NaNs (cle_run keeps track): one ordered compare per entry, and the order list is
// read eight bytes at a time (it is 8-byte aligned and padded to a multiple of 8 bytes).
template <bool MASK, typename IdT>
__device__ __forceinline__ void scan8(const MstLdsT<IdT>& m, const double* row, int l0, int cur, double& best, int& bi, int& t) {
  constexpr int PW = 8 / sizeof(IdT);  // ids per 8-byte word
  uint64_t w[sizeof(IdT)];
#pragma unroll
  for (int i = 0; i < (int)sizeof(IdT); ++i) w[i] = reinterpret_cast<const uint64_t*>(m.order + l0)[i];
  int j[8];
  double v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    j[u] = (int)((w[u / PW] >> (8 * sizeof(IdT) * (u % PW))) & (sizeof(IdT) == 1 ? 0xffu : 0xffffu));
    if (MASK && !(l0 + u >= 1 && l0 + u < cur)) j[u] = 0;  // bytes past the list are stale: read a slot that exists
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = row[j[u]];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const bool ok = !MASK || (l0 + u >= 1 && l0 + u < cur);
    const bool upd = ok && v[u] > best;
    const bool eq = ok && v[u] == best;
    t = upd ? 0 : (t | (eq ? 1 : 0));
    best = upd ? v[u] : best;
    bi = upd ? j[u] : bi;
  }
}

template <typename IdT>
__device__ __forceinline__ int scan_row(const MstLdsT<IdT>& m, const double* row, int cur, bool has_nan, double& bv, int& tie) {
  if (has_nan) return scan_row_exact(m, row, cur, bv, tie);
  double best = row[m.order[0]];
  int bi = m.order[0], t = 0;
  scan8<true, IdT>(m, row, 0, cur, best, bi, t);  // entry 0 is the start value: masked out
  int l0 = 8;
  for (; l0 + 8 <= cur; l0 += 8) scan8<false, IdT>(m, row, l0, cur, best, bi, t);
  if (l0 < cur) scan8<true, IdT>(m, row, l0, cur, best, bi, t);
  bv = best;
  tie = t;
  return bi;
}

// Runs the contraction/expansion on the n x n matrix in `m.S` (row = dependent, column = head;
// row 0 is virtual: e_0).  Diagonal entries must be zero on entry.  Result: m.tree[k] = head.
// Returns false if the record area overflowed (cannot happen for n <= MST_MAX_N).
// `has_nan`: the matrix holds a NaN (wave-uniform; build_scores reports it, step 5 keeps it up to date).
template <typename IdT>
__device__ bool cle_run(const MstLdsT<IdT>& m, int n, int lane, bool has_nan) {
  double* S = m.S;
  const int ld = m.ld;
  for (int k = lane; k < n; k += 64) { m.order[k] = (IdT)k; m.pos[k] = (IdT)k; m.alive[k] = 1; }
  MST_SYNC();
  int cur = n;       // nodes at this level
  int rec_end = 0;   // bytes of records written
  int levels = 0;
  int ndirty = 0;    // rows whose best head must be re-scanned at the next level
  while (true) {
    // ---- 1. best head per node, first maximum in logical order.  First level: every row, one row per lane, eight
    // entries in flight per lane.  Later levels: a row's arg-max can only change if its best head was a node of the
    // contracted cycle (for every other row the old best head survives, keeps its value and still precedes every
    // equal entry: the survivors keep their relative order and the contracted node goes last).  Such a row's maximum
    // V was attained inside the cycle, so its entry for the contracted node -- the largest of its entries into the
    // cycle -- is V again and every surviving entry is <= V: the new best head is the first survivor equal to V or,
    // if there is none, the contracted node.  `tie[k]` says whether row k holds a second entry equal to its maximum
    // (kept conservatively: set by the scans and when a new entry equals the maximum, never cleared by removals), so
    // the common case needs no scan at all (step 6 re-points the row) and only rows with a tie are scanned again
    // (`dlist`).  The contracted node's own row is scanned in step 6 from the values it was built from.
    {
      const int nscan = (levels == 0) ? cur : ndirty;
      for (int li = lane; li < nscan; li += 64) {
        const int k = (levels == 0) ? m.order[li] : m.dlist[li];
        if (k == 0) { m.tree[0] = 0; continue; }
        double bv;
        int tie;
        m.tree[k] = (IdT)scan_row(m, S + (size_t)k * ld, cur, has_nan, bv, tie);
        m.bval[k] = bv;
        m.tie[k] = (IdT)tie;
      }
    }
    MST_SYNC();
    // ---- 2. cycles by pointer doubling: p = f^(2^r), mn = smallest logical position on the walk
    for (int li = lane; li < cur; li += 64) {
      const int k = m.order[li];
      m.p0[k] = m.tree[k];
      m.mn0[k] = (IdT)li;
      m.flag[k] = 0;
    }
    MST_SYNC();
    IdT *pa = m.p0, *pb = m.p1, *ma = m.mn0, *mb = m.mn1;
    for (int span = 1; span < cur; span <<= 1) {
      for (int li = lane; li < cur; li += 64) {
        const int k = m.order[li];
        const int q = pa[k];
        pb[k] = pa[q];
        mb[k] = min(ma[k], ma[q]);
      }
      MST_SYNC();
      IdT* t = pa; pa = pb; pb = t;
      t = ma; ma = mb; mb = t;
    }
    for (int li = lane; li < cur; li += 64) m.flag[pa[m.order[li]]] = 1;  // eventual image = nodes on loops
    MST_SYNC();
    int chosen = -1;
    for (int li = lane; li < cur; li += 64) {
      const int k = m.order[li];
      if (k != 0 && m.flag[k]) chosen = max(chosen, (int)ma[k]);
    }
    chosen = wave_max_i(chosen);
    if (chosen < 0) break;  // a tree
    // ---- 3. ordered lists: cycle nodes C (logical order), survivors N (logical order)
    int c = 0, nn = 0;
    for (int base = 0; base < cur; base += 64) {
      const int li = base + lane;
      int k = 0;
      bool in_c = false, in_n = false;
      if (li < cur) {
        k = m.order[li];
        in_c = (k != 0) && m.flag[k] && (ma[k] == chosen);
        in_n = !in_c;
      }
      const unsigned long long bc = __ballot(in_c), bn = __ballot(in_n);
      const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      if (in_c) { m.clist[c + __popcll(bc & below)] = (IdT)k; m.incyc[k] = 1; }
      if (in_n) { m.p1[nn + __popcll(bn & below)] = (IdT)k; m.incyc[k] = 0; }  // p1 reused: new order list
      c += __popcll(bc);
      nn += __popcll(bn);
    }
    IdT* norder = m.p1;
    MST_SYNC();
    const int slot_m = m.clist[0];
    // rank of each survivor among surviving slots in ascending slot id (record addressing)
    {
      int run = 0;
      for (int base = 0; base < n; base += 64) {
        const int s = base + lane;
        const bool ok = (s < n) && m.alive[s] && !m.incyc[s];
        const unsigned long long bm = __ballot(ok);
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        if (ok) m.rankn[s] = (IdT)(run + __popcll(bm & below));
        run += __popcll(bm);
      }
    }
    // record header + cycle
    const int need = 2 + 2 * c + 2 * nn;
    if (rec_end + need > m.rec_cap) return false;
    // Exit condition every wave reaches: a contraction must shrink the graph.  With probabilities (>= 0,
    // zero diagonal, node 0 first in the order) a node never selects itself, so c >= 2; scores with
    // negative entries (not probabilities: a caller's bug, an overwritten buffer) can produce the
    // self-loop c == 1, which would contract nothing and spin forever -- report "no tree" instead.
    if (c < 2 || levels >= n) return false;
    IdT* rec = m.rec + rec_end;
    if (lane == 0) { rec[0] = (IdT)cur; rec[1] = (IdT)c; }
    for (int i = lane; i < c; i += 64) {
      const int k = m.clist[i];
      rec[2 + i] = (IdT)k;
      rec[2 + c + i] = m.tree[k];
      m.cs[i] = S[(size_t)k * ld + m.tree[k]];
    }
    MST_SYNC();
    // ---- 4. product of the cycle's arc scores, in cycle order
    double total = 1.0;
    for (int i = 0; i < c; ++i) total = total * m.cs[i];
    // ---- 5. best entry/exit of the contracted node per surviving node
    IdT* rec_h = rec + 2 + 2 * c;
    IdT* rec_d = rec_h + nn;
    bool new_nan = false;
    for (int a = lane; a < nn; a += 64) {
      const int j = norder[a];
      double hb = S[(size_t)m.clist[0] * ld + j] / m.cs[0] * total;
      int hi = 0;
      double db = (j == 0) ? 0.0 : S[(size_t)j * ld + m.clist[0]];
      int di = 0;
      for (int i = 1; i < c; ++i) {
        const int k = m.clist[i];
        const double hv = S[(size_t)k * ld + j] / m.cs[i] * total;
        if (hv > hb || (hv != hv && hb == hb)) { hb = hv; hi = i; }
        const double dv = (j == 0) ? 0.0 : S[(size_t)j * ld + k];
        if (dv > db || (dv != dv && db == db)) { db = dv; di = i; }
      }
      m.hval[a] = hb;
      m.dval[a] = db;
      rec_h[m.rankn[j]] = (IdT)hi;
      rec_d[m.rankn[j]] = (IdT)di;
      new_nan |= (hb != hb);
    }
    has_nan = has_nan || (__ballot(new_nan) != 0ull);  // 0/0 or inf*0 in the entry scores of this contraction
    MST_SYNC();
    // ---- 6. contracted node takes the slot of the cycle's first node and goes last in the order
    for (int a = lane; a < nn; a += 64) {
      const int j = norder[a];
      S[(size_t)slot_m * ld + j] = m.hval[a];
      if (j != 0) S[(size_t)j * ld + slot_m] = m.dval[a];
      m.order[a] = (IdT)j;
      m.pos[j] = (IdT)a;
    }
    // rows whose best head sat in the cycle (step 1): without a tie they now point at the contracted node, with one
    // they are scanned again; for every other row the new entry may have created a tie with its maximum.  With a NaN
    // in the matrix (now or before this level: the flags then predate it) every such row is scanned, exactly.
    ndirty = 0;
    for (int base = 0; base < nn; base += 64) {
      const int a = base + lane;
      int j = 0;
      bool dty = false, rescan = false;
      if (a < nn) {
        j = norder[a];
        dty = (j != 0) && m.incyc[m.tree[j]];
        if (dty) {
          rescan = has_nan || m.tie[j];
          if (!rescan) m.tree[j] = (IdT)slot_m;
        } else if (j != 0 && !(m.dval[a] < m.bval[j])) {
          m.tie[j] = 1;
        }
      }
      const unsigned long long bd = __ballot(rescan);
      const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      if (rescan) m.dlist[ndirty + __popcll(bd & below)] = (IdT)j;
      ndirty += __popcll(bd);
    }
    // the contracted node's row = hval[0 .. nn) in the new logical order, then its zero diagonal: all lanes scan it
    // together (first maximum: larger value, then smaller position), no order list and no trip through S
    if (has_nan) {
      if (lane == 0) m.dlist[ndirty] = (IdT)slot_m;
      ++ndirty;
    } else {
      double best = -INFINITY;
      int bl = 1 << 20;
      for (int l = lane; l <= nn; l += 64) {
        const double v = l < nn ? m.hval[l] : 0.0;
        if (v > best) { best = v; bl = l; }
      }
      if (bl == (1 << 20) && lane <= nn) bl = lane;  // all of this lane's entries are -inf: the first of them stands for them
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(best, o);
        const int ol = __shfl_xor(bl, o);
        if (ov > best || (ov == best && ol < bl)) { best = ov; bl = ol; }
      }
      int cnt = 0;
      for (int base = 0; base <= nn; base += 64) {
        const int l = base + lane;
        const double v = l < nn ? m.hval[l] : 0.0;
        cnt += __popcll(__ballot(l <= nn && v == best));
      }
      if (lane == 0) {
        m.tree[slot_m] = bl < nn ? norder[bl] : (IdT)slot_m;
        m.bval[slot_m] = best;
        m.tie[slot_m] = cnt >= 2 ? 1 : 0;
      }
    }
    for (int i = 1 + lane; i < c; i += 64) m.alive[m.clist[i]] = 0;
    if (lane == 0) {
      S[(size_t)slot_m * ld + slot_m] = 0.0;
      m.order[nn] = (IdT)slot_m;
      m.pos[slot_m] = (IdT)nn;
    }
    MST_SYNC();
    cur = nn + 1;
    rec_end += need;
    ++levels;
  }
  // ---- expansion, deepest level first (utils.py:491-518)
  MST_SYNC();
  int off_stack_end = rec_end;
  // records are variable length: walk forward once to find offsets is O(levels^2) if repeated,
  // so walk backwards using the stored sizes: each record's length is 2 + 2*m - ... recomputed
  // from its header; keep a small offset table in the (now free) hval area.
  int* offs = reinterpret_cast<int*>(m.hval);  // n entries of 4 bytes fit in np*8 bytes
  {
    int o = 0, l = 0;
    while (o < rec_end) {  // uniform loop, every lane walks the headers
      if (lane == 0) offs[l] = o;
      const int mm = m.rec[o], cc = m.rec[o + 1];
      o += 2 + 2 * cc + 2 * (mm - cc);
      ++l;
    }
  }
  MST_SYNC();
  (void)off_stack_end;
  for (int l = levels - 1; l >= 0; --l) {
    const IdT* rec = m.rec + offs[l];
    const int mm = rec[0], c = rec[1];
    const int nn = mm - c;
    const IdT* cl = rec + 2;
    const IdT* tc = rec + 2 + c;
    const IdT* rec_h = rec + 2 + 2 * c;
    const IdT* rec_d = rec_h + nn;
    const int slot_m = cl[0];
    const int head_m = m.tree[slot_m];
    MST_SYNC();
    int run = 0, rank_head = 0;
    for (int base = 0; base < n; base += 64) {
      const int s = base + lane;
      const bool ok = (s < n) && m.alive[s] && (s != slot_m);
      const unsigned long long bm = __ballot(ok);
      const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      const int rk = run + __popcll(bm & below);
      if (ok && m.tree[s] == slot_m) m.tree[s] = cl[rec_d[rk]];
      if (head_m >= base && head_m < base + 64) rank_head = run + __popcll(bm & ((head_m - base) == 0 ? 0ull : (~0ull >> (64 - (head_m - base)))));
      run += __popcll(bm);
    }
    MST_SYNC();
    for (int i = lane; i < c; i += 64) { m.tree[cl[i]] = tc[i]; m.alive[cl[i]] = 1; }
    MST_SYNC();
    if (lane == 0) m.tree[cl[rec_h[rank_head]]] = (IdT)head_m;
    MST_SYNC();
  }
  return true;
}

// S[k][j] for k >= 1 from the probability tensor (utils.py:569-574): crop to n columns, zero diagonal
// Returns whether the matrix holds a NaN (wave-uniform).
template <typename IdT>
__device__ inline bool build_scores(const MstLdsT<IdT>& m, const float* p, int n, int T, int lane, int root) {
  bool nan = false;
  // flattened over (k, j) with 8 independent loads in flight per lane (a row-by-row loop pays one
  // global-load latency per row: ~130 us of a 129-node task)
  const int total = (n - 1) * n;
  for (int e0 = lane; e0 < total; e0 += 64 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 64 * u;
      v[u] = 0.f;
      if (e < total) {
        const int k = e / n + 1, j = e - (k - 1) * n;
        v[u] = p[(size_t)(k - 1) * T + j];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 64 * u;
      if (e < total) {
        const int k = e / n + 1, j = e - (k - 1) * n;
        double d = (j == k) ? 0.0 : (double)v[u];
        if (root > 0) {  // set_root (utils.py:537-543)
          if (j == 0) d = (k == root) ? 1.0 : 0.0;
          else if (k == root) d = 0.0;
        }
        m.S[(size_t)k * m.ld + j] = d;
        nan |= (d != d);
      }
    }
  }
  MST_SYNC();
  return __ballot(nan) != 0ull;
}

struct MstWs {
  int32_t* ntasks;      // [4] (16-byte block at the start of the workspace)
  int32_t* sent_base;   // [B]
  int32_t* sent_cnt;    // [B]
  int32_t* task_sent;   // [B*L]
  int32_t* task_root;   // [B*L]
  double* task_score;   // [B*L]
  double* task_ub;      // [B*L] upper bound of the task's tree score (see mst_first_kernel)
  int32_t* sent_top;    // [B] the task with the largest bound of the sentence: runs first (-1: all of its tasks do)
  int32_t* top_list;    // [B*L] the tasks of the first round, compacted (count in ntasks[1])
  void* task_tree;      // [B*L][L+1] node ids (1 or 2 bytes each)
  double* s_first;      // long-sentence mode only: [B][n_max * ld] score matrices of the first pass
  double* s_root;       // and [mst_root_blocks][n_max * ld] of the root pass
  unsigned char* rec_first;  // 16-bit ids only: [B][rec_cap] contraction records of the first pass
  unsigned char* rec_root;   // workspace-resident root pass: [mst_root_blocks][rec_cap] records (1- or 2-byte ids)
};

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
__host__ inline size_t mst_s_elems(int L) { return (size_t)(L + 1) * ((L + 1) | 1); }
__host__ inline int mst_id_bytes(int L) { return L + 1 > MST_MAX_N_GLOBAL ? 2 : 1; }
__host__ inline size_t mst_root_blocks(int B, int L) {  // workgroups of the root kernel in workspace mode
  const size_t nt = (size_t)B * L;
  size_t cap = (size_t)MST_GLOBAL_ROOT_BLOCKS;
  if (mst_id_bytes(L) == 2) {  // a private matrix is up to 34 MB there: as many as the budget holds
    const size_t fit = MST_LONG_ROOT_BYTES / (8 * mst_s_elems(L) + 2 * mst_rec_cap(L + 1));
    cap = fit < 8 ? 8 : (fit < cap ? fit : cap);
  }
  return nt < cap ? nt : cap;
}
__host__ __device__ inline size_t mst_rec_stride(int T, int id) { return align16((size_t)id * mst_rec_cap(T)); }  // bytes per task
__host__ inline size_t mst_ws_bytes(int B, int L) {
  const size_t nt = (size_t)B * L;
  const int id = mst_id_bytes(L);
  size_t b = 16 + align16(4 * (size_t)B) * 3 + align16(4 * nt) * 3 + align16(8 * nt) * 2 + align16(nt * (L + 1) * id);
  if (L + 1 > MST_MAX_N) b += 8 * mst_s_elems(L) * (size_t)B;
  if (L + 1 >= MST_ROOT_GLOBAL_N) b += 8 * mst_s_elems(L) * mst_root_blocks(B, L);
  if (id == 2) b += mst_rec_stride(L + 1, 2) * (size_t)B;
  if (L + 1 >= MST_ROOT_GLOBAL_N) b += mst_rec_stride(L + 1, id) * mst_root_blocks(B, L);
  return b;
}
__host__ inline MstWs mst_ws_carve(void* ws, int B, int L) {
  MstWs w;
  unsigned char* q = (unsigned char*)ws;
  const size_t nt = (size_t)B * L;
  w.ntasks = (int32_t*)q; q += 16;
  w.sent_base = (int32_t*)q; q += align16(4 * (size_t)B);
  w.sent_cnt = (int32_t*)q; q += align16(4 * (size_t)B);
  w.task_sent = (int32_t*)q; q += align16(4 * nt);
  w.task_root = (int32_t*)q; q += align16(4 * nt);
  w.task_score = (double*)q; q += align16(8 * nt);
  w.task_ub = (double*)q; q += align16(8 * nt);
  w.sent_top = (int32_t*)q; q += align16(4 * (size_t)B);
  w.top_list = (int32_t*)q; q += align16(4 * nt);
  w.task_tree = (void*)q; q += align16(nt * (L + 1) * mst_id_bytes(L));
  w.s_first = (double*)q;
  w.s_root = w.s_first + (L + 1 > MST_MAX_N ? mst_s_elems(L) * B : 0);
  w.rec_first = (unsigned char*)(w.s_root + (L + 1 >= MST_ROOT_GLOBAL_N ? mst_s_elems(L) * mst_root_blocks(B, L) : 0));
  w.rec_root = w.rec_first + (mst_id_bytes(L) == 2 ? mst_rec_stride(L + 1, 2) * (size_t)B : 0);
  return w;
}

// the per-task slice of the contraction records in the workspace (GREC; otherwise they are in LDS)
template <typename IdT, bool GREC>
__device__ __forceinline__ IdT* mst_rec_slice(unsigned char* base, size_t task, int T) {
  if constexpr (GREC) return reinterpret_cast<IdT*>(base + task * mst_rec_stride(T, (int)sizeof(IdT)));
  else return nullptr;
}

// GS: score matrix in the workspace instead of LDS (sentences longer than MST_MAX_N - 1 words); GREC: the records too
template <bool GS, typename IdT, bool GREC>
__global__ __launch_bounds__(GS ? 64 : 64 * MST_MAX_WAVES) void mst_first_kernel(const float* __restrict__ probs,
                                                                       const int32_t* __restrict__ lens, int B, int L,
                                                                       int32_t* heads, int32_t* status, MstWs ws,
                                                                       int lds_task) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  unsigned char* const smem = smem_all + (size_t)wave * lds_task;
  const int b = blockIdx.x * wpb + wave;
  const int lane = threadIdx.x & 63;
  if (b >= B) return;
  const int T = L + 1;
  int len = lens[b];
  len = min(max(len, 0), L);
  const int n = len + 1;
  int32_t* hb = heads + (size_t)b * L;
  for (int i = lane; i < L; i += 64) hb[i] = 0;
  if (lane == 0) { ws.sent_cnt[b] = 0; ws.sent_base[b] = 0; status[b] = 0; }
  if (len == 0) return;
  const float* p = probs + (size_t)b * L * T;
  MstLdsT<IdT> m = mst_carve<IdT>(smem, n, GS ? ws.s_first + (size_t)b * T * (T | 1) : nullptr, mst_rec_slice<IdT, GREC>(ws.rec_first, b, T));
  const bool has_nan = build_scores(m, p, n, T, lane, 0);
  if (!cle_run(m, n, lane, has_nan)) { if (lane == 0) status[b] = 2; return; }
  MST_SYNC();
  // roots (utils.py:525-534)
  int nroots = 0;
  for (int base = 1; base < n; base += 64) {
    const int k = base + lane;
    nroots += __popcll(__ballot(k < n && m.tree[k] == 0));
  }
  if (nroots == 1) {
    for (int k = 1 + lane; k < n; k += 64) hb[k - 1] = m.tree[k];
    return;
  }
  // candidate flags in m.flag[k]
  int ncand = 0;
  if (nroots >= 2) {
    for (int k = lane; k < n; k += 64) m.flag[k] = (k >= 1 && m.tree[k] == 0) ? 1 : 0;
    ncand = nroots;
  } else {
    const double thr = 1.0 / (double)n;
    for (int base = 0; base < n; base += 64) {
      const int k = base + lane;
      const bool ok = (k >= 1 && k < n) && ((double)p[(size_t)(k - 1) * T] >= thr);
      if (k < n) m.flag[k] = ok ? 1 : 0;
      ncand += __popcll(__ballot(ok));
    }
    if (ncand == 0) {  // 1 + argmax(scores[1:, 0]), first maximum
      MST_SYNC();
      if (lane == 0) {
        float best = p[0];
        int bi = 1;
        for (int k = 2; k < n; ++k) {
          const float v = p[(size_t)(k - 1) * T];
          if (v > best || (v != v && best == best)) { best = v; bi = k; }
        }
        m.flag[bi] = 1;
      }
      ncand = 1;
    }
  }
  MST_SYNC();
  int base_task = 0;
  if (lane == 0) {
    base_task = atomicAdd(ws.ntasks, ncand);
    ws.sent_base[b] = base_task;
    ws.sent_cnt[b] = ncand;
    status[b] = -1;
  }
  base_task = __shfl(base_task, 0);
  // Upper bounds of the candidates' tree scores (no counterpart in the reference, which runs every candidate): in the
  // tree rooted in r every other word k takes a head j >= 1, j != k, so its log-score is at most
  //   ub_r = sum_{k != r} log max_j p[k][j]  +  log p[r][0].
  // The root kernel runs the candidate with the largest bound first and skips every candidate whose bound is below that
  // tree's exact score: it cannot be the reference's (strict, first) maximum.  Rows without a positive finite maximum or
  // a NaN anywhere switch the bounds off (+inf: every candidate runs).
  // Sentences with few candidates run them all in the first round: a second round costs a task's latency.
  const bool two_rounds = ncand > MST_BOUND_MIN_CAND;
  bool bound_ok = !has_nan && two_rounds;
  double tot = 0.0;
  if (bound_ok) {
    for (int k = 1 + lane; k < n; k += 64) {  // a row per lane, eight loads in flight
      const float* row = p + (size_t)(k - 1) * T;
      float best = -1.f;
      for (int j0 = 1; j0 < n; j0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (j0 + u < n && j0 + u != k) ? row[j0 + u] : -1.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) best = fmaxf(best, v[u]);
      }
      const double mk = (double)best;
      if (!(mk > 0.0) || mk > 1.0e300) bound_ok = false;
      const double lg = log(mk > 0.0 ? mk : 1.0);
      m.hval[k] = lg;
      tot += lg;
    }
    bound_ok = (__ballot(!bound_ok) == 0ull);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
    MST_SYNC();
  }
  int run = 0;
  double top_ub = -INFINITY;
  int top_t = 0x7fffffff;
  for (int base = 0; base < n; base += 64) {
    const int k = base + lane;
    const bool ok = (k < n) && m.flag[k];
    const unsigned long long bm = __ballot(ok);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (ok) {
      const int t = base_task + run + __popcll(bm & below);
      ws.task_sent[t] = b;
      ws.task_root[t] = k;
      double ub = INFINITY;
      if (bound_ok) {
        ub = (tot - m.hval[k]) + log((double)p[(size_t)(k - 1) * T]);
        if (ub != ub) ub = INFINITY;
      }
      ws.task_ub[t] = ub;
      if (top_t == 0x7fffffff || ub > top_ub) { top_ub = ub; top_t = t; }  // (t grows with the loop: the first maximum)
    }
    run += __popcll(bm);
  }
  if (!two_rounds) {
    int slot0 = 0;
    if (lane == 0) {
      ws.sent_top[b] = -1;
      slot0 = atomicAdd(ws.ntasks + 1, ncand);
    }
    slot0 = __shfl(slot0, 0);
    for (int i = lane; i < ncand; i += 64) ws.top_list[slot0 + i] = base_task + i;
    return;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {  // the largest bound, the smallest task among equals
    const double ou = __shfl_xor(top_ub, o);
    const int ot = __shfl_xor(top_t, o);
    if (ot != 0x7fffffff && (top_t == 0x7fffffff || ou > top_ub || (ou == top_ub && ot < top_t))) { top_ub = ou; top_t = ot; }
  }
  if (lane == 0) {
    ws.sent_top[b] = top_t;
    ws.top_list[atomicAdd(ws.ntasks + 1, 1)] = top_t;
  }
}

template <bool GS, typename IdT, bool GREC>
__global__ __launch_bounds__(GS ? 64 : 64 * MST_ROOT_MAX_WAVES) void mst_root_kernel(const float* __restrict__ probs,
                                                                      const int32_t* __restrict__ lens, int B, int L, MstWs ws,
                                                                      int lds_task, int phase) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  unsigned char* const smem = smem_all + (size_t)wave * lds_task;
  const int lane = threadIdx.x & 63;
  const int T = L + 1;
  // phase 0: the candidate with the largest bound of every sentence; phase 1 (a launch of its own, after it): the others,
  // except those whose bound is below the exact score phase 0 found (margin: the rounding of the two log-sums, ~1e-13)
  const int ntasks = min(ws.ntasks[phase == 0 ? 1 : 0], B * L);
  for (int ti = blockIdx.x * wpb + wave; ti < ntasks; ti += gridDim.x * wpb) {
    MST_SYNC();
    const int t = phase == 0 ? ws.top_list[ti] : ti;
    const int b = ws.task_sent[t];
    if (phase != 0) {
      const int top = ws.sent_top[b];
      if (top < 0 || t == top) continue;
      const double lb = ws.task_score[top];
      if (ws.task_ub[t] < lb - 1e-9 * (1.0 + fabs(lb))) {
        if (lane == 0) ws.task_score[t] = -INFINITY;
        continue;
      }
    }
    const int root = ws.task_root[t];
    const int n = min(max(lens[b], 0), L) + 1;
    const float* p = probs + (size_t)b * L * T;
    MstLdsT<IdT> m = mst_carve<IdT>(smem, n, GS ? ws.s_root + (size_t)(blockIdx.x * wpb + wave) * T * (T | 1) : nullptr,
                                    mst_rec_slice<IdT, GREC>(ws.rec_root, blockIdx.x * wpb + wave, T));
    const bool has_nan = build_scores(m, p, n, T, lane, root);
    IdT* tt = reinterpret_cast<IdT*>(ws.task_tree) + (size_t)t * T;
    if (!cle_run(m, n, lane, has_nan)) {
      if (lane == 0) ws.task_score[t] = -INFINITY;
      continue;
    }
    MST_SYNC();
    // tree score (utils.py:551-552): sum of log of the selected entries of the set_root matrix
    double s = 0.0;
    bool all_nz = true;
    for (int k = 1 + lane; k < n; k += 64) {
      const int h = m.tree[k];
      tt[k] = (IdT)h;
      double v;
      if (k == root) v = (h == 0) ? 1.0 : 0.0;
      else if (h == 0 || h == k) v = 0.0;
      else v = (double)p[(size_t)(k - 1) * T + h];
      if (v == 0.0) all_nz = false; else s += log(v);
    }
    all_nz = (__ballot(!all_nz) == 0ull);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const double root_score = (double)p[(size_t)(root - 1) * T];
    double sc = -INFINITY;
    if (all_nz) sc = s + log(root_score);
    if (lane == 0) ws.task_score[t] = sc;
  }
}

template <typename IdT>
__global__ __launch_bounds__(64) void mst_select_kernel(const int32_t* __restrict__ lens, int B, int L, int32_t* heads,
                                                        int32_t* status, MstWs ws) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const int cnt = ws.sent_cnt[b];
  if (cnt <= 0) return;
  const int base = ws.sent_base[b];
  double best = -INFINITY;
  int bt = -1;
  for (int i = 0; i < cnt; ++i) {
    const double sc = ws.task_score[base + i];
    if (sc > best) { best = sc; bt = base + i; }
  }
  if (bt < 0) {
    if (lane == 0) status[b] = 1;
    return;
  }
  const int n = min(max(lens[b], 0), L) + 1;
  const IdT* tt = reinterpret_cast<const IdT*>(ws.task_tree) + (size_t)bt * (L + 1);
  for (int k = 1 + lane; k < n; k += 64) heads[(size_t)b * L + k - 1] = tt[k];
  if (lane == 0) status[b] = 0;
}

}  // namespace

extern "C" size_t cube_mst_workspace_bytes(int B, int L) {
  if (B <= 0 || L <= 0) return 16;
  return mst_ws_bytes(B, L);
}

namespace {

// Launch sequence shared by cube_mst_decode and cube_mst_decode_noalloc; `workspace` holds mst_ws_bytes(B, L) bytes.
// Variants: S in LDS (up to 129 words; the rooted re-runs up to 94) / in the workspace, byte ids; 16-bit ids from 255 words.
template <bool GS_FIRST, bool GS_ROOT, typename IdT>
int mst_launch(const float* probs, const int32_t* lens, int B, int L, int32_t* heads, int32_t* status, const MstWs& ws,
               hipStream_t st) {
  constexpr int ID = (int)sizeof(IdT);
  constexpr bool GREC_FIRST = ID == 2, GREC_ROOT = GS_ROOT;  // records in the workspace
  const size_t task_first = (mst_lds_bytes(L + 1, GS_FIRST, ID, GREC_FIRST) + 15) & ~(size_t)15;
  const size_t task_root = (mst_lds_bytes(L + 1, GS_ROOT, ID, GREC_ROOT) + 15) & ~(size_t)15;
  // tasks per workgroup: as many as the CU's LDS holds (workspace-resident matrices: one, their blocks index the workspace)
  auto waves = [](size_t task, bool gs, int most) {
    if (gs) return 1;
    const size_t w = ((size_t)160 * 1024) / task;
    return (int)(w < 1 ? 1 : (w > (size_t)most ? (size_t)most : w));
  };
  const int w_first = waves(task_first, GS_FIRST, MST_MAX_WAVES), w_root = waves(task_root, GS_ROOT, MST_ROOT_MAX_WAVES);
  const size_t lds_first = task_first * w_first, lds_root = task_root * w_root;
  static CubeLdsReservation reserved[2];
  if (!reserved[0].ensure((const void*)mst_first_kernel<GS_FIRST, IdT, GREC_FIRST>, lds_first) ||
      !reserved[1].ensure((const void*)mst_root_kernel<GS_ROOT, IdT, GREC_ROOT>, lds_root))
    CUBE_FAIL(CUBE_ERR_LAUNCH, "cube_mst_decode: cannot reserve %zu bytes of LDS", lds_first > lds_root ? lds_first : lds_root);
  if (hipMemsetAsync(ws.ntasks, 0, 16, st) != hipSuccess) CUBE_FAIL(CUBE_ERR_LAUNCH, "cube_mst_decode: memset failed");
  hipLaunchKernelGGL((mst_first_kernel<GS_FIRST, IdT, GREC_FIRST>), dim3((B + w_first - 1) / w_first), dim3(64 * w_first), lds_first, st, probs,
                     lens, B, L, heads, status, ws, (int)task_first);
  CUBE_CHECK_LAUNCH("cube_mst_decode(first)");
  // one task per candidate root, at most B*L of them; the grid strides over the device-side count
  int nblk;
  if (GS_ROOT) {
    nblk = (int)mst_root_blocks(B, L);  // = the private score matrices the workspace holds
  } else {
    nblk = (B * L + w_root - 1) / w_root;
    const int cap = w_root > 1 ? MST_ROOT_MAX_BLOCKS : 4096;
    if (nblk > cap) nblk = cap;
  }
  int nblk0 = (MST_BOUND_MIN_CAND * B + w_root - 1) / w_root;  // one task per sentence, or its few candidates
  if (nblk0 > nblk) nblk0 = nblk;
  hipLaunchKernelGGL((mst_root_kernel<GS_ROOT, IdT, GREC_ROOT>), dim3(nblk0), dim3(64 * w_root), lds_root, st, probs, lens, B, L, ws,
                     (int)task_root, 0);
  CUBE_CHECK_LAUNCH("cube_mst_decode(root, first candidates)");
  hipLaunchKernelGGL((mst_root_kernel<GS_ROOT, IdT, GREC_ROOT>), dim3(nblk), dim3(64 * w_root), lds_root, st, probs, lens, B, L, ws,
                     (int)task_root, 1);
  CUBE_CHECK_LAUNCH("cube_mst_decode(root)");
  hipLaunchKernelGGL(mst_select_kernel<IdT>, dim3(B), dim3(64), 0, st, lens, B, L, heads, status, ws);
  CUBE_CHECK_LAUNCH("cube_mst_decode(select)");
  return CUBE_OK;
}

int mst_decode_impl(const float* probs, const int32_t* lens, int B, int L, int32_t* heads, int32_t* status,
                    void* workspace, hipStream_t st) {
  const MstWs ws = mst_ws_carve(workspace, B, L);
  if (L + 1 > MST_MAX_N_GLOBAL) return mst_launch<true, true, uint16_t>(probs, lens, B, L, heads, status, ws, st);
  if (L + 1 > MST_MAX_N) return mst_launch<true, true, uint8_t>(probs, lens, B, L, heads, status, ws, st);
  if (L + 1 >= MST_ROOT_GLOBAL_N) return mst_launch<false, true, uint8_t>(probs, lens, B, L, heads, status, ws, st);
  return mst_launch<false, false, uint8_t>(probs, lens, B, L, heads, status, ws, st);
}

}  // namespace

extern "C" int cube_mst_decode(const float* probs, const int32_t* lens, int B, int L, int32_t* heads, int32_t* status,
                               void* workspace, size_t workspace_bytes, void* stream) {
  CUBE_ENTRY();
  if (B <= 0 || L <= 0) return CUBE_OK;
  if (!probs || !lens || !heads || !status || !workspace) CUBE_FAIL(CUBE_ERR_ARG, "cube_mst_decode: null pointer");
  if (L + 1 > MST_MAX_N_LONG) CUBE_FAIL(CUBE_ERR_ARG, "cube_mst_decode: sentences longer than %d words are not supported", MST_MAX_N_LONG - 1);
  if (workspace_bytes < mst_ws_bytes(B, L)) CUBE_FAIL(CUBE_ERR_ARG, "cube_mst_decode: workspace too small (%zu < %zu)", workspace_bytes, mst_ws_bytes(B, L));
  if ((uintptr_t)workspace % 16) CUBE_FAIL(CUBE_ERR_ALIGN, "cube_mst_decode: workspace must be 16-byte aligned");
  return mst_decode_impl(probs, lens, B, L, heads, status, workspace, (hipStream_t)stream);
}

// ---- the entry point that allocates nothing (include/cube_hip.h) -----------------------------------
// The decoder's workspace (task list, candidate trees and scores) lives in a file-scope device array
// instead of caller memory.  Every GPU the library runs on holds its own copy of the array at its own
// address until the process ends; in host code the array's NAME is only a stand-in for the symbol, so the
// address is asked for once per device (cube_init, or the first call on that device) and kept.
static __device__ __align__(256) unsigned char scratch[CUBE_MST_NOALLOC_BYTES];

namespace {
std::atomic<void*> scratch_ptr[CUBE_MAX_DEVICES] = {};

// device address of `scratch` on the CURRENT device (nullptr + error text on failure)
void* mst_scratch() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CUBE_MAX_DEVICES) {
    cube_set_error("cube_init: no current HIP device (or device index >= %d)", CUBE_MAX_DEVICES);
    return nullptr;
  }
  void* p = scratch_ptr[dev].load(std::memory_order_acquire);
  if (p) return p;
  const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
  if (e != hipSuccess || !p) {
    cube_set_error("cube_init: hipGetSymbolAddress(scratch) failed on device %d: %s", dev, hipGetErrorString(e));
    return nullptr;
  }
  scratch_ptr[dev].store(p, std::memory_order_release);  // racing threads store the same address
  return p;
}
}  // namespace

extern "C" int cube_init(void) {
  CUBE_ENTRY();
  return mst_scratch() ? CUBE_OK : CUBE_ERR_LAUNCH;
}

extern "C" int cube_mst_decode_noalloc(const float* probs, const int32_t* lens, int B, int L, int32_t* heads,
                                       int32_t* status, void* stream) {
  CUBE_ENTRY();
  if (B <= 0 || L <= 0) return CUBE_OK;
  if (!probs || !lens || !heads || !status) CUBE_FAIL(CUBE_ERR_ARG, "cube_mst_decode_noalloc: null pointer");
  if (L + 1 > MST_MAX_N_LONG)
    CUBE_FAIL(CUBE_ERR_ARG, "cube_mst_decode_noalloc: sentences longer than %d words are not supported", MST_MAX_N_LONG - 1);
  if (mst_ws_bytes(B, L) > (size_t)CUBE_MST_NOALLOC_BYTES)
    CUBE_FAIL(CUBE_ERR_ARG, "cube_mst_decode_noalloc: B=%d x L=%d needs %zu bytes of scratch, the built-in buffer holds %zu "
              "(use cube_mst_decode with a caller workspace)", B, L, mst_ws_bytes(B, L), (size_t)CUBE_MST_NOALLOC_BYTES);
  void* ws = mst_scratch();  // the per-device address, never the array's name
  if (!ws) return CUBE_ERR_LAUNCH;
  return mst_decode_impl(probs, lens, B, L, heads, status, ws, (hipStream_t)stream);
}

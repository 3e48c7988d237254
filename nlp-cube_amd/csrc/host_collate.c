/* Host-side helpers of the featurisation (cube_amd/networks/utils.py, the counterpart of the
 * reference's MorphoCollate.collate_fn, cube/networks/utils.py:232-330): the two bulk copies that
 * NumPy can only do item by item.  CPython extension module `_cube_host`, built with gcc by
 * __graft_entry__.build(); no GPU code here.
 *
 *   stack_rows(rows, out)          rows: sequence of N one-dimensional contiguous buffers of one size and
 *                                  element type (word.emb[k]; ValueError otherwise),
 *                                  out: writable contiguous buffer of N * size bytes  -> None
 *   fill_ragged(flat, lens, out, width)
 *                                  flat: int64 values of all items back to back, lens: int64 [N],
 *                                  out: int64 [N, width], zero-initialised by the caller: row i gets its
 *                                  lens[i] values at columns 0..lens[i]-1 (utils.py:291-304: x_word,
 *                                  x_word_case)  -> None
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>
#include <stdint.h>
#include <string.h>

static PyObject* stack_rows(PyObject* self, PyObject* args) {
  PyObject* rows;
  Py_buffer out;
  if (!PyArg_ParseTuple(args, "Ow*", &rows, &out)) return NULL;
  PyObject* seq = PySequence_Fast(rows, "stack_rows: rows must be a sequence");
  if (!seq) {
    PyBuffer_Release(&out);
    return NULL;
  }
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
  PyObject** items = PySequence_Fast_ITEMS(seq);
  int ok = 1;
  if (n > 0 && (out.len % n) != 0) {
    PyErr_SetString(PyExc_ValueError, "stack_rows: output size is not a multiple of the row count");
    ok = 0;
  }
  const Py_ssize_t row_bytes = n > 0 ? out.len / n : 0;
  char* dst = (char*)out.buf;
  char fmt0[16] = "";
  int type0 = -1;
  for (Py_ssize_t i = 0; ok && i < n; ++i) {
    PyObject* it = items[i];
    // the rows are separate heap objects: pull the header of a later row and the data of a nearer one
    if (i + 8 < n) __builtin_prefetch(items[i + 8]);
    if (i + 4 < n && type0 >= 0 && PyArray_Check(items[i + 4])) {
      const char* d = (const char*)PyArray_DATA((PyArrayObject*)items[i + 4]);
      for (Py_ssize_t o = 0; o < row_bytes; o += 256) __builtin_prefetch(d + o);
    }
    if (PyArray_Check(it)) {  // NumPy rows (the usual case): no buffer-protocol round trip per word
      PyArrayObject* a = (PyArrayObject*)it;
      if (i == 0) type0 = PyArray_TYPE(a);
      if (PyArray_NDIM(a) != 1 || !PyArray_IS_C_CONTIGUOUS(a) || PyArray_NBYTES(a) != row_bytes ||
          PyArray_TYPE(a) != type0) {
        PyErr_Format(PyExc_ValueError, "stack_rows: row %zd differs from row 0 in size, rank or element type", i);
        ok = 0;
      } else {
        memcpy(dst + i * row_bytes, PyArray_DATA(a), (size_t)row_bytes);
      }
      continue;
    }
    Py_buffer b;
    if (type0 >= 0 || PyObject_GetBuffer(it, &b, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) {
      if (type0 >= 0) PyErr_Format(PyExc_ValueError, "stack_rows: row %zd is not a NumPy array like row 0", i);
      ok = 0;
      break;
    }
    const char* fmt = b.format ? b.format : "B";
    if (i == 0) strncpy(fmt0, fmt, sizeof(fmt0) - 1);
    if (b.len != row_bytes || b.ndim != 1 || strncmp(fmt, fmt0, sizeof(fmt0) - 1) != 0) {
      PyErr_Format(PyExc_ValueError, "stack_rows: row %zd differs from row 0 in size, rank or element type", i);
      ok = 0;
    } else {
      memcpy(dst + i * row_bytes, b.buf, (size_t)row_bytes);
    }
    PyBuffer_Release(&b);
  }
  Py_DECREF(seq);
  PyBuffer_Release(&out);
  if (!ok) return NULL;
  Py_RETURN_NONE;
}

static PyObject* fill_ragged(PyObject* self, PyObject* args) {
  Py_buffer flat, lens, out;
  Py_ssize_t width;
  if (!PyArg_ParseTuple(args, "y*y*w*n", &flat, &lens, &out, &width)) return NULL;
  int ok = 1;
  const Py_ssize_t n = lens.len / 8;
  if (width <= 0 || out.len != n * width * 8 || flat.len % 8 || lens.len % 8) {
    PyErr_SetString(PyExc_ValueError, "fill_ragged: sizes do not match (int64 flat, int64 lens[N], int64 out[N, width])");
    ok = 0;
  }
  if (ok) {
    const int64_t* f = (const int64_t*)flat.buf;
    const int64_t* l = (const int64_t*)lens.buf;
    int64_t* o = (int64_t*)out.buf;
    const Py_ssize_t total = flat.len / 8;
    Py_ssize_t pos = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
      const int64_t k = l[i];
      if (k < 0 || k > width || pos + k > total) {
        PyErr_Format(PyExc_ValueError, "fill_ragged: item %zd has length %lld (width %zd, %zd values left)", i,
                     (long long)k, width, total - pos);
        ok = 0;
        break;
      }
      memcpy(o + i * width, f + pos, (size_t)k * 8);
      pos += k;
    }
    if (ok && pos != total) {
      PyErr_SetString(PyExc_ValueError, "fill_ragged: the lengths do not add up to the number of values");
      ok = 0;
    }
  }
  PyBuffer_Release(&flat);
  PyBuffer_Release(&lens);
  PyBuffer_Release(&out);
  if (!ok) return NULL;
  Py_RETURN_NONE;
}

/* ======================================================================================================
 * featurize / write_back: the whole host side of one batch in two calls.
 *
 * featurize = MorphoCollate.collate_fn (reference cube/networks/utils.py:232-330) followed by the host half
 * of Parser._prepare (cube_amd/networks/parser.py: word-length groups of the ragged character CNN, the packed
 * int32 index upload, the external embeddings scattered to their padded [B*L, Epad] rows) in one pass over
 * the Sentence / Word objects, without the reference-format int64 tensors in between.  The arrays are
 * bit-identical to those of the Python path (tests/test_host.py).  write_back = the loop of
 * Parser.process that stores head / label / upos / xpos / attrs into the words (parser.py:432-446).
 * Words whose class keeps these attributes in __slots__ (cube_amd.io_utils.objects.Word) are read and
 * written through the slot offsets; any other object (the reference's own Word) through getattr / setattr.
 * ====================================================================================================== */
#include <structmember.h>
#include <pthread.h>
#include "../../include/cube_hip.h" /* cube_conv_groups, CUBE_MAX_CONV_GROUPS */

enum { F_WORD = 0, F_EMB, F_HEAD, F_LABEL, F_UPOS, F_XPOS, F_ATTRS, F_LEMMA, F_COUNT };
static const char* const FIELD_NAMES[F_COUNT] = {"word", "emb", "head", "label", "upos", "xpos", "attrs", "lemma"};
static PyObject* field_str[F_COUNT];
static PyObject *str_words, *str_lang_id, *str_lower;

typedef struct {
  PyTypeObject* tp;
  Py_ssize_t off[F_COUNT]; /* slot offsets, -1 = use getattr / setattr */
} WordLayout;

static void word_layout(PyTypeObject* tp, WordLayout* wl) {
  wl->tp = tp;
  for (int f = 0; f < F_COUNT; ++f) {
    wl->off[f] = -1;
    PyObject* d = tp->tp_dict ? PyDict_GetItem(tp->tp_dict, field_str[f]) : NULL; /* borrowed; the class's own slots only */
    if (d && Py_TYPE(d) == &PyMemberDescr_Type) {
      PyMemberDef* m = ((PyMemberDescrObject*)d)->d_member;
      if (m && m->type == T_OBJECT_EX && !(m->flags & READONLY)) wl->off[f] = m->offset;
    }
  }
}

/* new reference, or NULL with an exception set */
static inline PyObject* word_get(PyObject* w, WordLayout* wl, int f) {
  if (Py_TYPE(w) != wl->tp) word_layout(Py_TYPE(w), wl);
  if (wl->off[f] >= 0) {
    PyObject* v = *(PyObject**)((char*)w + wl->off[f]);
    if (v) {
      Py_INCREF(v);
      return v;
    }
    PyErr_SetObject(PyExc_AttributeError, field_str[f]);
    return NULL;
  }
  return PyObject_GetAttr(w, field_str[f]);
}

/* pointer valid while the GIL is held and the word is not modified; *owned = 1 when it is a NEW reference
 * (the getattr path: the value may be computed on the fly), which the caller must keep until it is done */
static inline PyObject* word_peek(PyObject* w, WordLayout* wl, int f, int* owned) {
  if (Py_TYPE(w) != wl->tp) word_layout(Py_TYPE(w), wl);
  *owned = 0;
  if (wl->off[f] >= 0) {
    PyObject* v = *(PyObject**)((char*)w + wl->off[f]);
    if (!v) PyErr_SetObject(PyExc_AttributeError, field_str[f]);
    return v;
  }
  *owned = 1;
  return PyObject_GetAttr(w, field_str[f]);
}

/* steals nothing; 0 on success */
static inline int word_set(PyObject* w, WordLayout* wl, int f, PyObject* v) {
  if (Py_TYPE(w) != wl->tp) word_layout(Py_TYPE(w), wl);
  if (wl->off[f] >= 0) {
    PyObject** slot = (PyObject**)((char*)w + wl->off[f]);
    PyObject* old = *slot;
    Py_INCREF(v);
    *slot = v;
    Py_XDECREF(old);
    return 0;
  }
  return PyObject_SetAttr(w, field_str[f], v);
}

#include <time.h>
static double g_t[12]; /* 0..7: the stages of featurize_times; 8..11: sweeps A+B, C, D+E of the object pass */
static long g_calls;
static inline double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* Grow-only scratch buffers of featurize: keeping them avoids a dozen malloc / mmap / page-fault / munmap rounds of
 * 64..800 KB per batch.  The call holds the GIL, but Python code can still run inside it (a Word class with properties or
 * __getattr__, finalisers run by an allocation) and switch to a thread that calls featurize again: `g_sc_busy` turns
 * that second call away (TypeError -> the caller's Python featurisation) instead of letting it realloc these buffers. */
static int g_sc_busy;
enum { SC_SLOTS = 20 };
static void* g_sc[SC_SLOTS];
static size_t g_sc_cap[SC_SLOTS];
static void* sc_alloc(int slot, size_t bytes, int zero) {
  if (bytes == 0) bytes = 8;
  if (g_sc_cap[slot] < bytes) {
    free(g_sc[slot]);
    size_t cap = bytes + bytes / 2 + 4096;
    g_sc[slot] = malloc(cap);
    g_sc_cap[slot] = g_sc[slot] ? cap : 0;
    if (!g_sc[slot]) return NULL;
  }
  if (zero) memset(g_sc[slot], 0, bytes);
  return g_sc[slot];
}

/* ---- raw token -> word id, C side.  utils.py:284-289 looks every word up as word2int[word.lower()]; `featurize` remembers
 * the answer per RAW token.  As a Python dict that was 140 ns per word on text that does not fit the caches (hash the str,
 * probe the index, follow the entry, compare the key string: four dependent misses that nothing prefetches) -- 82 % of the
 * object pass.  Here: open addressing over 64-byte slots keyed by the FNV-1a hash of the code points that the string sweep
 * computes anyway; a slot holds the hash, the id and the code points themselves (up to 12 inline, longer words in an
 * arena), so a hit is ONE cache line, verified exactly, and the line is prefetched while the strings are being read. */
enum { WID_INLINE = 12 };
typedef struct { uint64_t h; int32_t id; int32_t len; int32_t cp[WID_INLINE]; } WidSlot; /* len > WID_INLINE: cp[0] = arena offset */
typedef struct {
  WidSlot* slots;
  size_t mask, used;
  int32_t* arena;
  size_t arena_len, arena_cap;
} WidTable;
enum { WID_MAX_ENTRIES = 1 << 19 }; /* then the table starts over (as the dict did at 2^20) */

static void wid_free(WidTable* t) {
  free(t->slots);
  free(t->arena);
  t->slots = NULL; t->arena = NULL; t->mask = t->used = t->arena_len = t->arena_cap = 0;
}
static int wid_init(WidTable* t, size_t n_slots) {
  if (posix_memalign((void**)&t->slots, 64, n_slots * sizeof(WidSlot)) != 0) { t->slots = NULL; return -1; }
  memset(t->slots, 0, n_slots * sizeof(WidSlot));
  t->mask = n_slots - 1;
  t->used = 0;
  return 0;
}
static inline const int32_t* wid_cps(const WidTable* t, const WidSlot* s) {
  return s->len > WID_INLINE ? t->arena + s->cp[0] : s->cp;
}
/* -> slot of the token, or the empty slot where it belongs */
static inline WidSlot* wid_find(const WidTable* t, uint64_t h, const int32_t* cp, int32_t len) {
  size_t i = (size_t)h & t->mask;
  for (;;) {
    WidSlot* s = &t->slots[i];
    if (s->h == 0) return s;
    if (s->h == h && s->len == len && memcmp(wid_cps(t, s), cp, sizeof(int32_t) * (size_t)len) == 0) return s;
    i = (i + 1) & t->mask;
  }
}
static int wid_insert(WidTable* t, uint64_t h, const int32_t* cp, int32_t len, int32_t id) {
  if (t->used >= WID_MAX_ENTRIES) { /* start over, small */
    wid_free(t);
    if (wid_init(t, 1 << 16) != 0) return -1;
  } else if (2 * (t->used + 1) > t->mask + 1) { /* grow: rehash into twice the slots */
    WidTable big = *t;
    if (wid_init(&big, 2 * (t->mask + 1)) != 0) return -1;
    for (size_t i = 0; i <= t->mask; ++i) {
      const WidSlot* s = &t->slots[i];
      if (s->h == 0) continue;
      size_t j = (size_t)s->h & big.mask;
      while (big.slots[j].h != 0) j = (j + 1) & big.mask;
      big.slots[j] = *s;
      ++big.used;
    }
    free(t->slots);
    *t = big;
  }
  WidSlot* s = wid_find(t, h, cp, len);
  if (s->h != 0) { s->id = id; return 0; }
  if (len > WID_INLINE) {
    if (t->arena_len + (size_t)len > t->arena_cap) {
      const size_t cap = 2 * (t->arena_len + (size_t)len) + 4096;
      int32_t* a = (int32_t*)realloc(t->arena, cap * sizeof(int32_t));
      if (!a) return -1;
      t->arena = a;
      t->arena_cap = cap;
    }
    if (t->arena_len + (size_t)len > 0x7fffffff) return -1;
    memcpy(t->arena + t->arena_len, cp, sizeof(int32_t) * (size_t)len);
    s->cp[0] = (int32_t)t->arena_len;
    t->arena_len += (size_t)len;
  } else {
    memcpy(s->cp, cp, sizeof(int32_t) * (size_t)len);
  }
  s->h = h; s->id = id; s->len = len;
  ++t->used;
  return 0;
}
static void wid_capsule_free(PyObject* cap) {
  WidTable* t = (WidTable*)PyCapsule_GetPointer(cap, "cube_wid_table");
  if (t) { wid_free(t); free(t); }
}
/* wid_table_new() -> capsule: an empty token -> word id table for one vocabulary (one MorphoCollate) */
static PyObject* wid_table_new(PyObject* self, PyObject* args) {
  WidTable* t = (WidTable*)calloc(1, sizeof(WidTable));
  if (!t || wid_init(t, 1 << 16) != 0) { free(t); return PyErr_NoMemory(); }
  return PyCapsule_New(t, "cube_wid_table", wid_capsule_free);
}
/* wid_table_size(capsule) -> (entries, slots) */
static PyObject* wid_table_size(PyObject* self, PyObject* cap) {
  WidTable* t = (WidTable*)PyCapsule_GetPointer(cap, "cube_wid_table");
  if (!t) return NULL;
  return Py_BuildValue("(nn)", (Py_ssize_t)t->used, (Py_ssize_t)(t->mask + 1));
}


/* ---- a small persistent thread pool for the memory-bound sweeps of featurize.  The calling thread holds the GIL for the
 * whole call and no Python code runs inside it, so the workers may READ Python objects (struct fields, list item arrays,
 * str and ndarray payloads) without it; they never touch reference counts or call the C API.  Threads are created once
 * (a pthread_create per call cost more than a sweep); a forked child starts without them. */
enum { POOL_MAX = 16 };
typedef void (*pool_fn)(void* ctx, int chunk, int n_chunks);
static struct {
  pthread_mutex_t mu;
  pthread_cond_t wake, done;
  pthread_t th[POOL_MAX];
  int n_threads;        /* workers alive */
  unsigned long gen;    /* job generation */
  pool_fn fn;
  void* ctx;
  int n_chunks, next, running;
  int ready;
} g_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, 0, 0, NULL, NULL, 0, 0, 0, 0};

static void* pool_worker(void* arg) {
  unsigned long seen = 0;
  pthread_mutex_lock(&g_pool.mu);
  for (;;) {
    while (g_pool.gen == seen) pthread_cond_wait(&g_pool.wake, &g_pool.mu);
    seen = g_pool.gen;
    for (;;) {
      const int c = g_pool.next;
      if (c >= g_pool.n_chunks) break;
      g_pool.next = c + 1;
      ++g_pool.running;
      pthread_mutex_unlock(&g_pool.mu);
      g_pool.fn(g_pool.ctx, c, g_pool.n_chunks);
      pthread_mutex_lock(&g_pool.mu);
      --g_pool.running;
    }
    if (g_pool.running == 0) pthread_cond_signal(&g_pool.done);
  }
  return NULL;
}
static void pool_child_reset(void) { /* after fork(): the child has no workers, and the mutex state is the parent's */
  pthread_mutex_init(&g_pool.mu, NULL);
  pthread_cond_init(&g_pool.wake, NULL);
  pthread_cond_init(&g_pool.done, NULL);
  g_pool.n_threads = 0;
  g_pool.gen = 0;
  g_pool.n_chunks = g_pool.next = g_pool.running = 0;
}
/* fn(ctx, chunk, n_chunks) for every chunk, on up to `want` threads including the caller's; returns when all are done */
static void pool_run(pool_fn fn, void* ctx, int n_chunks, int want) {
  if (want > POOL_MAX) want = POOL_MAX;
  if (want <= 1 || n_chunks <= 1) {
    for (int c = 0; c < n_chunks; ++c) fn(ctx, c, n_chunks);
    return;
  }
  pthread_mutex_lock(&g_pool.mu);
  if (!g_pool.ready) {
    pthread_atfork(NULL, NULL, pool_child_reset);
    g_pool.ready = 1;
  }
  while (g_pool.n_threads < want - 1) {
    if (pthread_create(&g_pool.th[g_pool.n_threads], NULL, pool_worker, NULL) != 0) break;
    pthread_detach(g_pool.th[g_pool.n_threads]);
    ++g_pool.n_threads;
  }
  g_pool.fn = fn; g_pool.ctx = ctx; g_pool.n_chunks = n_chunks; g_pool.next = 0; g_pool.running = 0;
  ++g_pool.gen;
  pthread_cond_broadcast(&g_pool.wake);
  for (;;) { /* the caller works too */
    const int c = g_pool.next;
    if (c >= g_pool.n_chunks) break;
    g_pool.next = c + 1;
    ++g_pool.running;
    pthread_mutex_unlock(&g_pool.mu);
    fn(ctx, c, n_chunks);
    pthread_mutex_lock(&g_pool.mu);
    --g_pool.running;
  }
  while (g_pool.running > 0) pthread_cond_wait(&g_pool.done, &g_pool.mu);
  g_pool.n_chunks = 0; /* late wakers find nothing */
  pthread_mutex_unlock(&g_pool.mu);
}

/* The object pass of featurize on the pool, for documents in the plain format: every Word of one class with `word` and
 * `emb` in __slots__, exact str, ready; emb a list / tuple of K contiguous float32 vectors.  Anything else sets `fallback`
 * and featurize runs its serial sweeps (which also produce the error messages). */
typedef struct {
  Py_ssize_t N, K, B, L, top;
  PyObject** wobj;
  PyObject** wstr;
  PyTypeObject* tp;
  Py_ssize_t off_word, off_emb;
  const Py_ssize_t* E;
  int32_t *wlen, *wid, *cp_off, *cps;
  const int32_t *sb, *si, *tab_char;
  const float** esrc;
  uint64_t* whash;
  const WidTable* wtab;
  Py_ssize_t chunk_maxlen[64];
  volatile int fallback, missing;
} ParCtx;

static void par_sweep_objects(void* vctx, int chunk, int n_chunks) {
  ParCtx* c = (ParCtx*)vctx;
  const Py_ssize_t w0 = c->N * chunk / n_chunks, w1 = c->N * (chunk + 1) / n_chunks;
  enum { PFD = 8 };
  Py_ssize_t maxlen = 0;
  /* Word -> str, emb list */
  for (Py_ssize_t w = w0; w < w1; ++w) {
    if (w + PFD < w1) { __builtin_prefetch(c->wobj[w + PFD]); __builtin_prefetch((char*)c->wobj[w + PFD] + 64); }
    PyObject* o = c->wobj[w];
    if (Py_TYPE(o) != c->tp) { c->fallback = 1; return; }
    PyObject* t = *(PyObject**)((char*)o + c->off_word);
    if (!t) { c->fallback = 1; return; }
    c->wstr[w] = t;
    __builtin_prefetch(t);
    if (c->K > 0) {
      PyObject* el = *(PyObject**)((char*)o + c->off_emb);
      if (!el) { c->fallback = 1; return; }
      __builtin_prefetch(el);
    }
  }
  /* str -> length;  emb list -> arrays -> data pointers */
  for (Py_ssize_t w = w0; w < w1; ++w) {
    PyObject* t = c->wstr[w];
    if (!PyUnicode_CheckExact(t) || !PyUnicode_IS_READY(t)) { c->fallback = 1; return; }
    const Py_ssize_t len = PyUnicode_GET_LENGTH(t);
    if (len > 0x3fffffff) { c->fallback = 1; return; }
    c->wlen[w] = (int32_t)len;
    if (len > maxlen) maxlen = len;
    if (c->K > 0) {
      PyObject* el = *(PyObject**)((char*)c->wobj[w] + c->off_emb);
      if (!(PyList_CheckExact(el) || PyTuple_CheckExact(el)) || PySequence_Fast_GET_SIZE(el) != c->K) { c->fallback = 1; return; }
      PyObject** items = PySequence_Fast_ITEMS(el);
      __builtin_prefetch(items);
    }
  }
  if (c->K > 0) {
    for (Py_ssize_t w = w0; w < w1; ++w) {
      PyObject** items = PySequence_Fast_ITEMS(*(PyObject**)((char*)c->wobj[w] + c->off_emb));
      for (Py_ssize_t k = 0; k < c->K; ++k) __builtin_prefetch((char*)items[k] + 16);
    }
    for (Py_ssize_t w = w0; w < w1; ++w) {
      PyObject** items = PySequence_Fast_ITEMS(*(PyObject**)((char*)c->wobj[w] + c->off_emb));
      for (Py_ssize_t k = 0; k < c->K; ++k) {
        PyObject* a = items[k];
        if (!PyArray_Check(a) || PyArray_TYPE((PyArrayObject*)a) != NPY_FLOAT32 || PyArray_NDIM((PyArrayObject*)a) != 1 ||
            PyArray_DIM((PyArrayObject*)a, 0) != c->E[k] || !PyArray_IS_C_CONTIGUOUS((PyArrayObject*)a)) {
          c->fallback = 1;
          return;
        }
        c->esrc[(size_t)k * c->B * c->L + (size_t)c->sb[w] * c->L + c->si[w]] = (const float*)PyArray_DATA((PyArrayObject*)a);
      }
    }
  }
  c->chunk_maxlen[chunk] = maxlen;
}

/* code points + their hash (cp_off is known), then the word ids of the tokens the table knows; -2 = ask Python */
static void par_sweep_strings(void* vctx, int chunk, int n_chunks) {
  ParCtx* c = (ParCtx*)vctx;
  const Py_ssize_t w0 = c->N * chunk / n_chunks, w1 = c->N * (chunk + 1) / n_chunks;
  enum { PFD = 8 };
  for (Py_ssize_t w = w0; w < w1; ++w) {
    if (w + PFD < w1) __builtin_prefetch(c->wstr[w + PFD]);
    PyObject* t = c->wstr[w];
    const Py_ssize_t len = c->wlen[w];
    const int kind = PyUnicode_KIND(t);
    const void* data = PyUnicode_DATA(t);
    int32_t* out = c->cps + c->cp_off[w];
    uint64_t h = 1469598103934665603ULL;
    for (Py_ssize_t i = 0; i < len; ++i) {
      const Py_UCS4 cp = PyUnicode_READ(kind, data, i);
      out[i] = (int32_t)cp;
      h = (h ^ (uint64_t)cp) * 1099511628211ULL;
      if ((Py_ssize_t)cp >= c->top || c->tab_char[cp] < 0) c->missing = 1;
    }
    c->whash[w] = h;
    __builtin_prefetch(&c->wtab->slots[(size_t)(h ? h : 1) & c->wtab->mask]);
  }
  for (Py_ssize_t w = w0; w < w1; ++w) {
    const uint64_t h = c->whash[w] ? c->whash[w] : 1;
    const WidSlot* sl = wid_find(c->wtab, h, c->cps + c->cp_off[w], c->wlen[w]);
    c->wid[w] = sl->h != 0 ? sl->id : -2;
  }
}

typedef struct {
  const float* const* src; /* [rows] source vector of a row, NULL = pad row */
  float* dst;
  Py_ssize_t E, Epad, row0, row1;
} EmbJob;

static void* emb_worker(void* arg) {
  EmbJob* j = (EmbJob*)arg;
  for (Py_ssize_t r = j->row0; r < j->row1; ++r) {
    float* d = j->dst + r * j->Epad;
    const float* s = j->src[r];
    if (r + 4 < j->row1 && j->src[r + 4]) {
      const char* q = (const char*)j->src[r + 4];
      for (Py_ssize_t o = 0; o < j->E * 4; o += 256) __builtin_prefetch(q + o);
    }
    if (s) {
      memcpy(d, s, (size_t)j->E * 4);
      if (j->Epad > j->E) memset(d + j->E, 0, (size_t)(j->Epad - j->E) * 4);
    } else {
      memset(d, 0, (size_t)j->Epad * 4);
    }
  }
  return NULL;
}

#define FEAT_FAIL(...)                     \
  do {                                     \
    PyErr_Format(PyExc_ValueError, __VA_ARGS__); \
    goto fail;                             \
  } while (0)

/* featurize(sentences, word2int, wid_cache, unk, cp_tab, root, n_layers, n_case, L, emb_out, emb_dims, groups_addr, n_threads, ints_out)
 *   sentences : list of Sentence (attributes words: list of Word, lang_id: int)
 *   word2int  : dict lower-cased word -> id;  wid_cache: dict raw token -> id (filled here);  unk: id of <UNK>
 *   cp_tab    : C-contiguous int32 [2, top]: (char id, case id) by code point, -1 = not yet known
 *   root      : 1 = every sentence gets a root row in front (parser), 0 = not (tagger)
 *   n_layers  : gated layers of the character CNN;  n_case: > 0 = one combined id char * n_case + case per
 *               position (first layer by table look-up), 0 = separate character and case arrays
 *   L         : longest sentence of the batch (words)
 *   emb_out   : tuple of K writable float32 buffers of >= B*L*Epad_k elements;  emb_dims: tuple of (E_k, Epad_k)
 *   groups_addr: address of n_layers cube_conv_groups structs (ctypes array), filled here
 *   ints_out  : writable int32 buffer that receives the packed index arrays
 *   dedup     : (optional) 1 = words of the batch with the same spelling and language go through the character CNN
 *               once: the per-word arrays then hold the N_u distinct words, and two more arrays list, for every
 *               repeated word, its row of the word-level input and the row of the first occurrence to copy from
 * -> (int32 elements written to ints_out, sizes tuple of 10 (dedup: 12) ints, N (dedup: N_u), W, rows tuple of n_layers+1 ints)
 *    or a list of code points missing from cp_tab (the caller extends the table and calls again)
 *    or an int: the number of int32 elements ints_out must hold (it was too small; the caller calls again). */
static PyObject* featurize(PyObject* self, PyObject* args) {
  PyObject *sents, *word2int, *wid_cache, *emb_out, *emb_dims, *ints_out;
  PyArrayObject* cp_tab;
  long unk;
  int root, n_layers, n_case, n_threads, dedup = 0;
  Py_ssize_t L;
  unsigned long long groups_addr;
  if (!PyArg_ParseTuple(args, "O!O!OlO!iiinO!O!KiO|i", &PyList_Type, &sents, &PyDict_Type, &word2int, &wid_cache,
                        &unk, &PyArray_Type, &cp_tab, &root, &n_layers, &n_case, &L, &PyTuple_Type, &emb_out,
                        &PyTuple_Type, &emb_dims, &groups_addr, &n_threads, &ints_out, &dedup))
    return NULL;
  if (PyArray_TYPE(cp_tab) != NPY_INT32 || PyArray_NDIM(cp_tab) != 2 || PyArray_DIM(cp_tab, 0) != 2 ||
      !PyArray_IS_C_CONTIGUOUS(cp_tab)) {
    PyErr_SetString(PyExc_ValueError, "featurize: cp_tab must be a C-contiguous int32 [2, top] array");
    return NULL;
  }
  WidTable* wtab = NULL; /* the C-side token cache (wid_table_new), or a dict as in round 2 */
  if (PyCapsule_CheckExact(wid_cache)) {
    wtab = (WidTable*)PyCapsule_GetPointer(wid_cache, "cube_wid_table");
    if (!wtab) return NULL;
  } else if (!PyDict_Check(wid_cache)) {
    PyErr_SetString(PyExc_TypeError, "featurize: the token cache must be a dict or a wid_table_new() capsule");
    return NULL;
  }
  const Py_ssize_t K = PyTuple_GET_SIZE(emb_out);
  if (PyTuple_GET_SIZE(emb_dims) != K || n_layers < 1 || n_layers > 16 || root < 0 || root > 1 || L < 1 || K > 64) {
    PyErr_SetString(PyExc_ValueError, "featurize: bad n_layers / root / L / embedding arguments");
    return NULL;
  }
  if (g_sc_busy) {
    PyErr_SetString(PyExc_TypeError, "featurize: re-entered while another call is using its scratch buffers");
    return NULL;
  }
  g_sc_busy = 1;
  const Py_ssize_t top = PyArray_DIM(cp_tab, 1);
  const int32_t* tab_char = (const int32_t*)PyArray_DATA(cp_tab);
  const int32_t* tab_case = tab_char + top;
  const Py_ssize_t B = PyList_GET_SIZE(sents);
  const Py_ssize_t T = L + root;

  /* everything that must be released at the end */
  PyObject** wstr = NULL;   /* [N] the words' str objects (borrowed) */
  PyObject** wemb = NULL;   /* references to drop at the end (values that came from getattr) */
  int32_t *wlen = NULL, *wid = NULL, *sb = NULL, *si = NULL, *slen = NULL, *lang = NULL, *order = NULL, *cp_off = NULL;
  const float** esrc = NULL;
  Py_buffer ebuf[64];
  Py_ssize_t n_ebuf = 0, n_wstr = 0, n_wemb = 0;
  PyObject* missing = NULL;
  PyObject* result = NULL;
  Py_buffer ibuf;
  int have_ibuf = 0;
  WordLayout wl;
  wl.tp = NULL;
  int32_t* cps = NULL;      /* code points of all words, back to back (cp_off) */
  uint64_t* whash = NULL;   /* FNV-1a of a word's code points (the language is folded in where it is used) */
  EmbJob jobs[16 * 64];     /* the embedding copies run on worker threads beside the index-array pass */
  pthread_t th[16 * 64];
  int n_started = 0;

  Py_ssize_t E[64], Epad[64];
  if (PyObject_GetBuffer(ints_out, &ibuf, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) goto fail;
  have_ibuf = 1;
  for (Py_ssize_t k = 0; k < K; ++k) {
    PyObject* d = PyTuple_GET_ITEM(emb_dims, k);
    if (!PyTuple_Check(d) || PyTuple_GET_SIZE(d) != 2) FEAT_FAIL("featurize: emb_dims must hold (E, Epad) pairs");
    E[k] = PyLong_AsSsize_t(PyTuple_GET_ITEM(d, 0));
    Epad[k] = PyLong_AsSsize_t(PyTuple_GET_ITEM(d, 1));
    if (PyErr_Occurred() || E[k] < 1 || Epad[k] < E[k]) FEAT_FAIL("featurize: bad embedding sizes");
    if (PyObject_GetBuffer(PyTuple_GET_ITEM(emb_out, k), &ebuf[k], PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) goto fail;
    ++n_ebuf;
    if (ebuf[k].len < (Py_ssize_t)(B * L * Epad[k] * 4)) FEAT_FAIL("featurize: embedding buffer %zd is too small", k);
  }

  double t_start = now_s();
  /* ---- pass 0: sizes */
  slen = (int32_t*)sc_alloc(0, sizeof(int32_t) * (size_t)(B + 1), 0);
  lang = (int32_t*)sc_alloc(1, sizeof(int32_t) * (size_t)(B + 1), 0);
  if (!slen || !lang) { PyErr_NoMemory(); goto fail; }
  Py_ssize_t N = 0;
  for (Py_ssize_t b = 0; b < B; ++b) {
    PyObject* s = PyList_GET_ITEM(sents, b);
    PyObject* ws = PyObject_GetAttr(s, str_words);
    if (!ws) goto fail;
    if (!PyList_Check(ws)) { Py_DECREF(ws); FEAT_FAIL("featurize: sentence %zd: .words is not a list", b); }
    const Py_ssize_t n = PyList_GET_SIZE(ws);
    Py_DECREF(ws); /* the sentence keeps it alive */
    if (n > L) FEAT_FAIL("featurize: sentence %zd has %zd words, L = %zd", b, n, L);
    slen[b] = (int32_t)n;
    N += n;
    PyObject* lg = PyObject_GetAttr(s, str_lang_id);
    if (!lg) goto fail;
    const long lv = PyLong_AsLong(lg);
    Py_DECREF(lg);
    if (lv == -1 && PyErr_Occurred()) goto fail;
    lang[b] = (int32_t)(lv + 1); /* utils.py:266: x_lang = lang_id + 1 */
  }
  if (N == 0) FEAT_FAIL("featurize: the batch holds no words");
  wstr = (PyObject**)sc_alloc(2, sizeof(PyObject*) * (size_t)N, 0);
  wemb = (PyObject**)sc_alloc(3, sizeof(PyObject*) * (size_t)(2 * N), 0); /* references taken on the getattr path */
  wlen = (int32_t*)sc_alloc(4, sizeof(int32_t) * (size_t)N, 0);
  wid = (int32_t*)sc_alloc(5, sizeof(int32_t) * (size_t)N, 0);
  sb = (int32_t*)sc_alloc(6, sizeof(int32_t) * (size_t)N, 0);
  si = (int32_t*)sc_alloc(7, sizeof(int32_t) * (size_t)N, 0);
  order = (int32_t*)sc_alloc(8, sizeof(int32_t) * (size_t)N, 0);
  cp_off = (int32_t*)sc_alloc(9, sizeof(int32_t) * (size_t)(N + 1), 0);
  esrc = (const float**)sc_alloc(10, sizeof(float*) * (size_t)(B * L * (K ? K : 1)), 1); /* NULL = pad row */
  whash = (uint64_t*)sc_alloc(17, sizeof(uint64_t) * (size_t)N, 0);
  cps = (int32_t*)sc_alloc(14, sizeof(int32_t) * (size_t)(8 * N + 64), 0); /* grown in sweep C when the words are longer */
  if (!wstr || !wemb || !wlen || !wid || !sb || !si || !order || !cp_off || !esrc || !whash || !cps) { PyErr_NoMemory(); goto fail; }

  g_t[0] += now_s() - t_start; t_start = now_s();
  /* ---- pass 1: the objects, in sweeps over all words with the next level's memory prefetched a few words
   * ahead (a Word, its str, its emb list, the list's item array, the arrays and their data are separate heap
   * objects: walked word by word, every step is a cache miss).  The GIL is held throughout and nothing below
   * runs Python code, so borrowed pointers stay valid until the references are taken in sweep C. */
  enum { PF = 12 };
  Py_ssize_t i = 0, total_chars = 0, W = 0;
  {
    PyObject** wobj = (PyObject**)sc_alloc(11, sizeof(PyObject*) * (size_t)N, 0);
    PyObject** wel = (PyObject**)sc_alloc(12, sizeof(PyObject*) * (size_t)N, 0);
    PyObject*** witems = (PyObject***)sc_alloc(13, sizeof(PyObject**) * (size_t)N, 0);
    int bad = (!wobj || !wel || !witems);
    if (bad) PyErr_NoMemory();
    /* sweep A: the Word objects of every sentence */
    for (Py_ssize_t b = 0; !bad && b < B; ++b) {
      PyObject* ws = PyObject_GetAttr(PyList_GET_ITEM(sents, b), str_words);
      if (!ws) { bad = 1; break; }
      if (!PyList_Check(ws) || PyList_GET_SIZE(ws) != slen[b]) {
        Py_DECREF(ws);
        PyErr_Format(PyExc_ValueError, "featurize: sentence %zd changed", b);
        bad = 1;
        break;
      }
      for (Py_ssize_t j = 0; j < slen[b]; ++j, ++i) {
        wobj[i] = PyList_GET_ITEM(ws, j);
        sb[i] = (int32_t)b;
        si[i] = (int32_t)j;
      }
      Py_DECREF(ws); /* the sentence keeps its list */
    }
    /* The plain format on the thread pool (par_sweep_*): sweeps B .. E and the cached word ids, in chunks of words. */
    int par_done = 0;
    if (!bad && wtab && n_threads > 1 && N >= 1024) {
      PyTypeObject* tp = Py_TYPE(wobj[0]);
      if (wl.tp != tp) word_layout(tp, &wl);
      if (wl.off[F_WORD] >= 0 && (K == 0 || wl.off[F_EMB] >= 0)) {
        ParCtx pc;
        memset(&pc, 0, sizeof(pc));
        pc.N = N; pc.K = K; pc.B = B; pc.L = L; pc.top = top;
        pc.wobj = wobj; pc.wstr = wstr; pc.tp = tp; pc.off_word = wl.off[F_WORD]; pc.off_emb = K > 0 ? wl.off[F_EMB] : 0;
        pc.E = E; pc.wlen = wlen; pc.wid = wid; pc.cp_off = cp_off; pc.sb = sb; pc.si = si; pc.tab_char = tab_char;
        pc.esrc = esrc; pc.whash = whash; pc.wtab = wtab;
        int nch = 4 * n_threads;
        if (nch > 64) nch = 64;
        if (nch > N / 256) nch = (int)(N / 256);
        pool_run(par_sweep_objects, &pc, nch, n_threads);
        if (!pc.fallback) {
          total_chars = 0;
          for (Py_ssize_t w = 0; w < N; ++w) { cp_off[w] = (int32_t)total_chars; total_chars += wlen[w]; }
          cp_off[N] = (int32_t)total_chars;
          for (int c = 0; c < nch; ++c) if (pc.chunk_maxlen[c] > W) W = pc.chunk_maxlen[c];
          if (total_chars > 0x3fffffff) { PyErr_SetString(PyExc_ValueError, "featurize: too many characters in one batch"); bad = 1; }
          if (!bad && (size_t)(total_chars + 1) * sizeof(int32_t) > g_sc_cap[14]) {
            const size_t cap = 2 * (size_t)(total_chars + 1) * sizeof(int32_t);
            int32_t* grown = (int32_t*)malloc(cap);
            if (!grown) { PyErr_NoMemory(); bad = 1; }
            else { free(g_sc[14]); g_sc[14] = grown; g_sc_cap[14] = cap; cps = grown; }
          }
          if (!bad) {
            pc.cps = cps;
            pool_run(par_sweep_strings, &pc, nch, n_threads);
            if (pc.missing) { /* code points the table does not know yet go back to the caller, in order of appearance */
              missing = PyList_New(0);
              if (!missing) bad = 1;
              for (Py_ssize_t c = 0; !bad && c < total_chars; ++c) {
                const int32_t cp = cps[c];
                if ((Py_ssize_t)cp >= top || tab_char[cp] < 0) {
                  PyObject* v = PyLong_FromLong((long)cp);
                  if (!v || PyList_Append(missing, v) != 0) { Py_XDECREF(v); bad = 1; break; }
                  Py_DECREF(v);
                }
              }
            }
            for (Py_ssize_t w = 0; !bad && w < N; ++w) { /* tokens seen for the first time: word2int[word.lower()], once */
              if (wid[w] != -2) continue;
              const uint64_t h = whash[w] ? whash[w] : 1;
              const int32_t* cp = cps + cp_off[w];
              WidSlot* sl = wid_find(wtab, h, cp, wlen[w]); /* (an earlier word of this batch may have brought it in) */
              if (sl->h != 0) { wid[w] = sl->id; continue; }
              PyObject* lo = PyObject_CallMethodNoArgs(wstr[w], str_lower);
              if (!lo) { bad = 1; break; }
              PyObject* id = PyDict_GetItemWithError(word2int, lo);
              Py_DECREF(lo);
              if (!id && PyErr_Occurred()) { bad = 1; break; }
              const long v = id ? PyLong_AsLong(id) : unk;
              if (v == -1 && PyErr_Occurred()) { bad = 1; break; }
              wid[w] = (int32_t)v;
              if (wid_insert(wtab, h, cp, wlen[w], (int32_t)v) != 0) { PyErr_NoMemory(); bad = 1; break; }
            }
          }
          par_done = 1;
        }
      }
    }
    /* sweep B: the str and the emb list of every word (slot reads: borrowed; getattr results are kept) */
    for (Py_ssize_t w = 0; !bad && !par_done && w < N; ++w) {
      if (w + PF < N) { __builtin_prefetch(wobj[w + PF]); __builtin_prefetch((char*)wobj[w + PF] + 64); }
      int owned = 0;
      PyObject* t = word_peek(wobj[w], &wl, F_WORD, &owned);
      if (!t) { bad = 1; break; }
      wstr[w] = t;
      if (owned) wemb[n_wemb++] = t; /* `wemb` doubles as the list of references to drop at the end */
      if (K > 0) {
        PyObject* el = word_peek(wobj[w], &wl, F_EMB, &owned);
        if (!el) { bad = 1; break; }
        wel[w] = el;
        if (owned) wemb[n_wemb++] = el;
      }
    }
    g_t[8] += now_s() - t_start;
    double t_sw = now_s();
    /* sweep C: the strings: length, word id through the raw-token cache (lower-cased look-up, utils.py:284-289) */
    for (Py_ssize_t w = 0; !bad && !par_done && w < N; ++w) {
      if (w + PF < N) {
        __builtin_prefetch(wstr[w + PF]);
        if (K > 0) __builtin_prefetch(wel[w + PF]);
      }
      PyObject* t = wstr[w];
      if (!PyUnicode_CheckExact(t) || PyUnicode_READY(t) != 0) {
        /* a str subclass could run Python code in .lower() / __hash__ / __eq__ below: not the fast format (TypeError ->
         * the caller's Python featurisation); anything else is an error of the document */
        if (PyUnicode_Check(t)) PyErr_Format(PyExc_TypeError, "featurize: word %d of sentence %d is a str subclass", (int)si[w], (int)sb[w]);
        else PyErr_Format(PyExc_ValueError, "featurize: word %d of sentence %d is not a str", (int)si[w], (int)sb[w]);
        bad = 1;
        break;
      }
      const Py_ssize_t len = PyUnicode_GET_LENGTH(t);
      wlen[w] = (int32_t)len;
      if (len > W) W = len;
      cp_off[w] = (int32_t)total_chars;
      /* the code points, while the string is in the cache: -> cps, their hash, and the ones the table does not know */
      if ((size_t)(total_chars + len + 1) * sizeof(int32_t) > g_sc_cap[14]) {
        const size_t cap = 2 * (size_t)(total_chars + len + 1) * sizeof(int32_t);
        int32_t* grown = (int32_t*)malloc(cap);
        if (!grown) { PyErr_NoMemory(); bad = 1; break; }
        memcpy(grown, cps, sizeof(int32_t) * (size_t)total_chars);
        free(g_sc[14]);
        g_sc[14] = grown;
        g_sc_cap[14] = cap;
        cps = grown;
      }
      {
        const int kind = PyUnicode_KIND(t);
        const void* data = PyUnicode_DATA(t);
        uint64_t h = 1469598103934665603ULL;
        for (Py_ssize_t c = 0; c < len; ++c) {
          const Py_UCS4 cp = PyUnicode_READ(kind, data, c);
          cps[total_chars + c] = (int32_t)cp;
          h = (h ^ (uint64_t)cp) * 1099511628211ULL;
          if ((Py_ssize_t)cp >= top || tab_char[cp] < 0) {
            if (!missing) missing = PyList_New(0);
            PyObject* v = missing ? PyLong_FromLong((long)cp) : NULL;
            if (!v || PyList_Append(missing, v) != 0) { Py_XDECREF(v); bad = 1; break; }
            Py_DECREF(v);
          }
        }
        if (bad) break;
        whash[w] = h;
        if (wtab) __builtin_prefetch(&wtab->slots[(size_t)(h ? h : 1) & wtab->mask]); /* read by sweep C2 */
      }
      total_chars += len;
      if (wtab) continue;
      PyObject* id = PyDict_GetItemWithError(wid_cache, t); /* borrowed */
      if (!id) {
        if (PyErr_Occurred()) { bad = 1; break; }
        PyObject* lo = PyObject_CallMethodNoArgs(t, str_lower);
        if (!lo) { bad = 1; break; }
        id = PyDict_GetItemWithError(word2int, lo);
        Py_DECREF(lo);
        if (!id && PyErr_Occurred()) { bad = 1; break; }
        long v = id ? PyLong_AsLong(id) : unk;
        if (v == -1 && PyErr_Occurred()) { bad = 1; break; }
        wid[w] = (int32_t)v;
        if (PyDict_GET_SIZE(wid_cache) > (1 << 20)) PyDict_Clear(wid_cache);
        PyObject* pv = PyLong_FromLong(v);
        if (!pv || PyDict_SetItem(wid_cache, t, pv) != 0) { Py_XDECREF(pv); bad = 1; break; }
        Py_DECREF(pv);
      } else {
        const long v = PyLong_AsLong(id);
        if (v == -1 && PyErr_Occurred()) { bad = 1; break; }
        wid[w] = (int32_t)v;
      }
    }
    /* sweep C2: the word ids from the C-side cache (its slots were prefetched by sweep C); a token seen for the first
     * time asks Python once: word2int[word.lower()], utils.py:284-289 */
    if (wtab && !bad && !par_done) {
      cp_off[N] = (int32_t)total_chars;
      for (Py_ssize_t w = 0; w < N; ++w) {
        const uint64_t h = whash[w] ? whash[w] : 1; /* 0 marks an empty slot */
        const int32_t* cp = cps + cp_off[w];
        WidSlot* sl = wid_find(wtab, h, cp, wlen[w]);
        if (sl->h != 0) { wid[w] = sl->id; continue; }
        PyObject* lo = PyObject_CallMethodNoArgs(wstr[w], str_lower);
        if (!lo) { bad = 1; break; }
        PyObject* id = PyDict_GetItemWithError(word2int, lo);
        Py_DECREF(lo);
        if (!id && PyErr_Occurred()) { bad = 1; break; }
        const long v = id ? PyLong_AsLong(id) : unk;
        if (v == -1 && PyErr_Occurred()) { bad = 1; break; }
        wid[w] = (int32_t)v;
        if (wid_insert(wtab, h, cp, wlen[w], (int32_t)v) != 0) { PyErr_NoMemory(); bad = 1; break; }
      }
    }
    g_t[9] += now_s() - t_sw; t_sw = now_s();
    if (!bad && !par_done && K > 0) {
      int good = 1;
      /* sweep D: the emb lists -> their item arrays */
      for (Py_ssize_t w = 0; good && w < N; ++w) {
        PyObject* el = wel[w];
        if (!(PyList_Check(el) || PyTuple_Check(el)) || PySequence_Fast_GET_SIZE(el) != K) { good = 0; break; }
        witems[w] = PySequence_Fast_ITEMS(el);
        __builtin_prefetch(witems[w]); /* the item array, read by sweep E */
      }
      /* sweep E: the array objects, level by level (borrowed: the GIL is held until the copies are done) */
      PyObject** arr = (PyObject**)wobj; /* the Word pointers are no longer needed */
      for (Py_ssize_t k = 0; good && k < K; ++k) {
        for (Py_ssize_t w = 0; w < N; ++w) {
          arr[w] = witems[w][k];
          __builtin_prefetch((char*)arr[w] + 16); /* the PyArrayObject header, read by the next loop */
        }
        for (Py_ssize_t w = 0; w < N; ++w) {
          PyObject* a = arr[w];
          if (!PyArray_Check(a) || PyArray_TYPE((PyArrayObject*)a) != NPY_FLOAT32 || PyArray_NDIM((PyArrayObject*)a) != 1 ||
              PyArray_DIM((PyArrayObject*)a, 0) != E[k] || !PyArray_IS_C_CONTIGUOUS((PyArrayObject*)a)) {
            good = 0;
            break;
          }
          esrc[(size_t)k * B * L + (size_t)sb[w] * L + si[w]] = (const float*)PyArray_DATA((PyArrayObject*)a);
        }
      }
      if (!good) {
        PyErr_SetString(PyExc_TypeError, "featurize: word.emb is not a list of contiguous float32 vectors of the model's sizes");
        bad = 1;
      }
    }
    g_t[10] += now_s() - t_sw;
    if (bad) goto fail;
  }
  cp_off[N] = (int32_t)total_chars;

  g_t[1] += now_s() - t_start; t_start = now_s();
  /* ---- external embeddings -> padded [B*L, Epad] rows: worker threads, started here and joined at the end, so that
   * the copies (memory-bound) run beside the index-array pass below.  The GIL stays held all the while: the vectors
   * are borrowed, and nothing else of this process's Python may run meanwhile. */
  if (K > 0) {
    int nt = n_threads < 1 ? 1 : (n_threads > 16 ? 16 : n_threads);
    const Py_ssize_t rows = B * L;
    const int small = rows * Epad[0] < (1 << 16); /* a thread costs more than such a copy */
    if (small) nt = 1;
    int nj = 0;
    for (Py_ssize_t k = 0; k < K && nj < 16 * 64; ++k)
      for (int t = 0; t < nt && nj < 16 * 64; ++t) {
        EmbJob* j = &jobs[nj++];
        j->src = esrc + (size_t)k * B * L;
        j->dst = (float*)ebuf[k].buf;
        j->E = E[k];
        j->Epad = Epad[k];
        j->row0 = rows * t / nt;
        j->row1 = rows * (t + 1) / nt;
      }
    for (; !small && n_started < nj; ++n_started)
      if (pthread_create(&th[n_started], NULL, emb_worker, &jobs[n_started]) != 0) break;
    for (int t = n_started; t < nj; ++t) emb_worker(&jobs[t]); /* (no thread to be had: the caller's thread copies) */
  }
  /* ---- characters: code point -> (char id, case id); unknown code points (found in sweep C) go back to the caller */
  {
    if (missing) {
      result = missing;
      missing = NULL;
      goto done;
    }
    double t_sub = now_s();
    g_t[4] += t_sub - t_start;

    /* ---- repeated words: rep[w] = the first word of the batch with the same code points and language (itself if
     * there is none).  The character CNN's output for a word depends on nothing else (and on W, which is the batch's). */
    int32_t* rep = (int32_t*)sc_alloc(15, sizeof(int32_t) * (size_t)N, 0);
    if (!rep) { PyErr_NoMemory(); goto fail; }
    size_t cap = 16;
    while (cap < 2 * (size_t)N) cap <<= 1;
    int32_t* slots = dedup ? (int32_t*)sc_alloc(16, sizeof(int32_t) * cap, 0) : NULL;
    if (dedup && !slots) { PyErr_NoMemory(); goto fail; }
    /* From here to the join of the embedding copies nothing touches a Python object or the interpreter: the index arrays
     * are built from this call's own scratch copies (code points, lengths, ids) and the copies read the vectors' data
     * through pointers taken above.  The GIL is released for that stretch, so that a caller that featurises on a thread
     * of its own (Parser.feature_thread) overlaps it with the launches and the write-back of other batches.  What keeps
     * the borrowed data alive meanwhile is the caller's contract: the sentences of the batch being featurised are not
     * modified until the call returns (Parser.process never touches a batch before its featurisation is done). */
    Py_ssize_t need_ints = -1;
    PyThreadState* ts_saved = PyEval_SaveThread();
    Py_ssize_t n_dup = 0;
    if (dedup) {
      memset(slots, 0xff, sizeof(int32_t) * cap);
      for (Py_ssize_t w = 0; w < N; ++w) {
        const int32_t* c = cps + cp_off[w];
        const int32_t lw = lang[sb[w]];
        const uint64_t h = (whash[w] ^ (uint64_t)(uint32_t)lw) * 1099511628211ULL;
        size_t sl = (size_t)(h ^ (h >> 29)) & (cap - 1);
        for (;;) {
          const int32_t v = slots[sl];
          if (v < 0) { slots[sl] = (int32_t)w; rep[w] = (int32_t)w; break; }
          if (wlen[v] == wlen[w] && lang[sb[v]] == lw && memcmp(cps + cp_off[v], c, sizeof(int32_t) * (size_t)wlen[w]) == 0) {
            rep[w] = v;
            ++n_dup;
            break;
          }
          sl = (sl + 1) & (cap - 1);
        }
      }
    } else {
      for (Py_ssize_t w = 0; w < N; ++w) rep[w] = (int32_t)w;
    }
    const Py_ssize_t NU = N - n_dup; /* words the character CNN sees */
    g_t[5] += now_s() - t_sub; t_sub = now_s();

    /* ---- word-length groups of the ragged character CNN (Parser._char_layout) */
    const int MAXG = CUBE_MAX_CONV_GROUPS;
    int cnt[CUBE_MAX_CONV_GROUPS] = {0}, gmax_key[CUBE_MAX_CONV_GROUPS] = {0}, start_key[CUBE_MAX_CONV_GROUPS + 1];
    for (Py_ssize_t w = 0; w < N; ++w) {
      if (rep[w] != w) continue;
      int key = wlen[w] < 1 ? 1 : wlen[w];
      if (key > MAXG - 1) key = MAXG - 1;
      cnt[key]++;
      const int l1 = wlen[w] < 1 ? 1 : wlen[w];
      if (l1 > gmax_key[key]) gmax_key[key] = l1;
    }
    start_key[0] = 0;
    for (int k = 0; k < MAXG; ++k) start_key[k + 1] = start_key[k] + cnt[k];
    {
      int fill[CUBE_MAX_CONV_GROUPS];
      for (int k = 0; k < MAXG; ++k) fill[k] = start_key[k];
      for (Py_ssize_t w = 0; w < N; ++w) { /* stable counting sort */
        if (rep[w] != w) continue;
        int key = wlen[w] < 1 ? 1 : wlen[w];
        if (key > MAXG - 1) key = MAXG - 1;
        order[fill[key]++] = (int32_t)w;
      }
    }
    int G = 0, gkey[CUBE_MAX_CONV_GROUPS];
    for (int k = 0; k < MAXG; ++k)
      if (cnt[k]) gkey[G++] = k;
    const int n = n_layers;
    /* seq[k][g], rows_g[k][g], row0[k][g] for layouts k = 0..n */
    long seq[17][CUBE_MAX_CONV_GROUPS], row0[17][CUBE_MAX_CONV_GROUPS], rows_total[17];
    for (int k = 0; k <= n; ++k) {
      long run = 0;
      for (int g = 0; g < G; ++g) {
        long sq = gmax_key[gkey[g]] + 2L * (n - k);
        if (sq > W) sq = W;
        seq[k][g] = sq;
        row0[k][g] = run;
        run += sq * cnt[gkey[g]];
      }
      rows_total[k] = run;
    }
    cube_conv_groups* cg = (cube_conv_groups*)(uintptr_t)groups_addr;
    for (int l = 0; l < n; ++l) {
      cg[l].n = G;
      for (int g = 0; g < G; ++g) {
        cg[l].row0_in[g] = (int)row0[l][g];
        cg[l].n_rows[g] = (int)(seq[l][g] * cnt[gkey[g]]);
        cg[l].seq_in[g] = (int)seq[l][g];
        cg[l].row0_out[g] = (int)row0[l + 1][g];
        cg[l].seq_out[g] = (int)seq[l + 1][g];
        cg[l].word0[g] = start_key[gkey[g]];
      }
    }

    g_t[6] += now_s() - t_sub; t_sub = now_s();
    /* ---- the packed int32 upload (order and 4-element alignment of Parser._prepare) */
    const Py_ssize_t n_char = rows_total[0];
    const int n_parts = dedup ? 12 : 10;
    Py_ssize_t sizes[12] = {n_char, n_case > 0 ? 0 : n_char, NU, NU, NU, B * T, B * T, B, B, (Py_ssize_t)n * NU, n_dup, n_dup};
    Py_ssize_t offs[13];
    offs[0] = 0;
    for (int p = 0; p < n_parts; ++p) offs[p + 1] = offs[p] + (sizes[p] + 3) / 4 * 4;
    const Py_ssize_t n_ints = offs[n_parts];
    if ((Py_ssize_t)(ibuf.len / 4) < n_ints) { /* the caller's index buffer is too small: tell it the size, it calls again */
      need_ints = n_ints;
      goto arrays_done;
    }
    int32_t* P = (int32_t*)ibuf.buf;
    memset(P, 0, (size_t)n_ints * 4);
    {
      int32_t* pc = P + offs[0];
      int32_t* pk = P + offs[1];
      Py_ssize_t r = 0;
      for (int g = 0; g < G; ++g) {
        const long q0 = seq[0][g];
        for (int a = start_key[gkey[g]]; a < start_key[gkey[g] + 1]; ++a) {
          const int w = order[a];
          const int32_t* c = cps + cp_off[w];
          const long len = wlen[w];
          for (long pos = 0; pos < q0; ++pos, ++r) {
            int32_t ch = 0, ca = 0;
            if (pos < len) { ch = tab_char[c[pos]]; ca = tab_case[c[pos]]; }
            if (n_case > 0) pc[r] = ch * n_case + ca;
            else { pc[r] = ch; pk[r] = ca; }
          }
        }
      }
    }
    g_t[7] += now_s() - t_sub;
    {
      int32_t *p_lang = P + offs[2], *p_len = P + offs[3], *p_rowmap = P + offs[4], *p_widx = P + offs[5];
      int32_t *p_lrows = P + offs[6], *p_root = P + offs[7], *p_lens = P + offs[8], *p_wrows = P + offs[9];
      for (Py_ssize_t a = 0; a < NU; ++a) {
        const int w = order[a];
        p_lang[a] = lang[sb[w]];
        p_len[a] = wlen[w];
        p_rowmap[a] = (int32_t)(sb[w] * T + si[w] + root);
      }
      if (dedup) {
        int32_t *p_dst = P + offs[10], *p_src = P + offs[11];
        Py_ssize_t d = 0;
        for (Py_ssize_t w = 0; w < N; ++w)
          if (rep[w] != w) {
            p_dst[d] = (int32_t)(sb[w] * T + si[w] + root);
            p_src[d++] = (int32_t)(sb[rep[w]] * T + si[rep[w]] + root);
          }
      }
      for (Py_ssize_t w = 0; w < N; ++w) p_widx[sb[w] * T + si[w] + root] = wid[w];
      for (Py_ssize_t b = 0; b < B; ++b) {
        for (Py_ssize_t t = 0; t < T; ++t) p_lrows[b * T + t] = lang[b];
        p_root[b] = (int32_t)(b * T);
        p_lens[b] = slen[b];
      }
      for (int l = 0; l < n; ++l)
        for (int g = 0; g < G; ++g) {
          const int s0 = start_key[gkey[g]];
          for (int a = s0; a < start_key[gkey[g] + 1]; ++a)
            p_wrows[(Py_ssize_t)l * NU + a] = (int32_t)(row0[l + 1][g] + (long)(a - s0) * seq[l + 1][g]);
        }
    }

    g_t[2] += now_s() - t_start; t_start = now_s();
  arrays_done:
    /* the embedding copies started above: what is left of them after the index arrays */
    for (int t = 0; t < n_started; ++t) pthread_join(th[t], NULL);
    n_started = 0;
    PyEval_RestoreThread(ts_saved);
    if (need_ints >= 0) {
      result = PyLong_FromSsize_t(need_ints);
      goto done;
    }

    g_t[3] += now_s() - t_start; t_start = now_s();
    ++g_calls;
    PyObject* rows_t = PyTuple_New(n + 1);
    PyObject* sizes_t = PyTuple_New(n_parts);
    if (!rows_t || !sizes_t) { Py_XDECREF(rows_t); Py_XDECREF(sizes_t); goto fail; }
    for (int k = 0; k <= n; ++k) PyTuple_SET_ITEM(rows_t, k, PyLong_FromLong(rows_total[k]));
    for (int p = 0; p < n_parts; ++p) PyTuple_SET_ITEM(sizes_t, p, PyLong_FromSsize_t(sizes[p]));
    result = Py_BuildValue("(nNnnN)", n_ints, sizes_t, NU, W, rows_t);
  }
  goto done;
fail:
  Py_XDECREF(missing);
  result = NULL;
done:
  for (int t = 0; t < n_started; ++t) pthread_join(th[t], NULL); /* (early exits: the copies write into ebuf) */
  if (have_ibuf) PyBuffer_Release(&ibuf);
  for (Py_ssize_t k = 0; k < n_ebuf; ++k) PyBuffer_Release(&ebuf[k]);
  if (wemb) for (Py_ssize_t w = 0; w < n_wemb; ++w) Py_XDECREF(wemb[w]);
  g_sc_busy = 0;
  return result;
}

/* write_back_flat(sentences, ids, tables): ids C-contiguous int32 [n_words, 5] = (head, label, upos, xpos, attrs)
 * of every word of `sentences` in order (the layout in which a shard's results travel between ranks,
 * cube_amd/sharding.py); tables as for write_back. */
static PyObject* write_back_flat(PyObject* self, PyObject* args) {
  PyObject *sents, *tabs;
  PyArrayObject* ids;
  if (!PyArg_ParseTuple(args, "O!O!O!", &PyList_Type, &sents, &PyArray_Type, &ids, &PyTuple_Type, &tabs)) return NULL;
  if (PyArray_TYPE(ids) != NPY_INT32 || PyArray_NDIM(ids) != 2 || PyArray_DIM(ids, 1) != 5 || !PyArray_IS_C_CONTIGUOUS(ids) ||
      PyTuple_GET_SIZE(tabs) != 4) {
    PyErr_SetString(PyExc_ValueError, "write_back_flat: ids must be C-contiguous int32 [n_words, 5], tables a 4-tuple");
    return NULL;
  }
  PyObject* tab[4];
  Py_ssize_t ntab[4];
  for (int t = 0; t < 4; ++t) {
    tab[t] = PyTuple_GET_ITEM(tabs, t);
    if (!PyList_Check(tab[t])) {
      PyErr_SetString(PyExc_ValueError, "write_back_flat: the id -> string tables must be lists");
      return NULL;
    }
    ntab[t] = PyList_GET_SIZE(tab[t]);
  }
  static const int field_of[4] = {F_LABEL, F_UPOS, F_XPOS, F_ATTRS};
  const int32_t* R = (const int32_t*)PyArray_DATA(ids);
  const Py_ssize_t total = PyArray_DIM(ids, 0);
  Py_ssize_t k = 0;
  WordLayout wl;
  wl.tp = NULL;
  const Py_ssize_t B = PyList_GET_SIZE(sents);
  for (Py_ssize_t b = 0; b < B; ++b) {
    PyObject* ws = PyObject_GetAttr(PyList_GET_ITEM(sents, b), str_words);
    if (!ws) return NULL;
    if (!PyList_Check(ws) || k + PyList_GET_SIZE(ws) > total) {
      Py_DECREF(ws);
      PyErr_Format(PyExc_ValueError, "write_back_flat: the sentences hold more words than the %zd result rows", total);
      return NULL;
    }
    const Py_ssize_t n = PyList_GET_SIZE(ws);
    for (Py_ssize_t j = 0; j < n; ++j, ++k) {
      PyObject* w = PyList_GET_ITEM(ws, j);
      if (j + 8 < n) { __builtin_prefetch(PyList_GET_ITEM(ws, j + 8), 1); __builtin_prefetch((char*)PyList_GET_ITEM(ws, j + 8) + 64, 1); }
      PyObject* h = PyLong_FromLong((long)R[k * 5]);
      if (!h || word_set(w, &wl, F_HEAD, h) != 0) { Py_XDECREF(h); Py_DECREF(ws); return NULL; }
      Py_DECREF(h);
      for (int t = 0; t < 4; ++t) {
        const int32_t id = R[k * 5 + 1 + t];
        if (id < 0 || id >= ntab[t]) {
          Py_DECREF(ws);
          PyErr_Format(PyExc_IndexError, "write_back_flat: id %d outside table %d", (int)id, t);
          return NULL;
        }
        if (word_set(w, &wl, field_of[t], PyList_GET_ITEM(tab[t], id)) != 0) { Py_DECREF(ws); return NULL; }
      }
    }
    Py_DECREF(ws);
  }
  if (k != total) {
    PyErr_Format(PyExc_ValueError, "write_back_flat: %zd result rows for %zd words", total, k);
    return NULL;
  }
  Py_RETURN_NONE;
}

/* timing() -> (calls, seconds in: sizes pass, object pass, index arrays, wait for the embedding copies, and of the index
 * arrays: code points, repeated words, length groups, character rows) since the last call */
static PyObject* timing(PyObject* self, PyObject* args) {
  PyObject* r = Py_BuildValue("(lddddddddddd)", g_calls, g_t[0], g_t[1], g_t[2], g_t[3], g_t[4], g_t[5], g_t[6], g_t[7], g_t[8], g_t[9], g_t[10]);
  g_calls = 0;
  for (int i = 0; i < 12; ++i) g_t[i] = 0.0;
  return r;
}

/* max_words(sentences) -> the longest sentence's word count (L of a batch), without a Python-level loop */
static PyObject* max_words(PyObject* self, PyObject* args) {
  PyObject* sents;
  if (!PyArg_ParseTuple(args, "O!", &PyList_Type, &sents)) return NULL;
  Py_ssize_t mx = 0;
  for (Py_ssize_t b = 0; b < PyList_GET_SIZE(sents); ++b) {
    PyObject* ws = PyObject_GetAttr(PyList_GET_ITEM(sents, b), str_words);
    if (!ws) return NULL;
    const Py_ssize_t m = PyObject_Length(ws);
    Py_DECREF(ws);
    if (m < 0) return NULL;
    if (m > mx) mx = m;
  }
  return PyLong_FromSsize_t(mx);
}

/* geometry(sentences[, n_threads]) -> (int32 [n] words per sentence, int32 [n] longest word per sentence): what decides
 * the padding of a reference batch (longest sentence, longest word), for merging reference batches of equal geometry
 * (cube_amd/sharding.py) without a Python-level pass over every word.  This pass runs BEFORE the first batch of a
 * document can be enqueued (nothing overlaps it), so plain-format documents (every Word of one class with `word` in
 * __slots__, exact str) are read on the thread pool, as featurize reads them: the GIL is held by the caller, the workers
 * only read object memory. */
typedef struct {
  Py_ssize_t n;
  PyObject** lists;  /* [n] the sentences' word lists (borrowed; the sentences keep them) */
  PyTypeObject* tp;
  Py_ssize_t off_word;
  int32_t* wmax;
  volatile int fallback;
} GeoCtx;

static void par_geometry(void* vctx, int chunk, int n_chunks) {
  GeoCtx* c = (GeoCtx*)vctx;
  const Py_ssize_t b0 = c->n * chunk / n_chunks, b1 = c->n * (chunk + 1) / n_chunks;
  enum { SPAN = 64 };
  PyObject* strs[SPAN];
  for (Py_ssize_t b = b0; b < b1; ++b) {
    if (b + 1 < b1) __builtin_prefetch(((PyListObject*)c->lists[b + 1])->ob_item);
    const Py_ssize_t m = PyList_GET_SIZE(c->lists[b]);
    PyObject** items = ((PyListObject*)c->lists[b])->ob_item;
    Py_ssize_t mx = 0;
    for (Py_ssize_t j0 = 0; j0 < m; j0 += SPAN) { /* Word -> str -> length in sweeps, each level prefetched by the one before */
      const Py_ssize_t k = m - j0 < SPAN ? m - j0 : SPAN;
      for (Py_ssize_t j = 0; j < k; ++j) __builtin_prefetch((char*)items[j0 + j] + c->off_word);
      for (Py_ssize_t j = 0; j < k; ++j) {
        PyObject* o = items[j0 + j];
        if (Py_TYPE(o) != c->tp) { c->fallback = 1; return; }
        PyObject* t = *(PyObject**)((char*)o + c->off_word);
        if (!t) { c->fallback = 1; return; }
        strs[j] = t;
        __builtin_prefetch(t);
      }
      for (Py_ssize_t j = 0; j < k; ++j) {
        PyObject* t = strs[j];
        if (!PyUnicode_CheckExact(t) || !PyUnicode_IS_READY(t)) { c->fallback = 1; return; }
        if (PyUnicode_GET_LENGTH(t) > mx) mx = PyUnicode_GET_LENGTH(t);
      }
    }
    c->wmax[b] = (int32_t)(mx > 0x7fffffff ? 0x7fffffff : mx);
  }
}

static PyObject* geometry(PyObject* self, PyObject* args) {
  PyObject* sents;
  int n_threads = 1;
  if (!PyArg_ParseTuple(args, "O!|i", &PyList_Type, &sents, &n_threads)) return NULL;
  const Py_ssize_t n = PyList_GET_SIZE(sents);
  npy_intp dims[1] = {(npy_intp)n};
  PyArrayObject* a_len = (PyArrayObject*)PyArray_ZEROS(1, dims, NPY_INT32, 0);
  PyArrayObject* a_wmax = (PyArrayObject*)PyArray_ZEROS(1, dims, NPY_INT32, 0);
  if (!a_len || !a_wmax) { Py_XDECREF(a_len); Py_XDECREF(a_wmax); return NULL; }
  int32_t* pl = (int32_t*)PyArray_DATA(a_len);
  int32_t* pw = (int32_t*)PyArray_DATA(a_wmax);
  WordLayout wl;
  wl.tp = NULL;
  if (n_threads > 1 && n >= 512) { /* the plain format on the pool; anything unusual falls through to the serial pass below */
    PyObject** lists = (PyObject**)malloc(sizeof(PyObject*) * (size_t)n);
    if (!lists) { Py_DECREF(a_len); Py_DECREF(a_wmax); return PyErr_NoMemory(); }
    int ok = 1;
    PyObject* first = NULL;
    for (Py_ssize_t b = 0; ok && b < n; ++b) {
      PyObject* ws = PyObject_GetAttr(PyList_GET_ITEM(sents, b), str_words);
      if (!ws) { PyErr_Clear(); ok = 0; break; }
      if (!PyList_CheckExact(ws) || Py_REFCNT(ws) < 2) ok = 0; /* a list made for this call would not outlive the DECREF below */
      else {
        lists[b] = ws;
        pl[b] = (int32_t)PyList_GET_SIZE(ws);
        if (!first && PyList_GET_SIZE(ws) > 0) first = PyList_GET_ITEM(ws, 0);
      }
      Py_DECREF(ws); /* the sentence keeps its list; no Python code runs until the pool is done */
    }
    if (ok && first) {
      word_layout(Py_TYPE(first), &wl);
      if (wl.off[F_WORD] >= 0) {
        GeoCtx gc;
        gc.n = n; gc.lists = lists; gc.tp = Py_TYPE(first); gc.off_word = wl.off[F_WORD]; gc.wmax = pw; gc.fallback = 0;
        int nch = 4 * n_threads;
        if (nch > 64) nch = 64;
        if (nch > n / 64) nch = (int)(n / 64);
        pool_run(par_geometry, &gc, nch, n_threads);
        if (!gc.fallback) {
          free(lists);
          return Py_BuildValue("(NN)", (PyObject*)a_len, (PyObject*)a_wmax);
        }
      }
    }
    free(lists);
  }
  for (Py_ssize_t b = 0; b < n; ++b) {
    PyObject* ws = PyObject_GetAttr(PyList_GET_ITEM(sents, b), str_words);
    if (!ws || !PyList_Check(ws)) {
      if (ws) PyErr_SetString(PyExc_TypeError, "geometry: sentence.words is not a list");
      Py_XDECREF(ws); Py_DECREF(a_len); Py_DECREF(a_wmax);
      return NULL;
    }
    const Py_ssize_t m = PyList_GET_SIZE(ws);
    pl[b] = (int32_t)m;
    Py_ssize_t mx = 0;
    for (Py_ssize_t j = 0; j < m; ++j) {
      if (j + 8 < m) __builtin_prefetch(PyList_GET_ITEM(ws, j + 8));
      PyObject* t = word_get(PyList_GET_ITEM(ws, j), &wl, F_WORD);
      if (!t || !PyUnicode_Check(t) || PyUnicode_READY(t) != 0) {
        if (t) PyErr_SetString(PyExc_TypeError, "geometry: word.word is not a str");
        Py_XDECREF(t); Py_DECREF(ws); Py_DECREF(a_len); Py_DECREF(a_wmax);
        return NULL;
      }
      if (PyUnicode_GET_LENGTH(t) > mx) mx = PyUnicode_GET_LENGTH(t);
      Py_DECREF(t);
    }
    pw[b] = (int32_t)mx;
    Py_DECREF(ws);
  }
  return Py_BuildValue("(NN)", (PyObject*)a_len, (PyObject*)a_wmax);
}

/* write_back(sentences, res, tables): res C-contiguous int32 [5, B, L] = heads, labels, upos, xpos, attrs ids;
 * tables = (labels, upos_list, xpos_list, attrs_list) lists of str.  Sets word.head (int) and the four strings
 * of every word of every sentence (parser.py:432-446). */
static PyObject* write_back(PyObject* self, PyObject* args) {
  PyObject *sents, *tabs;
  PyArrayObject* res;
  if (!PyArg_ParseTuple(args, "O!O!O!", &PyList_Type, &sents, &PyArray_Type, &res, &PyTuple_Type, &tabs)) return NULL;
  if (PyArray_TYPE(res) != NPY_INT32 || PyArray_NDIM(res) != 3 || PyArray_DIM(res, 0) != 5 || !PyArray_IS_C_CONTIGUOUS(res) ||
      PyTuple_GET_SIZE(tabs) != 4) {
    PyErr_SetString(PyExc_ValueError, "write_back: res must be C-contiguous int32 [5, B, L], tables a 4-tuple");
    return NULL;
  }
  const Py_ssize_t B = PyArray_DIM(res, 1), L = PyArray_DIM(res, 2);
  if (PyList_GET_SIZE(sents) != B) {
    PyErr_SetString(PyExc_ValueError, "write_back: one result row per sentence expected");
    return NULL;
  }
  PyObject* tab[4];
  Py_ssize_t ntab[4];
  for (int t = 0; t < 4; ++t) {
    tab[t] = PyTuple_GET_ITEM(tabs, t);
    if (!PyList_Check(tab[t])) {
      PyErr_SetString(PyExc_ValueError, "write_back: the id -> string tables must be lists");
      return NULL;
    }
    ntab[t] = PyList_GET_SIZE(tab[t]);
  }
  static const int field_of[4] = {F_LABEL, F_UPOS, F_XPOS, F_ATTRS};
  const int32_t* R = (const int32_t*)PyArray_DATA(res);
  WordLayout wl;
  wl.tp = NULL;
  for (Py_ssize_t b = 0; b < B; ++b) {
    PyObject* ws = PyObject_GetAttr(PyList_GET_ITEM(sents, b), str_words);
    if (!ws) return NULL;
    if (!PyList_Check(ws) || PyList_GET_SIZE(ws) > L) {
      Py_DECREF(ws);
      PyErr_Format(PyExc_ValueError, "write_back: sentence %zd does not fit the result array", b);
      return NULL;
    }
    const Py_ssize_t n = PyList_GET_SIZE(ws);
    for (Py_ssize_t j = 0; j < n; ++j) {
      PyObject* w = PyList_GET_ITEM(ws, j);
      if (j + 8 < n) { __builtin_prefetch(PyList_GET_ITEM(ws, j + 8), 1); __builtin_prefetch((char*)PyList_GET_ITEM(ws, j + 8) + 64, 1); }
      PyObject* h = PyLong_FromLong((long)R[(0 * B + b) * L + j]); /* small ints come from CPython's cache */
      if (!h || word_set(w, &wl, F_HEAD, h) != 0) { Py_XDECREF(h); Py_DECREF(ws); return NULL; }
      Py_DECREF(h);
      for (int t = 0; t < 4; ++t) {
        const int32_t id = R[((t + 1) * B + b) * L + j];
        if (id < 0 || id >= ntab[t]) {
          Py_DECREF(ws);
          PyErr_Format(PyExc_IndexError, "write_back: id %d outside table %d", (int)id, t);
          return NULL;
        }
        if (word_set(w, &wl, field_of[t], PyList_GET_ITEM(tab[t], id)) != 0) { Py_DECREF(ws); return NULL; }
      }
    }
    Py_DECREF(ws);
  }
  Py_RETURN_NONE;
}

/* ---- Lemmatizer.process_bucketed: the two passes over every word of the document that nothing on the device overlaps ----
 * lemma_keys(sentences, max_len) -> (words list, upos list, int32 lang_id [n_keys], int32 word length [n_keys], int32 key
 * index [n_words]): the DISTINCT (word, upos, sentence.lang_id) triples of the words of at most `max_len` characters, in
 * order of first appearance (what a dict keyed by the tuple gives: equal means equal by ==), and for every such word, in
 * document order, the index of its triple.  Words longer than max_len are skipped (utils.py:109-110). */
typedef struct { Py_hash_t h; int32_t idx; } KeySlot;

static PyObject* lemma_keys(PyObject* self, PyObject* args) {
  PyObject* sents;
  Py_ssize_t max_len;
  if (!PyArg_ParseTuple(args, "O!n", &PyList_Type, &sents, &max_len)) return NULL;
  const Py_ssize_t B = PyList_GET_SIZE(sents);
  Py_ssize_t N = 0;
  for (Py_ssize_t b = 0; b < B; ++b) {
    PyObject* ws = PyObject_GetAttr(PyList_GET_ITEM(sents, b), str_words);
    if (!ws) return NULL;
    if (!PyList_Check(ws)) { Py_DECREF(ws); PyErr_SetString(PyExc_TypeError, "lemma_keys: sentence.words is not a list"); return NULL; }
    N += PyList_GET_SIZE(ws);
    Py_DECREF(ws);
  }
  if (N > 0x3fffffff) { PyErr_SetString(PyExc_ValueError, "lemma_keys: too many words"); return NULL; }
  size_t cap = 64;
  while (cap < 2 * (size_t)N) cap <<= 1;
  KeySlot* tab = (KeySlot*)malloc(sizeof(KeySlot) * cap);
  int32_t* widx = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N ? N : 1));
  int32_t* klang = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N ? N : 1));
  PyObject* kwords = PyList_New(0);
  PyObject* kupos = PyList_New(0);
  PyObject* result = NULL;
  Py_ssize_t n_words = 0, n_keys = 0;
  WordLayout wl;
  wl.tp = NULL;
  if (!tab || !widx || !klang || !kwords || !kupos) { PyErr_NoMemory(); goto done; }
  for (size_t i = 0; i < cap; ++i) tab[i].idx = -1;
  for (Py_ssize_t b = 0; b < B; ++b) {
    PyObject* s = PyList_GET_ITEM(sents, b);
    PyObject* lg = PyObject_GetAttr(s, str_lang_id);
    if (!lg) goto done;
    const long lang = PyLong_AsLong(lg);
    Py_DECREF(lg);
    if (lang == -1 && PyErr_Occurred()) goto done;
    if (lang < -0x7fffffffL || lang > 0x7fffffffL) { PyErr_SetString(PyExc_ValueError, "lemma_keys: lang_id out of range"); goto done; }
    PyObject* ws = PyObject_GetAttr(s, str_words);
    if (!ws) goto done;
    if (!PyList_Check(ws)) { Py_DECREF(ws); PyErr_SetString(PyExc_TypeError, "lemma_keys: sentence.words is not a list"); goto done; }
    const Py_ssize_t m = PyList_GET_SIZE(ws);
    if (n_words + m > N) { Py_DECREF(ws); PyErr_SetString(PyExc_ValueError, "lemma_keys: the document changed"); goto done; }
    for (Py_ssize_t j = 0; j < m; ++j) {
      if (j + 4 < m) __builtin_prefetch(PyList_GET_ITEM(ws, j + 4));
      PyObject* w = PyList_GET_ITEM(ws, j);
      PyObject* word = word_get(w, &wl, F_WORD);
      if (!word) { Py_DECREF(ws); goto done; }
      if (!PyUnicode_Check(word) || PyUnicode_READY(word) != 0) {
        Py_DECREF(word); Py_DECREF(ws);
        PyErr_SetString(PyExc_TypeError, "lemma_keys: word.word is not a str");
        goto done;
      }
      if (PyUnicode_GET_LENGTH(word) > max_len) { Py_DECREF(word); continue; }
      PyObject* upos = word_get(w, &wl, F_UPOS);
      if (!upos) { Py_DECREF(word); Py_DECREF(ws); goto done; }
      const Py_hash_t hw = PyObject_Hash(word), hu = PyObject_Hash(upos);
      if ((hw == -1 || hu == -1) && PyErr_Occurred()) { Py_DECREF(word); Py_DECREF(upos); Py_DECREF(ws); goto done; }
      const Py_hash_t h = (Py_hash_t)(((size_t)hw * 1000003u) ^ ((size_t)hu * 8191u) ^ ((size_t)lang * 2654435761u));
      size_t pos = ((size_t)h ^ ((size_t)h >> 17)) & (cap - 1);
      int32_t found = -1;
      for (;;) {
        const int32_t k = tab[pos].idx;
        if (k < 0) break;
        if (tab[pos].h == h && klang[k] == (int32_t)lang) {
          PyObject* kw = PyList_GET_ITEM(kwords, k);
          PyObject* ku = PyList_GET_ITEM(kupos, k);
          int eq = kw == word ? 1 : PyObject_RichCompareBool(kw, word, Py_EQ);
          if (eq > 0) eq = ku == upos ? 1 : PyObject_RichCompareBool(ku, upos, Py_EQ);
          if (eq < 0) { Py_DECREF(word); Py_DECREF(upos); Py_DECREF(ws); goto done; }
          if (eq) { found = k; break; }
        }
        pos = (pos + 1) & (cap - 1);
      }
      if (found < 0) {
        if (PyList_Append(kwords, word) != 0 || PyList_Append(kupos, upos) != 0) { Py_DECREF(word); Py_DECREF(upos); Py_DECREF(ws); goto done; }
        found = (int32_t)n_keys++;
        klang[found] = (int32_t)lang;
        tab[pos].h = h;
        tab[pos].idx = found;
      }
      widx[n_words++] = found;
      Py_DECREF(word);
      Py_DECREF(upos);
    }
    Py_DECREF(ws);
  }
  {
    npy_intp dk[1] = {(npy_intp)n_keys}, dw[1] = {(npy_intp)n_words};
    PyArrayObject* a_lang = (PyArrayObject*)PyArray_SimpleNew(1, dk, NPY_INT32);
    PyArrayObject* a_len = (PyArrayObject*)PyArray_SimpleNew(1, dk, NPY_INT32);
    PyArrayObject* a_idx = (PyArrayObject*)PyArray_SimpleNew(1, dw, NPY_INT32);
    if (!a_lang || !a_len || !a_idx) { Py_XDECREF(a_lang); Py_XDECREF(a_len); Py_XDECREF(a_idx); goto done; }
    memcpy(PyArray_DATA(a_lang), klang, sizeof(int32_t) * (size_t)n_keys);
    memcpy(PyArray_DATA(a_idx), widx, sizeof(int32_t) * (size_t)n_words);
    int32_t* pl = (int32_t*)PyArray_DATA(a_len);
    for (Py_ssize_t k = 0; k < n_keys; ++k) pl[k] = (int32_t)PyUnicode_GET_LENGTH(PyList_GET_ITEM(kwords, k));
    result = Py_BuildValue("(OONNN)", kwords, kupos, (PyObject*)a_lang, (PyObject*)a_len, (PyObject*)a_idx);
  }
done:
  free(tab); free(widx); free(klang);
  Py_XDECREF(kwords); Py_XDECREF(kupos);
  return result;
}

/* assign_lemmas(sentences, key index int32 [n_words], lemmas list [n_keys], max_len): word.lemma = lemmas[index] for the
 * words of at most max_len characters, in the order lemma_keys saw them; the others keep their lemma. */
static PyObject* assign_lemmas(PyObject* self, PyObject* args) {
  PyObject *sents, *lemmas;
  PyArrayObject* idx;
  Py_ssize_t max_len;
  if (!PyArg_ParseTuple(args, "O!O!O!n", &PyList_Type, &sents, &PyArray_Type, &idx, &PyList_Type, &lemmas, &max_len)) return NULL;
  if (PyArray_TYPE(idx) != NPY_INT32 || PyArray_NDIM(idx) != 1 || !PyArray_IS_C_CONTIGUOUS(idx)) {
    PyErr_SetString(PyExc_ValueError, "assign_lemmas: the key indices must be a C-contiguous int32 vector");
    return NULL;
  }
  const int32_t* pi = (const int32_t*)PyArray_DATA(idx);
  const Py_ssize_t n_idx = PyArray_DIM(idx, 0), n_keys = PyList_GET_SIZE(lemmas);
  Py_ssize_t k = 0;
  WordLayout wl;
  wl.tp = NULL;
  const Py_ssize_t B = PyList_GET_SIZE(sents);
  for (Py_ssize_t b = 0; b < B; ++b) {
    PyObject* ws = PyObject_GetAttr(PyList_GET_ITEM(sents, b), str_words);
    if (!ws) return NULL;
    if (!PyList_Check(ws)) { Py_DECREF(ws); PyErr_SetString(PyExc_TypeError, "assign_lemmas: sentence.words is not a list"); return NULL; }
    for (Py_ssize_t j = 0; j < PyList_GET_SIZE(ws); ++j) {
      if (j + 4 < PyList_GET_SIZE(ws)) __builtin_prefetch(PyList_GET_ITEM(ws, j + 4));
      PyObject* w = PyList_GET_ITEM(ws, j);
      int owned = 0;
      PyObject* word = word_peek(w, &wl, F_WORD, &owned);
      if (!word) { Py_DECREF(ws); return NULL; }
      const int is_str = PyUnicode_Check(word) && PyUnicode_READY(word) == 0;
      const Py_ssize_t len = is_str ? PyUnicode_GET_LENGTH(word) : 0;
      if (owned) Py_DECREF(word);
      if (!is_str) { Py_DECREF(ws); PyErr_SetString(PyExc_TypeError, "assign_lemmas: word.word is not a str"); return NULL; }
      if (len > max_len) continue;
      if (k >= n_idx || pi[k] < 0 || pi[k] >= n_keys) {
        Py_DECREF(ws);
        PyErr_SetString(PyExc_ValueError, "assign_lemmas: the document does not match the key indices");
        return NULL;
      }
      if (word_set(w, &wl, F_LEMMA, PyList_GET_ITEM(lemmas, pi[k])) != 0) { Py_DECREF(ws); return NULL; }
      ++k;
    }
    Py_DECREF(ws);
  }
  if (k != n_idx) { PyErr_SetString(PyExc_ValueError, "assign_lemmas: the document does not match the key indices"); return NULL; }
  Py_RETURN_NONE;
}


static PyMethodDef methods[] = {
    {"stack_rows", stack_rows, METH_VARARGS, "copy N equal-size contiguous buffers into one [N, size] buffer"},
    {"fill_ragged", fill_ragged, METH_VARARGS, "ragged int64 values -> zero-padded int64 [N, width] rows"},
    {"featurize", featurize, METH_VARARGS, "Sentence objects -> packed index arrays + padded embedding rows of one batch"},
    {"write_back_flat", write_back_flat, METH_VARARGS, "result ids [n_words, 5] -> the words, in order"},
    {"wid_table_new", wid_table_new, METH_NOARGS, "an empty C-side raw token -> word id table (capsule) for featurize"},
    {"wid_table_size", wid_table_size, METH_O, "(entries, slots) of a wid_table_new() capsule"},
    {"timing", timing, METH_NOARGS, "accumulated seconds per stage of featurize since the last call"},
    {"geometry", geometry, METH_VARARGS, "words per sentence and longest word per sentence"},
    {"max_words", max_words, METH_VARARGS, "word count of the longest sentence"},
    {"lemma_keys", lemma_keys, METH_VARARGS, "distinct (word, upos, lang_id) triples of a document + the triple of every word"},
    {"assign_lemmas", assign_lemmas, METH_VARARGS, "word.lemma = lemmas[key index] for the words lemma_keys listed"},
    {"write_back", write_back, METH_VARARGS, "result ids -> head / label / upos / xpos / attrs of every word"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_cube_host", "host side of the parser path: featurisation and write-back", -1,
                                    methods};

PyMODINIT_FUNC PyInit__cube_host(void) {
  import_array();
  for (int f = 0; f < F_COUNT; ++f) {
    field_str[f] = PyUnicode_InternFromString(FIELD_NAMES[f]);
    if (!field_str[f]) return NULL;
  }
  str_words = PyUnicode_InternFromString("words");
  str_lang_id = PyUnicode_InternFromString("lang_id");
  str_lower = PyUnicode_InternFromString("lower");
  if (!str_words || !str_lang_id || !str_lower) return NULL;
  return PyModule_Create(&module);
}

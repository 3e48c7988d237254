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

static PyMethodDef methods[] = {
    {"stack_rows", stack_rows, METH_VARARGS, "copy N equal-size contiguous buffers into one [N, size] buffer"},
    {"fill_ragged", fill_ragged, METH_VARARGS, "ragged int64 values -> zero-padded int64 [N, width] rows"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_cube_host", "host-side bulk copies of the featurisation", -1,
                                    methods};

PyMODINIT_FUNC PyInit__cube_host(void) {
  import_array();
  return PyModule_Create(&module);
}

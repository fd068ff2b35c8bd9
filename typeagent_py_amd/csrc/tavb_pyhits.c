/* _tavb_pyhits: builds the `list[list[ScoredInt]]` a batched lookup returns, in C.
 *
 * What the reference builds per query is `[ScoredInt(int(i), float(scores[i])) for i in top_indices]` (vectorbase.py:188-190);
 * a 1024-query top-32 batch is 32 768 such objects.  Built by the interpreter (dataclass __init__ through map()) they take ~110 ns
 * apiece = 3.6 ms, 13 % on top of the 27 ms the batch takes on the device.  Here every object is allocated with the type's own
 * tp_alloc and its two slots are stored directly (exactly what the generated __init__ does: `self.item = item; self.score = score`).
 *
 * Host-side convenience only: no device code, no part of the C ABI of libtavb.so (include/tavb.h); when the module is absent the
 * Python loop of vectorbase._scored_lists does the same work.
 *
 *   build(cls, ordinals, scores, counts, width) -> list[list[cls]]
 *     cls       a class with `item` and `score` member descriptors (__slots__), e.g. the slots dataclass ScoredInt
 *     ordinals  C-contiguous int64   [n, width]   (buffer protocol)
 *     scores    C-contiguous float32 [n, width]
 *     counts    C-contiguous int32   [n]          hits of each query (<= width)
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <structmember.h>
#include <stdint.h>

static int slot_offset(PyObject* cls, const char* name, Py_ssize_t* out) {
  PyObject* descr = PyObject_GetAttrString(cls, name);
  if (!descr) return -1;
  if (Py_TYPE(descr) != &PyMemberDescr_Type) {
    Py_DECREF(descr);
    PyErr_Format(PyExc_TypeError, "%s is not a slot of the class", name);
    return -1;
  }
  PyMemberDef* m = ((PyMemberDescrObject*)descr)->d_member;
  if (m->type != T_OBJECT_EX || (m->flags & READONLY)) {
    Py_DECREF(descr);
    PyErr_Format(PyExc_TypeError, "%s is not a writable object slot", name);
    return -1;
  }
  *out = m->offset;
  Py_DECREF(descr);
  return 0;
}

static PyObject* build(PyObject* self, PyObject* args) {
  PyObject* cls;
  Py_buffer ords, scs, cnts;
  Py_ssize_t width;
  if (!PyArg_ParseTuple(args, "Oy*y*y*n", &cls, &ords, &scs, &cnts, &width)) return NULL;
  PyObject* out = NULL;
  Py_ssize_t off_item, off_score;
  if (!PyType_Check(cls)) {
    PyErr_SetString(PyExc_TypeError, "cls must be a class");
    goto done;
  }
  PyTypeObject* tp = (PyTypeObject*)cls;
  if (slot_offset(cls, "item", &off_item) || slot_offset(cls, "score", &off_score)) goto done;
  if (tp->tp_dictoffset != 0 || tp->tp_itemsize != 0 || off_item + (Py_ssize_t)sizeof(PyObject*) > tp->tp_basicsize ||
      off_score + (Py_ssize_t)sizeof(PyObject*) > tp->tp_basicsize) {
    PyErr_SetString(PyExc_TypeError, "cls must be a fixed-size __slots__ class without a __dict__");
    goto done;
  }
  const Py_ssize_t n = cnts.len / (Py_ssize_t)sizeof(int32_t);
  if (width < 0 || ords.len != n * width * (Py_ssize_t)sizeof(int64_t) || scs.len != n * width * (Py_ssize_t)sizeof(float)) {
    PyErr_SetString(PyExc_ValueError, "ordinals / scores / counts do not have the shapes [n, width] / [n, width] / [n]");
    goto done;
  }
  const int64_t* o = (const int64_t*)ords.buf;
  const float* s = (const float*)scs.buf;
  const int32_t* c = (const int32_t*)cnts.buf;
  out = PyList_New(n);
  if (!out) goto done;
  for (Py_ssize_t q = 0; q < n; ++q) {
    Py_ssize_t m = c[q];
    if (m < 0) m = 0;
    if (m > width) m = width;
    PyObject* row = PyList_New(m);
    if (!row) goto fail;
    PyList_SET_ITEM(out, q, row);
    for (Py_ssize_t j = 0; j < m; ++j) {
      PyObject* hit = tp->tp_alloc(tp, 0);
      if (!hit) goto fail;
      PyList_SET_ITEM(row, j, hit);  /* (owned by the list from here on: a failure below leaves None-free, half-filled slots that dealloc handles) */
      PyObject* item = PyLong_FromLongLong((long long)o[q * width + j]);
      PyObject* score = PyFloat_FromDouble((double)s[q * width + j]);
      if (!item || !score) {
        Py_XDECREF(item);
        Py_XDECREF(score);
        goto fail;
      }
      *(PyObject**)((char*)hit + off_item) = item;
      *(PyObject**)((char*)hit + off_score) = score;
    }
  }
  goto done;
fail:
  Py_CLEAR(out);
done:
  PyBuffer_Release(&ords);
  PyBuffer_Release(&scs);
  PyBuffer_Release(&cnts);
  return out;
}

static PyMethodDef methods[] = {
    {"build", build, METH_VARARGS, "build(cls, ordinals, scores, counts, width) -> list[list[cls]]"},
    {NULL, NULL, 0, NULL},
};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_tavb_pyhits", "ScoredInt lists of a batched lookup, built in C", -1, methods};

PyMODINIT_FUNC PyInit__tavb_pyhits(void) { return PyModule_Create(&module); }

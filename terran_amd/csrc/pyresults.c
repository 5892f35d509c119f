/* _pyresults -- the reference's RETURN TYPE, built in C.
 *
 * RetinaFace.call / Detection return list[N] of list[{'bbox': ndarray (4,), 'landmarks': ndarray (5,2), 'score': float32}]
 * (retinaface/wrapper.py:228-236, face/detection/__init__.py:59-84).  A 1080p video batch carries ~11 000 detections:
 * as a Python comprehension that is ~2 ms per batch with the GIL held (three array views + a dict per detection), a third
 * of a whole pipelined step for a four-lane StreamPipeline.  Here the same objects -- views INTO the packed result arrays,
 * numpy float32 scalars, dicts with interned keys -- are created straight through the C API.  terran_amd/results.py falls
 * back to the comprehension when this module is not built (host glue only: no arithmetic happens here).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>

static PyObject *k_bbox, *k_landmarks, *k_score;

static PyObject* row_view(PyArrayObject* a, npy_intp row) {
  const int nd = PyArray_NDIM(a) - 1;
  PyArray_Descr* d = PyArray_DESCR(a);
  Py_INCREF(d);
  PyObject* v = PyArray_NewFromDescr(&PyArray_Type, d, nd, PyArray_DIMS(a) + 1, PyArray_STRIDES(a) + 1,
                                     PyArray_BYTES(a) + row * PyArray_STRIDE(a, 0), PyArray_FLAGS(a) & ~NPY_ARRAY_OWNDATA, NULL);
  if (!v) return NULL;
  Py_INCREF(a);
  if (PyArray_SetBaseObject((PyArrayObject*)v, (PyObject*)a) < 0) {
    Py_DECREF(v);
    return NULL;
  }
  return v;
}

/* detections(counts int32 (N,), boxes (T,4), landmarks (T,5,2), scores (T,)) -> list[N] of list[dict] */
static PyObject* detections(PyObject* self, PyObject* args) {
  PyArrayObject *counts, *boxes, *lmks, *scores;
  if (!PyArg_ParseTuple(args, "O!O!O!O!", &PyArray_Type, &counts, &PyArray_Type, &boxes, &PyArray_Type, &lmks, &PyArray_Type, &scores))
    return NULL;
  if (PyArray_NDIM(counts) != 1 || PyArray_TYPE(counts) != NPY_INT32 || !PyArray_ISCARRAY_RO(counts) || PyArray_NDIM(boxes) != 2 ||
      PyArray_NDIM(lmks) != 3 || PyArray_NDIM(scores) != 1) {
    PyErr_SetString(PyExc_ValueError, "detections(counts int32 (N,), boxes (T,4), landmarks (T,5,2), scores (T,))");
    return NULL;
  }
  const npy_intp n = PyArray_DIM(counts, 0), total = PyArray_DIM(scores, 0);
  const int* c = (const int*)PyArray_DATA(counts);
  npy_intp sum = 0;
  for (npy_intp i = 0; i < n; ++i) {
    if (c[i] < 0) {
      PyErr_SetString(PyExc_ValueError, "negative count");
      return NULL;
    }
    sum += c[i];
  }
  if (sum > total || PyArray_DIM(boxes, 0) < sum || PyArray_DIM(lmks, 0) < sum) {
    PyErr_SetString(PyExc_ValueError, "counts exceed the result arrays");
    return NULL;
  }
  PyObject* out = PyList_New(n);
  if (!out) return NULL;
  npy_intp o = 0;
  for (npy_intp i = 0; i < n; ++i) {
    PyObject* lst = PyList_New(c[i]);
    if (!lst) goto fail;
    PyList_SET_ITEM(out, i, lst);
    for (int k = 0; k < c[i]; ++k, ++o) {
      PyObject* d = _PyDict_NewPresized(3);
      if (!d) goto fail;
      PyList_SET_ITEM(lst, k, d);
      PyObject* b = row_view(boxes, o);
      PyObject* l = row_view(lmks, o);
      PyObject* s = PyArray_Scalar(PyArray_BYTES(scores) + o * PyArray_STRIDE(scores, 0), PyArray_DESCR(scores), (PyObject*)scores);
      const int bad = !b || !l || !s || PyDict_SetItem(d, k_bbox, b) < 0 || PyDict_SetItem(d, k_landmarks, l) < 0 ||
                      PyDict_SetItem(d, k_score, s) < 0;
      Py_XDECREF(b);
      Py_XDECREF(l);
      Py_XDECREF(s);
      if (bad) goto fail;
    }
  }
  return out;
fail:
  Py_DECREF(out);
  return NULL;
}

static PyMethodDef methods[] = {{"detections", detections, METH_VARARGS, "list[N] of list[{'bbox', 'landmarks', 'score'}] over packed arrays"},
                                {NULL, NULL, 0, NULL}};
static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_pyresults", NULL, -1, methods};

PyMODINIT_FUNC PyInit__pyresults(void) {
  import_array();
  k_bbox = PyUnicode_InternFromString("bbox");
  k_landmarks = PyUnicode_InternFromString("landmarks");
  k_score = PyUnicode_InternFromString("score");
  return PyModule_Create(&module);
}

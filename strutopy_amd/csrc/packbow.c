/* packbow.c -- CPython helper of strutopy_amd.corpus.pack_bow: the reference's corpus format, list[list[(word_id, count)]]
 * (what STM.__init__ receives, src/modules/stm.py:311, and re-reads with np.array(documents[i]) for every document in every
 * EM iteration, stm.py:522-533), walked ONCE in C into CSR arrays the caller allocated.  Host plumbing only (no arithmetic);
 * corpus.py falls back to its iterator-based path when this module has not been built.
 *
 *   lengths(documents, lens: writable int64 buffer[N]) -> nnz
 *   fill(documents, indices: writable int32 buffer[nnz], counts: writable float64 buffer[nnz]) -> max word id (-1 if empty)
 * Both raise IndexError for what the reference's own indexing would trip over (an empty document, an entry that is not a
 * (word_id, count) pair) and for word ids that are not integers in [0, 2^31). */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <math.h>
#include <stdint.h>

static PyObject *pb_lengths(PyObject *self, PyObject *args) {
    PyObject *docs;
    Py_buffer lens;
    if (!PyArg_ParseTuple(args, "Ow*", &docs, &lens)) return NULL;
    PyObject *seq = PySequence_Fast(docs, "documents must be a sequence of documents");
    if (!seq) { PyBuffer_Release(&lens); return NULL; }
    const Py_ssize_t N = PySequence_Fast_GET_SIZE(seq);
    if (lens.len != (Py_ssize_t)sizeof(int64_t) * N) {
        PyErr_SetString(PyExc_ValueError, "lens must be an int64 buffer with one entry per document");
        goto fail;
    }
    {
        int64_t *out = (int64_t *)lens.buf, nnz = 0;
        PyObject **items = PySequence_Fast_ITEMS(seq);
        for (Py_ssize_t i = 0; i < N; ++i) {
            const Py_ssize_t n = PyObject_Length(items[i]);
            if (n < 0) goto fail;
            if (n < 1) { PyErr_SetString(PyExc_IndexError, "empty document: the reference indexes doc_array[:, 0] (stm.py:523)"); goto fail; }
            out[i] = (int64_t)n;
            nnz += n;
        }
        Py_DECREF(seq);
        PyBuffer_Release(&lens);
        return PyLong_FromLongLong(nnz);
    }
fail:
    Py_DECREF(seq);
    PyBuffer_Release(&lens);
    return NULL;
}

static int as_double(PyObject *o, double *v) {
    if (PyFloat_CheckExact(o)) { *v = PyFloat_AS_DOUBLE(o); return 0; }
    if (PyLong_CheckExact(o)) { *v = PyLong_AsDouble(o); return (*v == -1.0 && PyErr_Occurred()) ? -1 : 0; }
    *v = PyFloat_AsDouble(o);     /* numpy scalars and anything with __float__ */
    return (*v == -1.0 && PyErr_Occurred()) ? -1 : 0;
}

static PyObject *pb_fill(PyObject *self, PyObject *args) {
    PyObject *docs;
    Py_buffer bi, bc;
    if (!PyArg_ParseTuple(args, "Ow*w*", &docs, &bi, &bc)) return NULL;
    PyObject *seq = PySequence_Fast(docs, "documents must be a sequence of documents");
    if (!seq) { PyBuffer_Release(&bi); PyBuffer_Release(&bc); return NULL; }
    const Py_ssize_t N = PySequence_Fast_GET_SIZE(seq);
    const int64_t cap = (int64_t)(bi.len / (Py_ssize_t)sizeof(int32_t));
    int32_t *idx = (int32_t *)bi.buf;
    double *cnt = (double *)bc.buf;
    int64_t pos = 0, vmax = -1;
    PyObject *dseq = NULL, *pair = NULL;
    if (bc.len / (Py_ssize_t)sizeof(double) != cap) { PyErr_SetString(PyExc_ValueError, "indices and counts must have the same length"); goto fail; }
    {
        PyObject **items = PySequence_Fast_ITEMS(seq);
        for (Py_ssize_t i = 0; i < N; ++i) {
            dseq = PySequence_Fast(items[i], "a document must be a sequence of (word_id, count) pairs");
            if (!dseq) goto fail;
            const Py_ssize_t n = PySequence_Fast_GET_SIZE(dseq);
            PyObject **ent = PySequence_Fast_ITEMS(dseq);
            if (pos + n > cap) { PyErr_SetString(PyExc_ValueError, "the corpus changed between the two passes"); goto fail; }
            for (Py_ssize_t j = 0; j < n; ++j) {
                PyObject *e = ent[j], *w, *c;
                if (PyTuple_CheckExact(e) && PyTuple_GET_SIZE(e) == 2) { w = PyTuple_GET_ITEM(e, 0); c = PyTuple_GET_ITEM(e, 1); pair = NULL; }
                else {
                    pair = PySequence_Fast(e, "documents must hold (word_id, count) pairs (stm.py:522-526)");
                    if (!pair) { PyErr_Clear(); PyErr_SetString(PyExc_IndexError, "documents must hold (word_id, count) pairs (stm.py:522-526)"); goto fail; }
                    if (PySequence_Fast_GET_SIZE(pair) != 2) { PyErr_SetString(PyExc_IndexError, "documents must hold (word_id, count) pairs (stm.py:522-526)"); goto fail; }
                    w = PySequence_Fast_GET_ITEM(pair, 0); c = PySequence_Fast_GET_ITEM(pair, 1);
                }
                double wv, cv;
                if (as_double(w, &wv) || as_double(c, &cv)) { PyErr_Clear(); PyErr_SetString(PyExc_IndexError, "documents must hold (word_id, count) pairs (stm.py:522-526)"); goto fail; }
                if (!(wv >= 0.0 && wv < 2147483648.0) || wv != floor(wv)) { PyErr_SetString(PyExc_IndexError, "word ids must be non-negative integers below 2^31"); goto fail; }
                idx[pos] = (int32_t)wv;
                cnt[pos] = cv;
                if ((int64_t)wv > vmax) vmax = (int64_t)wv;
                ++pos;
                Py_CLEAR(pair);
            }
            Py_CLEAR(dseq);
        }
    }
    if (pos != cap) { PyErr_SetString(PyExc_ValueError, "the corpus changed between the two passes"); goto fail; }
    Py_DECREF(seq);
    PyBuffer_Release(&bi); PyBuffer_Release(&bc);
    return PyLong_FromLongLong(vmax);
fail:
    Py_XDECREF(pair); Py_XDECREF(dseq); Py_DECREF(seq);
    PyBuffer_Release(&bi); PyBuffer_Release(&bc);
    return NULL;
}

static PyMethodDef methods[] = {
    {"lengths", pb_lengths, METH_VARARGS, "lengths(documents, lens) -> nnz"},
    {"fill", pb_fill, METH_VARARGS, "fill(documents, indices, counts) -> max word id"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_packbow", "BoW lists -> CSR, one pass in C", -1, methods};
PyMODINIT_FUNC PyInit__packbow(void) { return PyModule_Create(&moddef); }

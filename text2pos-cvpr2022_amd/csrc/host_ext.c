/* _t2p_host: CPython helper of the drop-in entry point CellRetrievalNetwork.encode_objects (no GPU work).
 *
 * The reference takes every object's centre and mean colour as a float64 NumPy mean over the object's RAW points, one
 * Python call per object and per encode_objects call (`obj.get_center()` / `obj.get_color_rgb()`,
 * datapreparation/kitti360pose/imports.py:28-41, called from models/object_encoder.py:121-131): ~20 us per object, 20 x the
 * GPU time of the cell.  column_sums() walks a whole call's objects in C instead: one attribute look-up + one buffer
 * request per object, then the column sums (and the sums of absolute values, the caller's rounding bound) of all [m, 3]
 * float64 arrays with the GIL released, on a few threads.  The caller (data.py::object_means_many) turns sums into means
 * and proves each float32 result equal to the reference's bit for bit (or recomputes that row with np.mean itself), so the
 * summation order here is free: four interleaved partial sums per column.
 *
 *   column_sums(cells, attr, sums, abs_sums, rows, threads) -> n_objects
 *     cells      list of lists of objects (a call's `objects` argument, or the part of it that is not cached)
 *     attr       "xyz" or "rgb": the attribute holding a C-contiguous float64 [m, 3] array with m >= 1
 *     sums, abs_sums   writable float64 buffers of >= 3 n_objects items;  rows: writable int64 buffer of >= n_objects items
 *     threads    worker threads for the summation (<= 16)
 *   returns the number of objects, or -(i + 1) if flat object i does not hold such an array (nothing is written then:
 *   the caller falls back to the accessors of the objects).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef struct {
    const double* p;
    Py_ssize_t rows;
} arr_t;

typedef struct {
    const arr_t* a;
    Py_ssize_t lo, hi;
    double* sums;
    double* asums;
} job_t;

/* Column sums and sums of absolute values of one [m, 3] array.  The array is read as a flat stream of doubles with TWELVE
 * independent accumulators each (element i goes to accumulator i % 12: four interleaved partial sums per column), a form the
 * compiler turns into three 4-wide vector adds per 12 elements without re-associating anything; cloned for AVX2 with run-time
 * dispatch (the library is built in one container and runs in another). */
__attribute__((target_clones("avx2", "default")))
static void sum_one(const double* p, Py_ssize_t m, double* s, double* t) {
    double a[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, b[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const Py_ssize_t n = 3 * m;
    Py_ssize_t i = 0;
    for (; i + 12 <= n; i += 12)
        for (int j = 0; j < 12; j++) {
            const double v = p[i + j];
            a[j] += v;
            b[j] += fabs(v);
        }
    for (int j = 0; i < n; i++, j++) {
        a[j] += p[i];
        b[j] += fabs(p[i]);
    }
    for (int c = 0; c < 3; c++) {
        s[c] = (a[c] + a[3 + c]) + (a[6 + c] + a[9 + c]);
        t[c] = (b[c] + b[3 + c]) + (b[6 + c] + b[9 + c]);
    }
}

static void* run_job(void* arg) {
    job_t* j = (job_t*)arg;
    for (Py_ssize_t i = j->lo; i < j->hi; i++) sum_one(j->a[i].p, j->a[i].rows, j->sums + 3 * i, j->asums + 3 * i);
    return NULL;
}

static PyObject* column_sums(PyObject* self, PyObject* args) {
    PyObject *cells, *attr;
    Py_buffer sums, asums, rows;
    int threads = 1;
    if (!PyArg_ParseTuple(args, "O!Uw*w*w*i", &PyList_Type, &cells, &attr, &sums, &asums, &rows, &threads)) return NULL;
    PyObject* result = NULL;
    Py_buffer* views = NULL;
    arr_t* arrs = NULL;
    Py_ssize_t n = 0, got = 0;
    const Py_ssize_t n_cells = PyList_GET_SIZE(cells);
    for (Py_ssize_t c = 0; c < n_cells; c++) {
        PyObject* objs = PyList_GET_ITEM(cells, c);
        if (!PyList_Check(objs)) {
            PyErr_SetString(PyExc_TypeError, "column_sums: every cell must be a list of objects");
            goto done;
        }
        n += PyList_GET_SIZE(objs);
    }
    if (sums.len < (Py_ssize_t)(3 * n * sizeof(double)) || asums.len < (Py_ssize_t)(3 * n * sizeof(double)) ||
        rows.len < (Py_ssize_t)(n * sizeof(int64_t)) || sums.itemsize != 8 || asums.itemsize != 8 || rows.itemsize != 8) {
        PyErr_SetString(PyExc_ValueError, "column_sums: output buffers are too small (float64 [n, 3] x 2, int64 [n])");
        goto done;
    }
    views = (Py_buffer*)calloc((size_t)(n > 0 ? n : 1), sizeof(Py_buffer));
    arrs = (arr_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(arr_t));
    if (views == NULL || arrs == NULL) {
        PyErr_NoMemory();
        goto done;
    }
    Py_ssize_t bad = -1, total_rows = 0;
    for (Py_ssize_t c = 0; c < n_cells && bad < 0; c++) {
        PyObject* objs = PyList_GET_ITEM(cells, c);
        const Py_ssize_t k = PyList_GET_SIZE(objs);
        for (Py_ssize_t j = 0; j < k; j++) {
            PyObject* a = PyObject_GetAttr(PyList_GET_ITEM(objs, j), attr);
            if (a == NULL) {
                PyErr_Clear();
                bad = got;
                break;
            }
            const int rc = PyObject_GetBuffer(a, &views[got], PyBUF_FORMAT | PyBUF_STRIDES);
            Py_DECREF(a);   /* the view holds its own reference to the exporter */
            if (rc != 0) {
                PyErr_Clear();
                bad = got;
                break;
            }
            Py_buffer* v = &views[got];
            got++;          /* (released below whatever the checks say) */
            const char* f = v->format ? v->format : "B";
            if (f[0] == '<' || f[0] == '=' || f[0] == '@') f++;
            if (strcmp(f, "d") != 0 || v->ndim != 2 || v->shape[1] != 3 || v->shape[0] < 1 || v->itemsize != 8 ||
                v->strides[1] != 8 || v->strides[0] != 24) {
                bad = got - 1;
                break;
            }
            arrs[got - 1].p = (const double*)v->buf;
            arrs[got - 1].rows = v->shape[0];
            total_rows += v->shape[0];
        }
    }
    if (bad >= 0) {
        result = PyLong_FromSsize_t(-(bad + 1));
        goto done;
    }
    {
        int64_t* r = (int64_t*)rows.buf;
        for (Py_ssize_t i = 0; i < n; i++) r[i] = (int64_t)arrs[i].rows;
        if (threads > 16) threads = 16;
        if (threads < 1) threads = 1;
        if ((Py_ssize_t)threads > total_rows / 150000 + 1) threads = (int)(total_rows / 150000 + 1);   /* a thread start costs as much as ~100 k rows */
        job_t jobs[16];
        Py_ssize_t lo = 0, acc = 0;
        int nj = 0;
        for (Py_ssize_t i = 0; i < n; i++) {     /* contiguous object ranges with ~equal row counts */
            acc += arrs[i].rows;
            if (nj + 1 < threads && acc * (Py_ssize_t)threads >= total_rows * (Py_ssize_t)(nj + 1)) {
                jobs[nj].a = arrs; jobs[nj].lo = lo; jobs[nj].hi = i + 1;
                jobs[nj].sums = (double*)sums.buf; jobs[nj].asums = (double*)asums.buf;
                nj++;
                lo = i + 1;
            }
        }
        jobs[nj].a = arrs; jobs[nj].lo = lo; jobs[nj].hi = n;
        jobs[nj].sums = (double*)sums.buf; jobs[nj].asums = (double*)asums.buf;
        nj++;
        Py_BEGIN_ALLOW_THREADS
        pthread_t tid[16];
        int started[16];
        for (int t = 1; t < nj; t++) started[t] = pthread_create(&tid[t], NULL, run_job, &jobs[t]) == 0;
        run_job(&jobs[0]);
        for (int t = 1; t < nj; t++) {
            if (started[t]) pthread_join(tid[t], NULL);
            else run_job(&jobs[t]);
        }
        Py_END_ALLOW_THREADS
        result = PyLong_FromSsize_t(n);
    }
done:
    if (views != NULL) {
        for (Py_ssize_t i = 0; i < got; i++) PyBuffer_Release(&views[i]);
        free(views);
    }
    free(arrs);
    PyBuffer_Release(&sums);
    PyBuffer_Release(&asums);
    PyBuffer_Release(&rows);
    return result;
}

static PyMethodDef methods[] = {
    {"column_sums", column_sums, METH_VARARGS,
     "column_sums(cells, attr, sums, abs_sums, rows, threads) -> n_objects (or -(i + 1): flat object i has no float64 [m, 3] array)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_t2p_host", "host-side helpers of text2pos_amd (no GPU work)", -1, methods};

PyMODINIT_FUNC PyInit__t2p_host(void) { return PyModule_Create(&module); }

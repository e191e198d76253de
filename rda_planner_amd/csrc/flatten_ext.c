/* _flatten: the per-tick walk over the caller's obstacle OBJECTS in C (CPython API + buffer protocol, no numpy headers).
 *
 * The reference hands MPC.control a Python list of obstacle objects every tick (`.cone_type`, `.vertex` 2xk | `.center`, `.radius`,
 * `.velocity` - what MPC.convert_rda_obstacle reads, mpc.py:192-203).  Turning 200 of them into the flat arrays of rda_upload_scene
 * costs ~150 us in numpy (attribute access and small-array handling per object) - as much as the device needs for the rest of
 * the tick.  This module does the same walk in ~10 us.  It is an ACCELERATOR of host glue: RDA_solver.flatten_scene falls back to
 * its numpy implementation whenever this module is missing or declines an input (returns -1), and a test pins the two against
 * each other.  It declines anything it does not handle exactly like the numpy code: other cone types, vertices that are not a
 * float64 array of shape (2, k <= E), non-finite velocities, circles with E < 3.
 *
 *     flatten(objs, E, kind, nvert, geom, vel) -> 0 | -1
 *         objs  list / tuple of n obstacle objects
 *         kind  int32 [n]        out: 0 polygon, 1 circle
 *         nvert int32 [n]        out: vertices of a polygon (0 for a circle)
 *         geom  float64 [n][E][2] out (zero-filled by the caller): vertices | centre, (radius, 0)
 *         vel   float64 [n][2]   out
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <math.h>
#include <string.h>

static PyObject *s_cone_type, *s_vertex, *s_center, *s_radius, *s_velocity;

/* first `want` float64 entries of a buffer object in C order; returns the number of entries it has (0 on failure / other dtype) */
static Py_ssize_t first_doubles(PyObject *o, double *out, int want)
{
    Py_buffer b;
    if (PyObject_GetBuffer(o, &b, PyBUF_STRIDES | PyBUF_FORMAT) != 0) { PyErr_Clear(); return 0; }
    Py_ssize_t n = 0;
    if (b.format && strcmp(b.format, "d") == 0 && b.itemsize == 8) {
        n = 1;
        for (int d = 0; d < b.ndim; ++d) n *= b.shape[d];
        Py_ssize_t idx[8] = {0};
        if (b.ndim <= 8) {
            for (Py_ssize_t k = 0; k < n && k < want; ++k) {
                const char *p = (const char *)b.buf;
                for (int d = 0; d < b.ndim; ++d) p += idx[d] * b.strides[d];
                memcpy(&out[k], p, 8);
                for (int d = b.ndim - 1; d >= 0; --d) { if (++idx[d] < b.shape[d]) break; idx[d] = 0; }
            }
        } else n = 0;
    }
    PyBuffer_Release(&b);
    return n;
}

static PyObject *flatten(PyObject *self, PyObject *args)
{
    PyObject *objs, *okind, *onvert, *ogeom, *ovel;
    int E;
    if (!PyArg_ParseTuple(args, "OiOOOO", &objs, &E, &okind, &onvert, &ogeom, &ovel)) return NULL;
    PyObject *seq = PySequence_Fast(objs, "objs must be a sequence");
    if (!seq) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    Py_buffer bk, bn, bg, bv;
    int have = 0, rc = -1;
    if (PyObject_GetBuffer(okind, &bk, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) == 0) have |= 1;
    if (have == 1 && PyObject_GetBuffer(onvert, &bn, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) == 0) have |= 2;
    if (have == 3 && PyObject_GetBuffer(ogeom, &bg, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) == 0) have |= 4;
    if (have == 7 && PyObject_GetBuffer(ovel, &bv, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) == 0) have |= 8;
    if (have != 15) { PyErr_Clear(); goto done; }
    if (E < 1 || bk.len < n * 4 || bn.len < n * 4 || bg.len < n * (Py_ssize_t)E * 16 || bv.len < n * 16) goto done;
    {
        int *kind = (int *)bk.buf, *nvert = (int *)bn.buf;
        double *geom = (double *)bg.buf, *vel = (double *)bv.buf;
        for (Py_ssize_t i = 0; i < n; ++i) {
            PyObject *o = PySequence_Fast_GET_ITEM(seq, i);
            PyObject *ct = PyObject_GetAttr(o, s_cone_type);
            if (!ct) { PyErr_Clear(); goto done; }
            int is_poly = 0, is_circ = 0;
            if (PyUnicode_Check(ct)) {
                is_poly = PyUnicode_CompareWithASCIIString(ct, "Rpositive") == 0;
                is_circ = !is_poly && PyUnicode_CompareWithASCIIString(ct, "norm2") == 0;
            }
            Py_DECREF(ct);
            if (!is_poly && !is_circ) goto done;                        /* other cone types are skipped by the reference: numpy path */
            double *g = geom + (size_t)i * E * 2;
            if (is_poly) {
                PyObject *v = PyObject_GetAttr(o, s_vertex);
                if (!v) { PyErr_Clear(); goto done; }
                Py_buffer b;
                int ok = 0;
                if (PyObject_GetBuffer(v, &b, PyBUF_STRIDES | PyBUF_FORMAT) == 0) {
                    if (b.ndim == 2 && b.format && strcmp(b.format, "d") == 0 && b.itemsize == 8 && b.shape[0] == 2 && b.shape[1] <= E) {
                        const Py_ssize_t k = b.shape[1];
                        for (Py_ssize_t j = 0; j < k; ++j) {
                            memcpy(&g[2 * j], (const char *)b.buf + j * b.strides[1], 8);
                            memcpy(&g[2 * j + 1], (const char *)b.buf + b.strides[0] + j * b.strides[1], 8);
                        }
                        kind[i] = 0; nvert[i] = (int)k; ok = 1;
                    }
                    PyBuffer_Release(&b);
                } else PyErr_Clear();
                Py_DECREF(v);
                if (!ok) goto done;
            } else {
                if (E < 3) goto done;
                PyObject *c = PyObject_GetAttr(o, s_center), *r = c ? PyObject_GetAttr(o, s_radius) : NULL;
                double cxy[2], rad = 0;
                int ok = c && r && first_doubles(c, cxy, 2) >= 2;
                if (ok) { rad = PyFloat_AsDouble(r); if (rad == -1.0 && PyErr_Occurred()) { PyErr_Clear(); ok = 0; } }
                Py_XDECREF(c); Py_XDECREF(r);
                if (!ok) { PyErr_Clear(); goto done; }
                g[0] = cxy[0]; g[1] = cxy[1]; g[2] = rad;
                kind[i] = 1; nvert[i] = 0;
            }
            /* velocity: 2x1 column (what the reference's obstacles carry), any float64 array with >= 2 entries, one entry or a
             * Python / numpy scalar (the lidar examples pass 0: both coordinates alike) */
            PyObject *ve = PyObject_GetAttr(o, s_velocity);
            if (!ve) { PyErr_Clear(); goto done; }
            double w[2];
            Py_ssize_t m = first_doubles(ve, w, 2);
            if (m == 1) w[1] = w[0];
            else if (m == 0) {
                double sc = PyFloat_AsDouble(ve);
                if (sc == -1.0 && PyErr_Occurred()) { PyErr_Clear(); Py_DECREF(ve); goto done; }
                w[0] = w[1] = sc;
            }
            Py_DECREF(ve);
            if (!isfinite(w[0]) || !isfinite(w[1])) goto done;
            vel[2 * i] = w[0]; vel[2 * i + 1] = w[1];
        }
        rc = 0;
    }
done:
    if (have & 1) PyBuffer_Release(&bk);
    if (have & 2) PyBuffer_Release(&bn);
    if (have & 4) PyBuffer_Release(&bg);
    if (have & 8) PyBuffer_Release(&bv);
    Py_DECREF(seq);
    return PyLong_FromLong(rc);
}

static PyMethodDef methods[] = { {"flatten", flatten, METH_VARARGS, "obstacle objects -> flat scene arrays (0 ok, -1 declined)"}, {NULL, NULL, 0, NULL} };
static struct PyModuleDef moddef = { PyModuleDef_HEAD_INIT, "_flatten", NULL, -1, methods };

PyMODINIT_FUNC PyInit__flatten(void)
{
    s_cone_type = PyUnicode_InternFromString("cone_type"); s_vertex = PyUnicode_InternFromString("vertex");
    s_center = PyUnicode_InternFromString("center"); s_radius = PyUnicode_InternFromString("radius");
    s_velocity = PyUnicode_InternFromString("velocity");
    return PyModule_Create(&moddef);
}

/* _fast.c -- the two hot exits of ProcessingElement.render() and Snippet.__del__ in C (CPython 3.10 API).
 *
 * Small-block streaming (BASELINE config 1: 1024-frame blocks; the reference's own block loops, renderer.py:297-327,
 * audio_renderer.py:171-179) is served from resident windows: a pull that continues a stream returns a row view of the
 * window and launches nothing.  What a block then costs is the interpreter: a Python-level render() (two dictionary
 * look-ups, a few comparisons), Snippet.window_rows (an object and two tuples) and the Snippet's __del__ were 0.8 us per
 * block, 21x the CPU oracle on C1 for three rounds.  This module is those three pieces, same semantics, as method
 * descriptors installed on the Python classes; anything that is not the hot exit is handed to the Python functions they
 * replace (processing_element._render_slow, snippet._del_slow).  Host glue only: nothing numeric happens here, and
 * without the module (not built) the Python versions run.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <structmember.h>

static PyObject *g_snippet_type, *g_window_type, *g_slow_render, *g_slow_del, *g_diag_dict;
static PyObject *s_la_win, *s_ra_win, *s_la_last, *s_ra_last, *s_active, *s_shape;
static Py_ssize_t sn_start, sn_host, sn_dev, sn_shape, sn_ready, sn_copy, sn_base, sn_bank;
static Py_ssize_t w_first, w_end, w_buf, w_served, w_block;

#define SLOT(obj, off) (*(PyObject **)((char *)(obj) + (off)))

static Py_ssize_t slot_offset(PyObject *type, const char *name) {
    PyObject *d = PyObject_GetAttrString(type, name);
    if (!d) return -1;
    if (Py_TYPE(d) != &PyMemberDescr_Type) {
        Py_DECREF(d);
        PyErr_Format(PyExc_TypeError, "%s is not a __slots__ member", name);
        return -1;
    }
    const Py_ssize_t off = ((PyMemberDescrObject *)d)->d_member->offset;
    Py_DECREF(d);
    return off;
}

static void set_slot(PyObject *o, Py_ssize_t off, PyObject *v) {   /* steals nothing: takes its own reference */
    Py_INCREF(v);
    SLOT(o, off) = v;
}

/* Snippet.window_rows(start, window, first_row, rows) */
static PyObject *make_row(PyObject *start, PyObject *window, long long first_row, PyObject *rows) {
    PyObject *shape_w = PyObject_GetAttr(window, s_shape);
    if (!shape_w) return NULL;
    if (!PyTuple_Check(shape_w) || PyTuple_GET_SIZE(shape_w) < 2) {
        Py_DECREF(shape_w);
        PyErr_SetString(PyExc_ValueError, "window must be a (frames, channels) buffer");
        return NULL;
    }
    PyObject *shape = PyTuple_Pack(2, rows, PyTuple_GET_ITEM(shape_w, 1));
    Py_DECREF(shape_w);
    if (!shape) return NULL;
    PyObject *row0 = PyLong_FromLongLong(first_row);
    PyObject *base = row0 ? PyTuple_Pack(2, window, row0) : NULL;
    Py_XDECREF(row0);
    if (!base) {
        Py_DECREF(shape);
        return NULL;
    }
    PyTypeObject *tp = (PyTypeObject *)g_snippet_type;
    PyObject *o = tp->tp_alloc(tp, 0);
    if (!o) {
        Py_DECREF(shape);
        Py_DECREF(base);
        return NULL;
    }
    set_slot(o, sn_start, start);
    set_slot(o, sn_ready, Py_None);
    set_slot(o, sn_copy, Py_None);
    set_slot(o, sn_host, Py_None);
    set_slot(o, sn_dev, Py_None);
    SLOT(o, sn_base) = base;                 /* (references made above) */
    SLOT(o, sn_shape) = shape;
    set_slot(o, sn_bank, Py_False);
    return o;
}

static PyObject *call_slow(PyObject *self, PyObject *const *args, Py_ssize_t nargs, PyObject *kwnames) {
    const Py_ssize_t nkw = kwnames ? PyTuple_GET_SIZE(kwnames) : 0;
    PyObject *stack[8];
    if (nargs + nkw + 1 > 8) {
        PyErr_SetString(PyExc_TypeError, "render() takes (start, duration)");
        return NULL;
    }
    stack[0] = self;
    for (Py_ssize_t i = 0; i < nargs + nkw; ++i) stack[1 + i] = args[i];
    return PyObject_Vectorcall(g_slow_render, stack, (size_t)(nargs + 1), kwnames);
}

static int as_ll(PyObject *v, long long *out) {
    int overflow = 0;
    *out = PyLong_AsLongLongAndOverflow(v, &overflow);
    return !overflow && !(*out == -1 && PyErr_Occurred());
}

/* ProcessingElement.render(self, start, duration) */
static PyObject *fast_render(PyObject *self, PyObject *const *args, Py_ssize_t nargs, PyObject *kwnames) {
    if (nargs == 2 && !kwnames && PyLong_CheckExact(args[0]) && PyLong_CheckExact(args[1])) {
        long long s, d;
        PyObject *active = PyDict_GetItem(g_diag_dict, s_active);
        if (as_ll(args[0], &s) && as_ll(args[1], &d) && d > 0 && active && PyLong_CheckExact(active) &&
            Py_SIZE(active) == 0) {
            PyObject **dictptr = _PyObject_GetDictPtr(self);
            PyObject *dict = dictptr ? *dictptr : NULL;
            if (dict) {
                PyObject *win = PyDict_GetItem(dict, s_la_win);
                if (win) {
                    /* the next block of a stream served from a look-ahead window (look_ahead.py) */
                    if ((PyObject *)Py_TYPE(win) == g_window_type) {
                        long long served, end, first, blk;
                        PyObject *buf = SLOT(win, w_buf);
                        if (buf && SLOT(win, w_served) && SLOT(win, w_end) && SLOT(win, w_first) && SLOT(win, w_block) &&
                            as_ll(SLOT(win, w_served), &served) && as_ll(SLOT(win, w_end), &end) &&
                            as_ll(SLOT(win, w_first), &first) && as_ll(SLOT(win, w_block), &blk) && s == served &&
                            s + d <= end && (blk == 0 || d == blk)) {
                            PyObject *now = PyLong_FromLongLong(s + d);
                            if (!now) return NULL;
                            PyObject *old = SLOT(win, w_served);
                            Py_INCREF(now);
                            SLOT(win, w_served) = now;
                            Py_DECREF(old);
                            const int rc = PyDict_SetItem(dict, s_la_last, now);
                            Py_DECREF(now);
                            if (rc < 0) return NULL;
                            return make_row(args[0], buf, s - first, args[1]);
                        }
                    }
                    PyErr_Clear();
                } else {
                    /* ... or from a read-ahead window of a pure sub-graph: (first, end, buffer) (read_ahead.py) */
                    win = PyDict_GetItem(dict, s_ra_win);
                    if (win && PyTuple_CheckExact(win) && (PyTuple_GET_SIZE(win) == 3 || PyTuple_GET_SIZE(win) == 4)) {
                        /* (first, end, buffer[, period]): a window cut for one block length -- IdentityPE beyond
                         * 2^24 -- serves only the blocks of its grid */
                        long long first, end, period = 0;
                        const int on_grid = PyTuple_GET_SIZE(win) == 3 ||
                                            (as_ll(PyTuple_GET_ITEM(win, 3), &period) && period > 0 && d == period);
                        if (on_grid && as_ll(PyTuple_GET_ITEM(win, 0), &first) && as_ll(PyTuple_GET_ITEM(win, 1), &end) &&
                            first <= s && s + d <= end && (period == 0 || (s - first) % period == 0)) {
                            PyObject *now = PyLong_FromLongLong(s + d);
                            if (!now) return NULL;
                            const int rc = PyDict_SetItem(dict, s_ra_last, now);
                            Py_DECREF(now);
                            if (rc < 0) return NULL;
                            return make_row(args[0], PyTuple_GET_ITEM(win, 2), s - first, args[1]);
                        }
                        PyErr_Clear();
                    }
                }
            }
        }
        if (PyErr_Occurred()) PyErr_Clear();
    }
    return call_slow(self, args, nargs, kwnames);
}

/* Snippet.__del__(self): nothing pending (no `ready` hook, no copy in flight) is the case of every row handed out */
static PyObject *fast_del(PyObject *self, PyObject *ignored) {
    PyObject *ready = SLOT(self, sn_ready), *copy = SLOT(self, sn_copy);
    if ((ready == NULL || ready == Py_None) && (copy == NULL || copy == Py_None)) Py_RETURN_NONE;
    return PyObject_CallOneArg(g_slow_del, self);
}

static PyObject *py_window_rows(PyObject *module, PyObject *const *args, Py_ssize_t nargs) {
    long long first;
    if (nargs != 4 || !as_ll(args[2], &first)) {
        if (!PyErr_Occurred()) PyErr_SetString(PyExc_TypeError, "window_rows(start, window, first_row, rows)");
        return NULL;
    }
    return make_row(args[0], args[1], first, args[3]);
}

static PyMethodDef render_def = {"render", (PyCFunction)(void (*)(void))fast_render, METH_FASTCALL | METH_KEYWORDS,
                                 "render(start, duration) -> Snippet (hot exits in C, else the Python implementation)"};
static PyMethodDef del_def = {"__del__", (PyCFunction)fast_del, METH_NOARGS, "Snippet finalizer (fast exit in C)"};

/* install(ProcessingElement, Snippet, look_ahead._Window, slow_render, slow_del, vars(diagnostics)) */
static PyObject *py_install(PyObject *module, PyObject *args) {
    PyObject *pe_type, *snippet_type, *window_type, *slow_render, *slow_del, *diag_dict;
    if (!PyArg_ParseTuple(args, "OOOOOO", &pe_type, &snippet_type, &window_type, &slow_render, &slow_del, &diag_dict))
        return NULL;
    if (!PyType_Check(pe_type) || !PyType_Check(snippet_type) || !PyType_Check(window_type) || !PyDict_Check(diag_dict)) {
        PyErr_SetString(PyExc_TypeError, "install(PE type, Snippet type, _Window type, slow render, slow del, dict)");
        return NULL;
    }
#define OFF(var, type, name) if ((var = slot_offset(type, name)) < 0) return NULL
    OFF(sn_start, snippet_type, "_start"); OFF(sn_host, snippet_type, "_host"); OFF(sn_dev, snippet_type, "_dev");
    OFF(sn_shape, snippet_type, "_shape"); OFF(sn_ready, snippet_type, "_ready"); OFF(sn_copy, snippet_type, "_copy");
    OFF(sn_base, snippet_type, "_base"); OFF(sn_bank, snippet_type, "_bank_window");
    OFF(w_first, window_type, "first"); OFF(w_end, window_type, "end"); OFF(w_buf, window_type, "buf");
    OFF(w_served, window_type, "served"); OFF(w_block, window_type, "block");
#undef OFF
    Py_INCREF(snippet_type); Py_INCREF(window_type); Py_INCREF(slow_render); Py_INCREF(slow_del); Py_INCREF(diag_dict);
    g_snippet_type = snippet_type; g_window_type = window_type; g_slow_render = slow_render; g_slow_del = slow_del;
    g_diag_dict = diag_dict;
    PyObject *d = PyDescr_NewMethod((PyTypeObject *)pe_type, &render_def);
    if (!d || PyObject_SetAttrString(pe_type, "render", d) < 0) {
        Py_XDECREF(d);
        return NULL;
    }
    Py_DECREF(d);
    d = PyDescr_NewMethod((PyTypeObject *)snippet_type, &del_def);
    if (!d || PyObject_SetAttrString(snippet_type, "__del__", d) < 0) {
        Py_XDECREF(d);
        return NULL;
    }
    Py_DECREF(d);
    Py_RETURN_TRUE;
}

static PyMethodDef module_methods[] = {
    {"install", py_install, METH_VARARGS, "install the C render() / __del__ on the given classes"},
    {"window_rows", (PyCFunction)(void (*)(void))py_window_rows, METH_FASTCALL,
     "window_rows(start, window, first_row, rows) -> Snippet (as Snippet.window_rows)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_fast", "hot host paths of pygmu2_amd in C", -1,
                                       module_methods};

PyMODINIT_FUNC PyInit__fast(void) {
    s_la_win = PyUnicode_InternFromString("_la_win");
    s_ra_win = PyUnicode_InternFromString("_ra_win");
    s_la_last = PyUnicode_InternFromString("_la_last");
    s_ra_last = PyUnicode_InternFromString("_ra_last");
    s_active = PyUnicode_InternFromString("_ACTIVE");
    s_shape = PyUnicode_InternFromString("shape");
    if (!s_la_win || !s_ra_win || !s_la_last || !s_ra_last || !s_active || !s_shape) return NULL;
    return PyModule_Create(&moduledef);
}

/* fastcall.c — a CPython shim for the one call that sits on the per-step path of the Python classes.
 *
 * At 512 envs the loop of experiments/main.py:212-227 (`step(a); reset(done)` from Python) is bound by host time per
 * call, and a ctypes foreign call with ten arguments costs ~1.3 us of it; the same call through the CPython vectorcall
 * convention costs ~0.3 us.  Nothing else lives here: the function pointer of wurm_single_step_slot
 * (include/wurm_hip.h) is handed over by wurm_amd/_lib.py after it has loaded libwurm_hip.so, arguments are plain
 * integers (addresses and counters).  Built by wurm_amd/csrc/Makefile with the host C compiler.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

typedef int (*step_slot_fn)(void *call_block, const void *slabs, int64_t slot, void *actions, int actions_dtype,
                            uint64_t call, int apply_pending, uint64_t pre_call, int want_obs_after, void *stream);

static step_slot_fn g_step_slot = NULL;

static PyObject *fast_bind(PyObject *self, PyObject *arg)
{
    (void)self;
    void *p = PyLong_AsVoidPtr(arg);
    if (p == NULL && PyErr_Occurred()) return NULL;
    g_step_slot = (step_slot_fn)p;
    Py_RETURN_NONE;
}

/* step_slot(call_block_addr, slabs_addr, slot, actions_ptr, actions_dtype, call, apply_pending, pre_call,
 *           want_obs_after, stream) -> int return code of wurm_single_step_slot */
static PyObject *fast_step_slot(PyObject *self, PyObject *const *args, Py_ssize_t nargs)
{
    (void)self;
    if (nargs != 10) {
        PyErr_SetString(PyExc_TypeError, "step_slot takes exactly 10 arguments");
        return NULL;
    }
    if (g_step_slot == NULL) {
        PyErr_SetString(PyExc_RuntimeError, "wurm_amd._fastcall: bind() has not been called");
        return NULL;
    }
    void *blk = PyLong_AsVoidPtr(args[0]);
    void *slabs = PyLong_AsVoidPtr(args[1]);
    long long slot = PyLong_AsLongLong(args[2]);
    void *actions = PyLong_AsVoidPtr(args[3]);
    long dtype = PyLong_AsLong(args[4]);
    unsigned long long call = PyLong_AsUnsignedLongLong(args[5]);
    int pending = PyObject_IsTrue(args[6]);
    unsigned long long pre_call = PyLong_AsUnsignedLongLong(args[7]);
    int want = PyObject_IsTrue(args[8]);
    void *stream = PyLong_AsVoidPtr(args[9]);
    if (PyErr_Occurred()) return NULL;
    int rc = g_step_slot(blk, slabs, (int64_t)slot, actions, (int)dtype, (uint64_t)call, pending, (uint64_t)pre_call,
                         want, stream);
    return PyLong_FromLong(rc);
}

static PyMethodDef fast_methods[] = {
    {"bind", (PyCFunction)fast_bind, METH_O, "bind(address of wurm_single_step_slot)"},
    {"step_slot", (PyCFunction)(void (*)(void))fast_step_slot, METH_FASTCALL, "see fastcall.c"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef fast_module = {PyModuleDef_HEAD_INIT, "_fastcall", "per-step call shim (see fastcall.c)", -1,
                                         fast_methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__fastcall(void) { return PyModule_Create(&fast_module); }

/* fastcall.c — a CPython shim for the one call that sits on the per-step path of the Python classes
 * (SingleSnake / SimpleGridworld: step_slot; MultiSnake: multi_step -> wurm_multi_step_packed).
 *
 * At 512 envs the loop of experiments/main.py:212-227 (`step(a); reset(done)` from Python) is bound by host time per
 * call, and a ctypes foreign call with ten arguments costs ~1.3 us of it; the same call through the CPython vectorcall
 * convention costs ~0.1 us.  Nothing else lives here: the address of wurm_single_step_slot / wurm_grid_step_slot
 * (include/wurm_hip.h) is the first argument (wurm_amd/_lib.py takes it from libwurm_hip.so after loading it), the
 * others are plain integers (addresses and counters).  Built by wurm_amd/csrc/Makefile with the host C compiler.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

typedef int (*step_slot_fn)(void *call_block, const void *slabs, int64_t slot, void *actions, int actions_dtype,
                            uint64_t call, int apply_pending, uint64_t pre_call, int want_obs_after, void *stream);

/* step_slot(function_addr, call_block_addr, slabs_addr, slot, actions_ptr, actions_dtype, call, apply_pending,
 *           pre_call, want_obs_after, stream) -> int return code of the wurm_*_step_slot entry point at function_addr */
static PyObject *fast_step_slot(PyObject *self, PyObject *const *args, Py_ssize_t nargs)
{
    (void)self;
    if (nargs != 11) {
        PyErr_SetString(PyExc_TypeError, "step_slot takes exactly 11 arguments");
        return NULL;
    }
    step_slot_fn fn = (step_slot_fn)PyLong_AsVoidPtr(args[0]);
    void *blk = PyLong_AsVoidPtr(args[1]);
    void *slabs = PyLong_AsVoidPtr(args[2]);
    long long slot = PyLong_AsLongLong(args[3]);
    void *actions = PyLong_AsVoidPtr(args[4]);
    long dtype = PyLong_AsLong(args[5]);
    unsigned long long call = PyLong_AsUnsignedLongLong(args[6]);
    int pending = PyObject_IsTrue(args[7]);
    unsigned long long pre_call = PyLong_AsUnsignedLongLong(args[8]);
    int want = PyObject_IsTrue(args[9]);
    void *stream = PyLong_AsVoidPtr(args[10]);
    if (PyErr_Occurred()) return NULL;
    if (fn == NULL) {
        PyErr_SetString(PyExc_RuntimeError, "wurm_amd._fastcall: null function address");
        return NULL;
    }
    int rc = fn(blk, slabs, (int64_t)slot, actions, (int)dtype, (uint64_t)call, pending, (uint64_t)pre_call, want, stream);
    return PyLong_FromLong(rc);
}

typedef int (*multi_step_fn)(void *call_block, float *out_f32, uint8_t *out_u8, float *obs, float *obs_after,
                             const int64_t *actions, uint64_t call, int apply_pending, uint64_t pre_call, void *stream);

/* multi_step(function_addr, call_block_addr, out_f32, out_u8, obs, obs_after, actions, call, apply_pending, pre_call,
 * stream) -> int return code of wurm_multi_step_packed at function_addr */
static PyObject *fast_multi_step(PyObject *self, PyObject *const *args, Py_ssize_t nargs)
{
    (void)self;
    if (nargs != 11) {
        PyErr_SetString(PyExc_TypeError, "multi_step takes exactly 11 arguments");
        return NULL;
    }
    multi_step_fn fn = (multi_step_fn)PyLong_AsVoidPtr(args[0]);
    void *blk = PyLong_AsVoidPtr(args[1]);
    float *out_f32 = (float *)PyLong_AsVoidPtr(args[2]);
    uint8_t *out_u8 = (uint8_t *)PyLong_AsVoidPtr(args[3]);
    float *obs = (float *)PyLong_AsVoidPtr(args[4]);
    float *obs_after = (float *)PyLong_AsVoidPtr(args[5]);
    const int64_t *actions = (const int64_t *)PyLong_AsVoidPtr(args[6]);
    unsigned long long call = PyLong_AsUnsignedLongLong(args[7]);
    int pending = PyObject_IsTrue(args[8]);
    unsigned long long pre_call = PyLong_AsUnsignedLongLong(args[9]);
    void *stream = PyLong_AsVoidPtr(args[10]);
    if (PyErr_Occurred()) return NULL;
    if (fn == NULL) {
        PyErr_SetString(PyExc_RuntimeError, "wurm_amd._fastcall: null function address");
        return NULL;
    }
    int rc = fn(blk, out_f32, out_u8, obs, obs_after, actions, (uint64_t)call, pending, (uint64_t)pre_call, stream);
    return PyLong_FromLong(rc);
}

static PyMethodDef fast_methods[] = {
    {"step_slot", (PyCFunction)(void (*)(void))fast_step_slot, METH_FASTCALL, "see fastcall.c"},
    {"multi_step", (PyCFunction)(void (*)(void))fast_multi_step, METH_FASTCALL, "see fastcall.c"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef fast_module = {PyModuleDef_HEAD_INIT, "_fastcall", "per-step call shim (see fastcall.c)", -1,
                                         fast_methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__fastcall(void) { return PyModule_Create(&fast_module); }

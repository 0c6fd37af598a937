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

/* ------------------------------------------------------------------------------------------------ Stepper
 * The bodies of `env.step(actions)` and of the deferred `env.reset(done)` of SingleSnake / SimpleGridworld
 * (wurm_amd/envs/_fast_step.py describes the protocol; PyStepper there is the same logic in Python and the specification
 * of this type).  The loop being served is experiments/main.py:212-227 of the reference,
 *     obs, reward, done, info = env.step(action); env.reset(done)
 * which at 512 envs is bound by host time per iteration: here one iteration costs a handful of C-API calls on the
 * action tensor, one launch, and two prebuilt objects handed back.
 */
#include <structmember.h>

/* wurm_amd/csrc/torchinfo.cpp (optional): the facts about `actions` straight from the at::Tensor */
typedef struct wurm_tensor_info { void *ptr; long long size0; int dtype, dim, contiguous, device; } wurm_tensor_info;
typedef int (*tensor_info_fn)(PyObject *, wurm_tensor_info *);
typedef void *(*raw_stream_fn)(int);
typedef int (*current_device_fn)(void);
typedef int (*alias_free_fn)(PyObject *);

typedef int (*multi_slot_fn)(void *call_block, const void *slabs, int64_t slot, const int64_t *actions, uint64_t call,
                             int apply_pending, uint64_t pre_call, int want_obs_after, void *stream);

typedef struct {
    PyObject_HEAD
    step_slot_fn fn;           /* wurm_single_step_slot / wurm_grid_step_slot; with `multi`: a multi_slot_fn (wurm_multi_step_slot) */
    tensor_info_fn tinfo;      /* all three or none */
    raw_stream_fn raw_stream;
    current_device_fn cur_dev;
    alias_free_fn alias_fn;    /* optional fourth helper: "nobody else holds a tensor on these storages" */
    PyObject *state;           /* tuple of the state tensors a reset may only be postponed for while nobody else holds them, or None */
    PyObject *alias_free;      /* Python callable saying the same (used where alias_fn is missing), or None */
    void *blk, *slabs;
    long long slot, R, num_envs, dev_index, slab_version;
    unsigned long long call, pend_call, steps;
    char ok, pending, last_fresh, want_obs_after, lazy_ok, multi;
    long long num_agents;                         /* MultiSnake: K */
    PyObject *last_out;                           /* what the last step returned */
    PyObject *outs, *done2s, *obs_afters;         /* per slot of the current slab: output tuple, (N,1) done, reset obs */
    PyObject *last_done2, *done_view, *obs_after; /* of the last step */
    PyObject *watch;                              /* a tensor the caller may edit in place (the state), or None ... */
    long long watch_version;                      /* ... and its version counter when last looked at */
    PyObject *get_device, *get_stream;            /* torch's current-device / raw-stream accessors */
    PyObject *dt_i64, *dt_i32, *dt_i16;
} Stepper;

static PyObject *s_dtype, *s_data_ptr, *s_get_device, *s_dim, *s_is_contiguous, *s_version, *s_size, *s_zero;

static void set_obj(PyObject **slot, PyObject *v)
{
    PyObject *old = *slot;
    Py_INCREF(v);
    *slot = v;
    Py_XDECREF(old);
}

static int stepper_init(Stepper *self, PyObject *args, PyObject *kwds)
{
    (void)kwds;
    PyObject *fn, *blk, *slabs, *gd, *gs, *d64, *d32, *d16, *ti = NULL, *rs = NULL, *cd = NULL, *af = NULL;
    if (!PyArg_ParseTuple(args, "OOOOOOOO|OOOO", &fn, &blk, &slabs, &gd, &gs, &d64, &d32, &d16, &ti, &rs, &cd, &af)) return -1;
    self->alias_fn = (af && af != Py_None) ? (alias_free_fn)PyLong_AsVoidPtr(af) : NULL;
    self->fn = (step_slot_fn)PyLong_AsVoidPtr(fn);
    self->tinfo = NULL; self->raw_stream = NULL; self->cur_dev = NULL;
    if (ti && rs && cd && ti != Py_None && rs != Py_None && cd != Py_None) { /* addresses of torchinfo.cpp's helpers */
        self->tinfo = (tensor_info_fn)PyLong_AsVoidPtr(ti);
        self->raw_stream = (raw_stream_fn)PyLong_AsVoidPtr(rs);
        self->cur_dev = (current_device_fn)PyLong_AsVoidPtr(cd);
        if (!self->tinfo || !self->raw_stream || !self->cur_dev) { self->tinfo = NULL; self->raw_stream = NULL; self->cur_dev = NULL; }
    }
    self->blk = PyLong_AsVoidPtr(blk);
    self->slabs = PyLong_AsVoidPtr(slabs);
    if (PyErr_Occurred()) return -1;
    if (self->fn == NULL) {
        PyErr_SetString(PyExc_RuntimeError, "wurm_amd._fastcall.Stepper: null function address");
        return -1;
    }
    set_obj(&self->get_device, gd); set_obj(&self->get_stream, gs);
    set_obj(&self->dt_i64, d64); set_obj(&self->dt_i32, d32); set_obj(&self->dt_i16, d16);
    set_obj(&self->outs, Py_None); set_obj(&self->done2s, Py_None); set_obj(&self->obs_afters, Py_None);
    set_obj(&self->last_done2, Py_None); set_obj(&self->done_view, Py_None); set_obj(&self->obs_after, Py_None);
    set_obj(&self->watch, Py_None);
    set_obj(&self->last_out, Py_None);
    set_obj(&self->state, Py_None);
    set_obj(&self->alias_free, Py_None);
    self->multi = 0;
    self->num_agents = 0;
    self->watch_version = -1;
    self->slot = self->R = 0;
    self->slab_version = -1;
    self->call = self->pend_call = self->steps = 0;
    self->ok = self->pending = self->last_fresh = self->want_obs_after = self->lazy_ok = 0;
    return 0;
}

static int stepper_traverse(Stepper *self, visitproc visit, void *arg)
{
    Py_VISIT(self->outs); Py_VISIT(self->done2s); Py_VISIT(self->obs_afters);
    Py_VISIT(self->last_done2); Py_VISIT(self->done_view); Py_VISIT(self->obs_after); Py_VISIT(self->watch);
    Py_VISIT(self->last_out); Py_VISIT(self->state); Py_VISIT(self->alias_free);
    Py_VISIT(self->get_device); Py_VISIT(self->get_stream);
    Py_VISIT(self->dt_i64); Py_VISIT(self->dt_i32); Py_VISIT(self->dt_i16);
    return 0;
}

static int stepper_clear(Stepper *self)
{
    Py_CLEAR(self->outs); Py_CLEAR(self->done2s); Py_CLEAR(self->obs_afters);
    Py_CLEAR(self->last_done2); Py_CLEAR(self->done_view); Py_CLEAR(self->obs_after); Py_CLEAR(self->watch);
    Py_CLEAR(self->last_out); Py_CLEAR(self->state); Py_CLEAR(self->alias_free);
    Py_CLEAR(self->get_device); Py_CLEAR(self->get_stream);
    Py_CLEAR(self->dt_i64); Py_CLEAR(self->dt_i32); Py_CLEAR(self->dt_i16);
    return 0;
}

static void stepper_dealloc(Stepper *self)
{
    PyObject_GC_UnTrack(self);
    stepper_clear(self);
    Py_TYPE(self)->tp_free((PyObject *)self);
}

/* result of a no-argument method as a C long; -2 with an exception set on failure (the methods used return >= -1) */
static long method_long(PyObject *obj, PyObject *name)
{
    PyObject *r = PyObject_CallMethodNoArgs(obj, name);
    if (!r) return -2;
    long v = PyLong_Check(r) ? PyLong_AsLong(r) : (long)PyObject_IsTrue(r);
    Py_DECREF(r);
    return v;
}

/* step(actions) -> the prebuilt output tuple of this step;  None: the caller has to prepare something first (new slab,
 * state tensor to re-validate or edited in place, actions on another device / not a contiguous vector, another device current) and call
 * again;  a non-zero int: the entry point's error code.  Argument errors are raised as the reference raises them
 * (single_snake.py:198-203; int16 passes its check and fails in scatter_ at :229). */
static PyObject *stepper_finish(Stepper *self, long long i, int rc);

static PyObject *stepper_step(Stepper *self, PyObject *actions)
{
    wurm_tensor_info ti;
    if (self->multi) {
        PyErr_SetString(PyExc_RuntimeError, "Stepper.step: this machine drives wurm_multi_step_slot (step_multi)");
        return NULL;
    }
    if (self->tinfo && self->tinfo(actions, &ti) == 0 && ti.dim >= 1) {
        /* the same checks, in the same order, on the facts torchinfo.cpp read from the at::Tensor (ScalarType: Short = 2,
         * Int = 3, Long = 4); anything that is not a tensor of at least one dimension takes the generic path below */
        int code;
        if (ti.dtype == 4) code = 0;
        else if (ti.dtype == 3) code = 1;
        else {
            if (ti.dtype == 2) PyErr_SetString(PyExc_RuntimeError, "scatter_(): Expected dtype int32/int64 for index");
            else PyErr_SetString(PyExc_TypeError, "actions Tensor must be an integer type i.e. "
                                                  "{torch.ShortTensor, torch.IntTensor, torch.LongTensor}");
            return NULL;
        }
        if (ti.size0 != self->num_envs) {
            PyErr_SetString(PyExc_RuntimeError, "Must have the same number of actions as environments.");
            return NULL;
        }
        const long long i = self->slot;
        if (!self->ok || i >= self->R || (self->want_obs_after && self->obs_afters == Py_None)) Py_RETURN_NONE;
        if (self->watch != NULL && self->watch != Py_None) {
            long long ver = -1;
            PyObject *vo = PyObject_GetAttr(self->watch, s_version);
            if (vo) {
                ver = PyLong_AsLongLong(vo);
                Py_DECREF(vo);
                if (ver == -1 && PyErr_Occurred()) PyErr_Clear();
            } else {
                PyErr_Clear();
            }
            if (ver != self->watch_version) Py_RETURN_NONE;
        }
        if ((long long)ti.device != self->dev_index || ti.dim != 1 || !ti.contiguous) Py_RETURN_NONE;
        if ((long long)self->cur_dev() != self->dev_index) Py_RETURN_NONE; /* kernels launch on the current device */
        void *stream = self->raw_stream((int)self->dev_index);
        const int rc = self->fn(self->blk, self->slabs, (int64_t)i, ti.ptr, code, (uint64_t)self->call, self->pending,
                                (uint64_t)self->pend_call, self->want_obs_after, stream);
        return stepper_finish(self, i, rc);
    }
    PyObject *dt = PyObject_GetAttr(actions, s_dtype);
    if (!dt) return NULL;
    int code;
    if (dt == self->dt_i64) code = 0;
    else if (dt == self->dt_i32) code = 1;
    else {
        const int is16 = dt == self->dt_i16;
        Py_DECREF(dt);
        if (is16) PyErr_SetString(PyExc_RuntimeError, "scatter_(): Expected dtype int32/int64 for index");
        else PyErr_SetString(PyExc_TypeError, "actions Tensor must be an integer type i.e. "
                                              "{torch.ShortTensor, torch.IntTensor, torch.LongTensor}");
        return NULL;
    }
    Py_DECREF(dt);
    PyObject *n0 = PyObject_CallMethodOneArg(actions, s_size, s_zero); /* (len() of a tensor is a Python-level method) */
    if (!n0) return NULL;
    const long long n = PyLong_AsLongLong(n0);
    Py_DECREF(n0);
    if (n == -1 && PyErr_Occurred()) return NULL;
    if (n != self->num_envs) {
        PyErr_SetString(PyExc_RuntimeError, "Must have the same number of actions as environments.");
        return NULL;
    }
    const long long i = self->slot;
    if (!self->ok || i >= self->R || (self->want_obs_after && self->obs_afters == Py_None)) Py_RETURN_NONE;
    if (self->watch != NULL && self->watch != Py_None) { /* the state tensor edited in place since the caller took it: the caller re-validates */
        long long ver = -1;
        PyObject *vo = PyObject_GetAttr(self->watch, s_version);
        if (vo) {
            ver = PyLong_AsLongLong(vo);
            Py_DECREF(vo);
            if (ver == -1 && PyErr_Occurred()) PyErr_Clear();
        } else {
            PyErr_Clear();
        }
        if (ver != self->watch_version) Py_RETURN_NONE;
    }
    long v = method_long(actions, s_get_device);
    if (v == -2) return NULL;
    if ((long long)v != self->dev_index) Py_RETURN_NONE;
    v = method_long(actions, s_dim);
    if (v == -2) return NULL;
    if (v != 1) Py_RETURN_NONE;
    v = method_long(actions, s_is_contiguous);
    if (v == -2) return NULL;
    if (!v) Py_RETURN_NONE;
    PyObject *cur = PyObject_CallNoArgs(self->get_device);
    if (!cur) return NULL;
    const long cur_dev = PyLong_AsLong(cur);
    Py_DECREF(cur);
    if (cur_dev == -1 && PyErr_Occurred()) return NULL;
    if ((long long)cur_dev != self->dev_index) Py_RETURN_NONE; /* kernels launch on the current device */
    PyObject *idx = PyLong_FromLongLong(self->dev_index);
    if (!idx) return NULL;
    PyObject *st = PyObject_CallOneArg(self->get_stream, idx);
    Py_DECREF(idx);
    if (!st) return NULL;
    void *stream = PyLong_AsVoidPtr(st);
    Py_DECREF(st);
    if (PyErr_Occurred()) return NULL;
    PyObject *pp = PyObject_CallMethodNoArgs(actions, s_data_ptr);
    if (!pp) return NULL;
    void *aptr = PyLong_AsVoidPtr(pp);
    Py_DECREF(pp);
    if (PyErr_Occurred()) return NULL;

    const int rc = self->fn(self->blk, self->slabs, (int64_t)i, aptr, code, (uint64_t)self->call, self->pending,
                            (uint64_t)self->pend_call, self->want_obs_after, stream);
    return stepper_finish(self, i, rc);
}

/* bookkeeping after the launch of slot i */
static PyObject *stepper_finish(Stepper *self, long long i, int rc)
{
    if (rc) return PyLong_FromLong(rc); /* nothing consumed: the counter, the postponed reset and the slot stay */
    self->call += 1;
    self->steps += 1;
    self->pending = 0;
    self->slot = i + 1;
    set_obj(&self->last_done2, PyList_GET_ITEM(self->done2s, i));
    set_obj(&self->done_view, Py_None);
    self->last_fresh = 1;
    set_obj(&self->obs_after, self->want_obs_after ? PyList_GET_ITEM(self->obs_afters, i) : Py_None);
    PyObject *out = PyList_GET_ITEM(self->outs, i);
    set_obj(&self->last_out, out);
    Py_INCREF(out);
    return out;
}

/* ---- MultiSnake: the same machine over wurm_multi_step_slot (PyStepper.step_multi / launch_multi are the specification) */

static PyObject *stepper_launch_multi_ptr(Stepper *self, const int64_t *a_ptr, void *stream)
{
    const long long i = self->slot;
    const int rc = ((multi_slot_fn)self->fn)(self->blk, self->slabs, (int64_t)i, a_ptr, (uint64_t)self->call, self->pending,
                                             (uint64_t)self->pend_call, self->want_obs_after, stream);
    return stepper_finish(self, i, rc);
}

/* launch_multi(a_ptr): the launch of slot `slot` on a prepared (K, N) int64 action block, and the bookkeeping after it */
static PyObject *stepper_launch_multi(Stepper *self, PyObject *arg)
{
    const int64_t *a_ptr = (const int64_t *)PyLong_AsVoidPtr(arg);
    if (PyErr_Occurred()) return NULL;
    if (!self->multi || self->slot >= self->R || self->outs == Py_None) {
        PyErr_SetString(PyExc_RuntimeError, "launch_multi: no slot prepared");
        return NULL;
    }
    void *stream;
    if (self->raw_stream) stream = self->raw_stream((int)self->dev_index);
    else {
        PyObject *idx = PyLong_FromLongLong(self->dev_index);
        if (!idx) return NULL;
        PyObject *st = PyObject_CallOneArg(self->get_stream, idx);
        Py_DECREF(idx);
        if (!st) return NULL;
        stream = PyLong_AsVoidPtr(st);
        Py_DECREF(st);
        if (PyErr_Occurred()) return NULL;
    }
    return stepper_launch_multi_ptr(self, a_ptr, stream);
}

/* step_multi(actions: dict) -> the prebuilt (observations, rewards, dones, info) of this step; None: the caller prepares
 * something first (new slab, state to re-validate, actions of another type or device, not a plain dict, another device
 * current, no tensor-facts helper) and calls launch_multi; False: all that is missing is the (K, N) action block — the K
 * int64 device vectors are not the rows of one tensor: the caller stacks them and calls launch_multi; a non-zero int: the
 * entry point's error code.  Argument errors as the reference raises them (multi_snake.py:463-472), in its order. */
static PyObject *stepper_step_multi(Stepper *self, PyObject *actions)
{
    if (!self->multi || !self->tinfo || !PyDict_CheckExact(actions)) Py_RETURN_NONE;
    if ((long long)PyDict_GET_SIZE(actions) != self->num_agents) {
        PyErr_SetString(PyExc_RuntimeError, "Must have a Tensor of actions for each snake");
        return NULL;
    }
    Py_ssize_t pos = 0;
    PyObject *key, *val;
    const char *a0 = NULL;
    long long n = 0;
    int rows_ok = 1, stackable = 1; /* stackable: K int64 vectors on the device — all that keeps them from being used where they lie
                                     * is where they lie (a policy that emits one tensor per agent, experiments/multiagent.py) */
    const long long row = 8 * self->num_envs;
    while (PyDict_Next(actions, &pos, &key, &val)) {
        wurm_tensor_info ti;
        if (self->tinfo(val, &ti) != 0 || ti.dim < 1) Py_RETURN_NONE; /* not a tensor: the generic path raises what Python would */
        if (ti.dtype != 2 && ti.dtype != 3 && ti.dtype != 4) {
            PyErr_SetString(PyExc_TypeError, "actions Tensor must be an integer type i.e. "
                                             "{torch.ShortTensor, torch.IntTensor, torch.LongTensor}");
            return NULL;
        }
        if (ti.size0 != self->num_envs) {
            PyErr_SetString(PyExc_RuntimeError, "Must have the same number of actions as environments.");
            return NULL;
        }
        if (ti.dtype != 4 || ti.dim != 1 || (long long)ti.device != self->dev_index) stackable = 0;
        if (rows_ok) {
            if (ti.dtype != 4 || ti.dim != 1 || !ti.contiguous || (long long)ti.device != self->dev_index) rows_ok = 0;
            else if (n == 0) a0 = (const char *)ti.ptr;
            else if ((const char *)ti.ptr != a0 + n * row) rows_ok = 0;
        }
        ++n;
    }
    if (!self->ok || self->slot >= self->R || (self->want_obs_after && self->obs_afters == Py_None)) Py_RETURN_NONE;
    if ((long long)self->cur_dev() != self->dev_index) Py_RETURN_NONE; /* kernels launch on the current device */
    if (!rows_ok) { /* False: everything is in place but the action block — the caller stacks the K vectors and calls launch_multi */
        if (stackable) Py_RETURN_FALSE;
        Py_RETURN_NONE;
    }
    return stepper_launch_multi_ptr(self, (const int64_t *)a0, self->raw_stream((int)self->dev_index));
}

/* reset_lazy(done, return_observations) -> what reset(done) returns if the reset could be postponed into the next
 * step's launch (None, or the observation the last step's launch already wrote), else NotImplemented */
static PyObject *stepper_reset_lazy(Stepper *self, PyObject *const *args, Py_ssize_t nargs)
{
    if (nargs != 2) {
        PyErr_SetString(PyExc_TypeError, "reset_lazy takes exactly 2 arguments");
        return NULL;
    }
    PyObject *done = args[0];
    const int want = PyObject_IsTrue(args[1]);
    if (want < 0) return NULL;
    if (self->last_fresh && self->lazy_ok &&
        (done == self->last_done2 || (self->done_view != Py_None && done == self->done_view))) {
        /* only while nobody else holds a tensor on the state's storage: through an alias the caller could read or edit the
         * un-reset state, which the reference would show reset */
        int free_ = 1;
        if (self->alias_fn && self->state != Py_None) free_ = self->alias_fn(self->state);
        else if (self->alias_free != Py_None) free_ = -1;
        if (free_ < 0) {
            if (self->alias_free == Py_None) free_ = 0;
            else {
                PyObject *r = PyObject_CallNoArgs(self->alias_free);
                if (!r) return NULL;
                free_ = PyObject_IsTrue(r);
                Py_DECREF(r);
                if (free_ < 0) return NULL;
            }
        }
        if (!free_) {
            Py_INCREF(Py_NotImplemented);
            return Py_NotImplemented;
        }
        long long ver = -1;
        PyObject *vo = PyObject_GetAttr(done, s_version);
        if (vo) {
            ver = PyLong_AsLongLong(vo);
            Py_DECREF(vo);
            if (ver == -1 && PyErr_Occurred()) { PyErr_Clear(); ver = -1; }
        } else {
            PyErr_Clear(); /* inference tensors do not track versions: cannot prove `done` is unmodified */
        }
        if (ver >= 0 && ver == self->slab_version) {
            if (!want) {
                self->want_obs_after = 0;
                self->pending = 1; self->pend_call = self->call; self->call += 1; self->last_fresh = 0;
                Py_RETURN_NONE;
            }
            if (self->obs_after != Py_None) {
                PyObject *obs = self->obs_after; /* reference handed over */
                Py_INCREF(Py_None);
                self->obs_after = Py_None;
                self->pending = 1; self->pend_call = self->call; self->call += 1; self->last_fresh = 0;
                return obs;
            }
            self->want_obs_after = 1; /* from the next step on, the step launch also writes this observation */
        } else if (ver >= 0) {
            self->slab_version = ver; /* an in-place edit of one step's flags costs one eager reset, not the slab's rest */
        }
    }
    Py_INCREF(Py_NotImplemented);
    return Py_NotImplemented;
}

static PyMemberDef stepper_members[] = {
    {"slot", T_LONGLONG, offsetof(Stepper, slot), 0, NULL},
    {"R", T_LONGLONG, offsetof(Stepper, R), 0, NULL},
    {"num_envs", T_LONGLONG, offsetof(Stepper, num_envs), 0, NULL},
    {"dev_index", T_LONGLONG, offsetof(Stepper, dev_index), 0, NULL},
    {"slab_version", T_LONGLONG, offsetof(Stepper, slab_version), 0, NULL},
    {"call", T_ULONGLONG, offsetof(Stepper, call), 0, NULL},
    {"pend_call", T_ULONGLONG, offsetof(Stepper, pend_call), 0, NULL},
    {"steps", T_ULONGLONG, offsetof(Stepper, steps), 0, NULL},
    {"ok", T_BOOL, offsetof(Stepper, ok), 0, NULL},
    {"pending", T_BOOL, offsetof(Stepper, pending), 0, NULL},
    {"last_fresh", T_BOOL, offsetof(Stepper, last_fresh), 0, NULL},
    {"want_obs_after", T_BOOL, offsetof(Stepper, want_obs_after), 0, NULL},
    {"lazy_ok", T_BOOL, offsetof(Stepper, lazy_ok), 0, NULL},
    {"multi", T_BOOL, offsetof(Stepper, multi), 0, NULL},
    {"num_agents", T_LONGLONG, offsetof(Stepper, num_agents), 0, NULL},
    {"last_out", T_OBJECT, offsetof(Stepper, last_out), 0, NULL},
    {"state", T_OBJECT, offsetof(Stepper, state), 0, NULL},
    {"alias_free", T_OBJECT, offsetof(Stepper, alias_free), 0, NULL},
    {"outs", T_OBJECT, offsetof(Stepper, outs), 0, NULL},
    {"done2s", T_OBJECT, offsetof(Stepper, done2s), 0, NULL},
    {"obs_afters", T_OBJECT, offsetof(Stepper, obs_afters), 0, NULL},
    {"last_done2", T_OBJECT, offsetof(Stepper, last_done2), 0, NULL},
    {"done_view", T_OBJECT, offsetof(Stepper, done_view), 0, NULL},
    {"obs_after", T_OBJECT, offsetof(Stepper, obs_after), 0, NULL},
    {"watch", T_OBJECT, offsetof(Stepper, watch), 0, NULL},
    {"watch_version", T_LONGLONG, offsetof(Stepper, watch_version), 0, NULL},
    {NULL, 0, 0, 0, NULL}};

static PyMethodDef stepper_methods[] = {
    {"step", (PyCFunction)stepper_step, METH_O, "see fastcall.c"},
    {"reset_lazy", (PyCFunction)(void (*)(void))stepper_reset_lazy, METH_FASTCALL, "see fastcall.c"},
    {"step_multi", (PyCFunction)stepper_step_multi, METH_O, "see fastcall.c"},
    {"launch_multi", (PyCFunction)stepper_launch_multi, METH_O, "see fastcall.c"},
    {NULL, NULL, 0, NULL}};

static PyTypeObject StepperType = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "wurm_amd._fastcall.Stepper",
    .tp_basicsize = sizeof(Stepper),
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_GC,
    .tp_doc = "per-step state machine of SingleSnake / SimpleGridworld (see fastcall.c)",
    .tp_new = PyType_GenericNew,
    .tp_init = (initproc)stepper_init,
    .tp_dealloc = (destructor)stepper_dealloc,
    .tp_traverse = (traverseproc)stepper_traverse,
    .tp_clear = (inquiry)stepper_clear,
    .tp_members = stepper_members,
    .tp_methods = stepper_methods,
};

static PyMethodDef fast_methods[] = {
    {"step_slot", (PyCFunction)(void (*)(void))fast_step_slot, METH_FASTCALL, "see fastcall.c"},
    {"multi_step", (PyCFunction)(void (*)(void))fast_multi_step, METH_FASTCALL, "see fastcall.c"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef fast_module = {PyModuleDef_HEAD_INIT, "_fastcall", "per-step call shim (see fastcall.c)", -1,
                                         fast_methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__fastcall(void)
{
    s_dtype = PyUnicode_InternFromString("dtype");
    s_data_ptr = PyUnicode_InternFromString("data_ptr");
    s_get_device = PyUnicode_InternFromString("get_device");
    s_dim = PyUnicode_InternFromString("dim");
    s_is_contiguous = PyUnicode_InternFromString("is_contiguous");
    s_version = PyUnicode_InternFromString("_version");
    s_size = PyUnicode_InternFromString("size");
    s_zero = PyLong_FromLong(0);
    if (!s_size || !s_zero || !s_dtype || !s_data_ptr || !s_get_device || !s_dim || !s_is_contiguous || !s_version) return NULL;
    if (PyType_Ready(&StepperType) < 0) return NULL;
    PyObject *m = PyModule_Create(&fast_module);
    if (!m) return NULL;
    Py_INCREF(&StepperType);
    if (PyModule_AddObject(m, "Stepper", (PyObject *)&StepperType) < 0) {
        Py_DECREF(&StepperType);
        Py_DECREF(m);
        return NULL;
    }
    return m;
}

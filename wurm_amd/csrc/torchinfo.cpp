// torchinfo.cpp — what the per-step state machine (fastcall.c: Stepper) needs to know about an `actions` tensor and
// about torch's current device / stream, read straight from the at::Tensor behind the Python object.  Through the Python
// methods (dtype, size(0), get_device(), dim(), is_contiguous(), data_ptr(), the device and stream accessors) the same
// facts cost seven argument-parsing round trips, ~3 us of the ~7 us a `step` call spends on the host.
// OPTIONAL: built by the Makefile's best-effort `torchinfo` target against the torch headers of the running interpreter
// into wurm_amd/libwurm_torchinfo.so; without it the Stepper uses the generic attribute path.  Host C++, no device code.
#include <Python.h>
#include <torch/csrc/autograd/python_variable.h>
#include <c10/hip/HIPStream.h>
#include <c10/hip/HIPFunctions.h>
#include <stdint.h>

extern "C" {
typedef struct wurm_tensor_info { void *ptr; long long size0; int dtype, dim, contiguous, device; } wurm_tensor_info;

int wurm_torch_tensor_info(PyObject *obj, wurm_tensor_info *o)
{
    if (!THPVariable_Check(obj)) return -1;
    const at::Tensor &t = THPVariable_Unpack(obj);
    if (!t.defined() || !t.has_storage()) return -1;
    o->ptr = t.data_ptr();
    o->dtype = (int)t.scalar_type();
    o->dim = (int)t.dim();
    o->size0 = t.dim() > 0 ? (long long)t.size(0) : -1;
    o->contiguous = t.is_contiguous() ? 1 : 0;
    o->device = t.is_cuda() ? (int)t.get_device() : -1;
    return 0;
}
/* The leading `lead` dimensions of a contiguous tensor unbound at once: a list of prod(sizes[:lead]) tensors of shape
 * sizes[lead:], row-major — views on the same storage that share the base's version counter (an in-place edit through any of
 * them is seen through all of them, as with `unbind`), built straight from TensorImpls: ~0.15 us each instead of the
 * ~0.65 us of a dispatched `unbind` / `select`.  The host classes carve the per-agent output tensors of a whole slab of steps
 * this way (MultiSnake: 8 K + 1 tensors per step, wurm/envs/multi_snake.py:701-729 — most of what a step cost on the host).
 * With lead == 2 only the rows lo <= j < hi of the second dimension are made (hi < 0: all): the agent-major part of the packed
 * output blocks.  The base must not require grad (outputs of the kernels never do).  NULL with a Python exception set on failure. */
PyObject *wurm_torch_row_views(PyObject *obj, long long lead, long long lo, long long hi)
{
    try {
        if (!THPVariable_Check(obj)) { PyErr_SetString(PyExc_TypeError, "row_views: not a tensor"); return nullptr; }
        const at::Tensor &b = THPVariable_Unpack(obj);
        if (!b.defined() || !b.has_storage() || !b.is_contiguous() || b.requires_grad() || lead < 1 || lead > b.dim()) {
            PyErr_SetString(PyExc_ValueError, "row_views: needs a contiguous tensor without grad and 1 <= lead <= dim");
            return nullptr;
        }
        int64_t n = 1, step = 1;
        for (int64_t i = 0; i < lead; ++i) n *= b.size(i);
        for (int64_t i = lead; i < b.dim(); ++i) step *= b.size(i);
        const int64_t inner = lead == 2 ? b.size(1) : 1; // rows per outer index, of which [lo, hi) are wanted
        if (hi < 0 || lead != 2) { lo = 0; hi = inner; }
        if (lo < 0 || hi > inner || lo > hi) { PyErr_SetString(PyExc_ValueError, "row_views: bad row range"); return nullptr; }
        const int64_t outer = lead == 2 ? b.size(0) : n, per = lead == 2 ? hi - lo : 1;
        const auto sizes = b.sizes().slice(lead), strides = b.strides().slice(lead);
        c10::TensorImpl *bi = b.unsafeGetTensorImpl();
        PyObject *list = PyList_New((Py_ssize_t)(outer * per));
        if (!list) return nullptr;
        for (int64_t k = 0; k < outer * per; ++k) {
            const int64_t i = lead == 2 ? (k / per) * inner + lo + k % per : k; // flat row index in the base
            auto impl = c10::make_intrusive<c10::TensorImpl>(c10::TensorImpl::VIEW, c10::Storage(bi->storage()), bi->key_set(),
                                                             bi->dtype());
            impl->set_storage_offset(bi->storage_offset() + i * step);
            impl->set_sizes_and_strides(sizes, strides);
            impl->set_version_counter(bi->version_counter());
            PyObject *t = THPVariable_Wrap(at::Tensor(std::move(impl)));
            if (!t) { Py_DECREF(list); return nullptr; }
            PyList_SET_ITEM(list, (Py_ssize_t)k, t);
        }
        return list;
    } catch (const std::exception &e) {
        PyErr_SetString(PyExc_RuntimeError, e.what());
        return nullptr;
    } catch (...) {
        PyErr_SetString(PyExc_RuntimeError, "row_views: unknown C++ exception");
        return nullptr;
    }
}
/* 1 if every tensor of the tuple is the ONLY tensor on its storage (nobody else holds the state tensor or a view of it),
 * 0 if some storage is shared, -1 if the argument is not a tuple of tensors.  The host classes postpone `reset(done)` only
 * while that holds (wurm_amd/envs/_fast_step.py: _alias_free).
 * The storage's own Python wrapper is not a holder: once `t.untyped_storage()`, `copy.deepcopy(t)`, `pickle.dumps(t)` or
 * `t.is_shared()` has created it, torch keeps it alive on the StorageImpl (PyObject preservation) and it owns one
 * reference of the storage for good — use_count() == 2 with nobody else there (with the bare `== 1` test a
 * deepcopy of an env switched the postponed reset off on both objects for ever).  So the wrapper's reference is discounted,
 * which is also what the Python twin does (`_storage_use_count(...) <= 2` with its own handle kept). */
int wurm_torch_alias_free(PyObject *seq)
{
    if (!PyTuple_Check(seq)) return -1;
    const Py_ssize_t n = PyTuple_GET_SIZE(seq);
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *o = PyTuple_GET_ITEM(seq, i);
        if (!THPVariable_CheckExact(o)) return -1;
        const at::Tensor &t = THPVariable_Unpack(o);
        if (!t.defined() || !t.has_storage()) return -1;
        const c10::Storage &st = t.unsafeGetTensorImpl()->unsafe_storage();
        const size_t wrapper = st.unsafeGetStorageImpl()->pyobj_slot()->load_pyobj() != nullptr ? 1 : 0;
        if (st.use_count() != 1 + wrapper) return 0;
    }
    return 1;
}
/* (no exception may cross into the C caller: NULL stream = torch's default stream, device -2 = none) */
void *wurm_torch_raw_stream(int idx)
{
    try {
        return (void *)c10::hip::getCurrentHIPStream((c10::DeviceIndex)idx).stream();
    } catch (...) {
        return nullptr;
    }
}
int wurm_torch_current_device(void)
{
    try {
        return (int)c10::hip::current_device();
    } catch (...) {
        return -2;
    }
}
}

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

"""Can a kernel write its result straight into torch's pinned host memory (no copy kernel)?  Development probe."""
import ctypes, time, torch
x = torch.arange(1024, dtype=torch.float64, device="cuda")
host = torch.zeros(4, dtype=torch.float64).pin_memory()
hip = ctypes.CDLL("libamdhip64.so")
dptr = ctypes.c_void_p()
rc = hip.hipHostGetDevicePointer(ctypes.byref(dptr), ctypes.c_void_p(host.data_ptr()), 0)
print("hipHostGetDevicePointer rc", rc, hex(dptr.value or 0), hex(host.data_ptr()))
# torch view over the device alias of the pinned buffer
try:
    from torch.utils import dlpack  # noqa
    alias = torch.empty(0)
except Exception as e:
    print(e)
# use a torch kernel writing to a tensor created from the raw pointer via __cuda_array_interface__
class Raw:
    def __init__(self, p, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (p, False), "version": 2}
al = torch.as_tensor(Raw(dptr.value, 4), device="cuda")
ev = torch.cuda.Event()
t0 = time.perf_counter()
for k in range(100):
    al[0:1].copy_(x[k:k + 1] * 2)
    ev.record(); ev.synchronize()
    assert host[0].item() == 2.0 * k, (k, host[0].item())
print("direct device->pinned writes visible after event sync: OK", (time.perf_counter() - t0) / 100 * 1e6, "us/iter")

"""Is hipMemset on device memory asynchronous with respect to the host on this ROCm?  (CUDA's cudaMemset is; engine creation
used plain hipMemset for its buffers' initial zeros until round 5.)  Times the call and the hipDeviceSynchronize behind it."""
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
p = ctypes.c_void_p()
nbytes = 8 << 30
assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)) == 0
hip.hipDeviceSynchronize()
for rep in range(3):
    t0 = time.perf_counter()
    assert hip.hipMemset(p, 0, ctypes.c_size_t(nbytes)) == 0
    t1 = time.perf_counter()
    hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    print(f"hipMemset of 8 GiB: the call returned after {1e3 * (t1 - t0):.3f} ms, the device was idle {1e3 * (t2 - t1):.3f} ms later")
hip.hipFree(p)

# ... and hipMemcpy from PAGEABLE host memory (the layout uploads): does the call return before the data is on the device?
import numpy as np
nb = 1 << 30
src = np.ones(nb, dtype=np.uint8)
assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nb)) == 0
for rep in range(3):
    t0 = time.perf_counter()
    assert hip.hipMemcpy(p, src.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(nb), 1) == 0
    t1 = time.perf_counter()
    hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    print(f"hipMemcpy H2D of 1 GiB pageable: the call returned after {1e3 * (t1 - t0):.3f} ms, the device was idle {1e3 * (t2 - t1):.3f} ms later")
small = np.ones(4096, dtype=np.uint8)
for rep in range(3):
    t0 = time.perf_counter()
    assert hip.hipMemcpy(p, small.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(4096), 1) == 0
    t1 = time.perf_counter()
    hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    print(f"hipMemcpy H2D of 4 KiB pageable: the call returned after {1e3 * (t1 - t0):.3f} ms, the device was idle {1e3 * (t2 - t1):.3f} ms later")
hip.hipFree(p)

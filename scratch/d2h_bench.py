import time, torch, numpy as np, ctypes
n = 1024*1024*4
dev = torch.zeros(n, dtype=torch.float32, device="cuda")
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0)/reps
pageable = torch.empty(n, dtype=torch.float32)
pinned = torch.empty(n, dtype=torch.float32, pin_memory=True)
print("pageable  %.3f ms  %.1f GB/s" % (t(lambda: pageable.copy_(dev))*1e3, n*4/t(lambda: pageable.copy_(dev))/1e9))
print("pinned    %.3f ms  %.1f GB/s" % (t(lambda: pinned.copy_(dev, non_blocking=True))*1e3, n*4/t(lambda: pinned.copy_(dev, non_blocking=True))/1e9))
arr = np.empty(n, np.float32)
rt = ctypes.CDLL("libamdhip64.so")
rc = rt.hipHostRegister(ctypes.c_void_p(arr.ctypes.data), ctypes.c_size_t(n*4), 0)
print("hipHostRegister rc", rc)
reg = torch.from_numpy(arr)
def cp():
    rt.hipMemcpyAsync(ctypes.c_void_p(arr.ctypes.data), ctypes.c_void_p(dev.data_ptr()), ctypes.c_size_t(n*4), 2, None)
    rt.hipStreamSynchronize(None)
print("registered numpy %.3f ms  %.1f GB/s" % (t(cp)*1e3, n*4/t(cp)/1e9))
small = torch.empty(256*256*4, dtype=torch.float32, pin_memory=True); sdev = torch.zeros(256*256*4, device="cuda")
print("pinned 1 MB %.3f ms" % (t(lambda: small.copy_(sdev, non_blocking=True))*1e3))

"""Dev tool: does the rate of a streaming kernel depend on WHICH allocation it runs over (physical placement), within one process?
Allocates N buffers of S GiB one after the other and times a device-to-device copy and a fill over each."""
import sys, time
import torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
S = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
n = int(S*(1 << 30)/4)
bufs = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(N)]
torch.cuda.synchronize()


def rate(fn, bytes_moved):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 3*bytes_moved/(e0.elapsed_time(e1)*1e-3)/1e9


out = []
for i, b in enumerate(bufs):
    half = n//2
    w = rate(lambda: b.fill_(1.0), n*4)
    c = rate(lambda: b[half:half*2].copy_(b[:half]), half*4*2)
    out.append((b.data_ptr(), w, c))
print(" ".join("%.0f/%.0f" % (w, c) for _, w, c in out), "  (fill / copy GB/s per buffer, in allocation order; first at 0x%x)" % out[0][0])

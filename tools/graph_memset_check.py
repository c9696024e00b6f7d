"""hipMemsetAsync under torch.cuda.graph on this stack (ROCm 7.2, torch 2.10): from the second replay on the memset node fills its
range with a stale 16-byte pattern (two pointer-like words: denormals ~7e-310) instead of zeros.  Why csrc/step.hip clears
with a kernel of its own (zero_ranges).    python tools/graph_memset_check.py"""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device("cuda:0")
for n in (40000, 34147 + 1000):
    x = torch.full((n + 64,), 7.0, device=dev, dtype=torch.float64)
    y = torch.zeros(3, 8, device=dev, dtype=torch.float64)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        hip.hipMemsetAsync(x.data_ptr(), 0, n * 8, s.cuda_stream)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rc = hip.hipMemsetAsync(x.data_ptr(), 0, n * 8, torch.cuda.current_stream().cuda_stream)
        y[0].copy_(x[:8]); y[1].copy_(x[n - 8:n]); y[2].copy_(x[n // 2:n // 2 + 8])
        x.fill_(7.0)
    for r in range(3):
        g.replay(); torch.cuda.synchronize()
        print(n, "replay", r, "rc", rc, "start", y[0].tolist()[:3], "end", y[1].tolist()[-3:], "mid", y[2].tolist()[:2])

#!/usr/bin/env python3
"""lgn_step_finalize_f64 (L1 + Adam + loss assembly) in isolation: n parameters, nB per-jet loss terms.   python tools/adam_bench.py [n] [nB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lgn-autoencoder_amd"))
import ctypes as C
import torch
from lgn import _native as Nn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 63510
nB = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
w = torch.randn(n, device=dev, dtype=torch.float64); g = torch.randn_like(w); m = torch.zeros_like(w); v = torch.zeros_like(w)
lp = torch.rand(nB, device=dev, dtype=torch.float64)
step = torch.zeros(1, device=dev, dtype=torch.int64)
out = torch.zeros(3 + Nn.FINALIZE_SCRATCH, device=dev, dtype=torch.float64)
L = Nn.lib(); P = Nn.ptr
def fn():
    Nn._check(L.lgn_step_finalize_f64(P(w), P(g), n, P(lp), nB, 1e-8, P(m), P(v), P(step), 5e-4, 0.9, 0.999, 1e-8, 1, P(out), Nn.stream_ptr()), "finalize")
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): fn()
e1.record(); torch.cuda.synchronize()
print(f"finalize n={n} nB={nB}: {e0.elapsed_time(e1) * 20:.1f} us per call; step counter {int(step.item())}, loss {float(out[0]):.6f}")

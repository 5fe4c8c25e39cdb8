"""Micro-benchmark (GPU): 3x3 halo conv time vs Cin at fixed spatial size -> per-slice cost and fixed per-tile overhead."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvcs_amd import ops

dev = "cuda:0"
def bench(B, S, Cin, Cout, reps=10):
    x = torch.randn(B, S, S, Cin, device=dev).to(torch.bfloat16)
    if os.environ.get("DATA") == "zeros": x.zero_()
    if os.environ.get("DATA") == "relu": x.clamp_(min=0)
    w = (torch.randn(9, Cout, Cin, device=dev) / (9 * Cin) ** 0.5).to(torch.bfloat16)
    out = torch.empty(B, S, S, Cout, dtype=torch.bfloat16, device=dev)
    bias = torch.zeros(Cout, device=dev)
    rows = ops.conv_stat_rows(ops.view(x), Cout, 3, 3, 1, 1)   # forward convs of the model run with the BN statistics epilogue
    stats = tuple(torch.empty(rows * Cout, device=dev) for _ in range(2)) + (torch.empty(rows, device=dev),)
    for _ in range(3):
        ops.conv2d(ops.view(x), w, bias, ops.view(out), 3, 3, 1, 1, stats=stats)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.conv2d(ops.view(x), w, bias, ops.view(out), 3, 3, 1, 1, stats=stats)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    fl = 2.0 * B * S * S * Cin * Cout * 9
    return us, fl / us / 1e6

for (B, S, Cout) in ((32, 256, 128), (32, 512, 64), (32, 64, 512)):
    for Cin in (64, 128, 256, 512, 1024):
        if Cin * S * S * B * 2 > 6e9: continue
        us, tf = bench(B, S, Cin, Cout)
        print(f"B={B} S={S} Cout={Cout} Cin={Cin:5d}: {us:9.1f} us  {tf:7.1f} TFLOP/s", flush=True)

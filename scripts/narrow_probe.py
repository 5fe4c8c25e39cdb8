"""Probe (GPU): data-gradient-like 3x3 launches whose K loop is short, wide (128-column) vs narrow (64-column, grid.y doubled) tiles.
Run twice: CVCS_HALO_NARROW_CIN=0 and =64 / =128."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvcs_amd import ops
dev = "cuda:0"
for (B, S, Cin, Cout) in ((32, 512, 64, 64), (32, 512, 128, 64), (32, 512, 64, 128), (32, 256, 128, 128), (32, 256, 256, 128), (32, 128, 256, 256), (32, 128, 512, 256), (32, 64, 512, 512)):
    x = torch.randn(B, S, S, Cin, device=dev).clamp_(min=0).to(torch.bfloat16)
    w = (torch.randn(9, Cout, Cin, device=dev) / (9 * Cin) ** 0.5).to(torch.bfloat16)
    out = torch.empty(B, S, S, Cout, dtype=torch.bfloat16, device=dev)
    res = []
    for rnd in range(3):
        for _ in range(3): ops.conv2d(ops.view(x), w, None, ops.view(out), 3, 3, 1, 1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): ops.conv2d(ops.view(x), w, None, ops.view(out), 3, 3, 1, 1)
        e.record(); torch.cuda.synchronize()
        res.append(s.elapsed_time(e) * 100)
    print("narrow_cin", os.environ.get("CVCS_HALO_NARROW_CIN", "0"), "waves", os.environ.get("CVCS_HALO_WAVES", "8"), "pipe", os.environ.get("CVCS_HALO_PIPE", "1"), "narrow3", os.environ.get("CVCS_HALO_NARROW3", "0"), f"S={S} {Cin}->{Cout}: " + " ".join(f"{r:.1f}" for r in res), "us", flush=True)

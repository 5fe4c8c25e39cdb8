"""Two halo-conv launches for counter collection: level-1 (64->64 @512, narrow kernel) and level-2 (128->128 @256, wide)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvcs_amd import ops
dev = "cuda:0"
for (B, S, Cin, Cout) in ((32, 512, 64, 64), (32, 256, 128, 128), (32, 64, 512, 512)):
    x = torch.randn(B, S, S, Cin, device=dev).clamp_(min=0).to(torch.bfloat16)
    w = (torch.randn(9, Cout, Cin, device=dev) / (9 * Cin) ** 0.5).to(torch.bfloat16)
    out = torch.empty(B, S, S, Cout, dtype=torch.bfloat16, device=dev)
    bias = torch.zeros(Cout, device=dev)
    rows = ops.conv_stat_rows(ops.view(x), Cout, 3, 3, 1, 1)
    stats = tuple(torch.empty(rows * Cout, device=dev) for _ in range(2)) + (torch.empty(rows, device=dev),)
    for _ in range(3):
        ops.conv2d(ops.view(x), w, bias, ops.view(out), 3, 3, 1, 1, stats=stats)
    torch.cuda.synchronize()

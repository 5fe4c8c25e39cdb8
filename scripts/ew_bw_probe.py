"""achieved HBM bandwidth of the element-wise passes at the headline step's shapes, next to a plain torch stream of the same size
(GPU box; knobs such as CVCS_BN_BWD_ROWS are read at load time: one process per setting)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cvcs_amd import ops

dev = "cuda:0"
SHAPES = [(32, 512, 512, 16), (32, 256, 256, 32), (32, 256, 256, 64), (32, 128, 128, 64), (32, 128, 128, 256), (32, 64, 64, 128), (32, 64, 64, 512), (32, 32, 32, 256)]


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


for B, H, W, C in SHAPES:
    M = B * H * W
    y = torch.randn(B, H, W, C, device=dev).to(torch.bfloat16)
    g = torch.randn(B, H, W, C, device=dev).to(torch.bfloat16)
    o = torch.empty_like(y)
    vec = [torch.rand(C, device=dev) + 0.5 for _ in range(6)]
    rows = ops.bn_bwd_rows(M)
    p = [torch.empty(rows * C, device=dev) for _ in range(3)]
    T = M * C * 2 / 1e6
    yv, gv, ov = ops.view(y), ops.view(g), ops.view(o)
    res = {}
    res["bn_act 1R1W"] = (timed(lambda: ops.bn_act(yv, vec[0], vec[1], True, ov)), 2)
    res["bn_bwd_reduce 2R"] = (timed(lambda: ops.bn_bwd_reduce(yv, gv, None, vec[0], vec[1], vec[2], vec[3], 0, p[0], p[1])), 2)
    res["bn_bwd_apply 2R1W"] = (timed(lambda: ops.bn_bwd_apply(yv, gv, None, vec[0], vec[1], vec[2], vec[3], vec[4], vec[5], 0, ov, p[2])), 3)
    res["relu_bwd_sum 3R1W"] = (timed(lambda: ops.relu_bwd_sum(yv, [(gv, False), (ov, False)], ov)), 4)
    res["torch copy 1R1W"] = (timed(lambda: o.copy_(y)), 2)
    res["torch add 2R1W"] = (timed(lambda: torch.add(y, g, out=o)), 3)
    res["torch sum 1R"] = (timed(lambda: y.view(torch.int16).sum()), 1)
    print(f"{B}x{H}x{W}x{C} ({T:.0f} MB per tensor): " + "  ".join(f"{k} {us:.1f}us {n * T / us:.2f}TB/s" for k, (us, n) in res.items()), flush=True)

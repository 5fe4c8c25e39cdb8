"""Probe (GPU): the four ConvTranspose2d(k2,s2) weight-gradient launches of a B=32, S=512 step.  CVCS_WGRAD_S2=0 = generic kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvcs_amd import ops
dev = "cuda:0"
B = 32
for (H, Cin) in ((32, 1024), (64, 512), (128, 256), (256, 128)):
    Cout = Cin // 2     # ConvTranspose Cin -> Cout; as a 2x2/s2 conv: x = grad of the upsampled map (Cout ch), dy = its input (Cin ch)
    x = torch.randn(B, 2 * H, 2 * H, Cout, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
    ws = torch.empty(ops.wgrad_workspace_floats(B, H, H, Cin, Cout, 2, 2, 2), device=dev)
    dw = torch.empty(Cin, Cout, 2, 2, device=dev)
    fn = lambda: ops.conv2d_wgrad(ops.view(x), ops.view(dy), dw, 2, 2, 2, 0, ws)
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 100
    fl = 2.0 * B * H * H * Cin * 4 * Cout
    print(f"s2={os.environ.get('CVCS_WGRAD_S2', '1')} H={H} {Cin}->{Cout}: {us:.1f} us ({fl / us / 1e6:.0f} TF) incl. reduce", flush=True)

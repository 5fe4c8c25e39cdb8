"""Probe (GPU): the eight ConvTranspose2d(k2,s2) launches of a B=32, S=512 U-Net step (four forward 1x1 + pixel shuffle, four
data gradients 2x2/s2), per-launch time.  CVCS_CONV_TAPS=0 routes them through the generic implicit-GEMM kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvcs_amd import ops
dev = "cuda:0"
def timeit(fn):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 100
B = 32
for (H, Cin) in ((32, 1024), (64, 512), (128, 256), (256, 128)):
    Cout = Cin // 2
    x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
    wf = (torch.randn(1, 4 * Cout, Cin, device=dev) / Cin ** 0.5).to(torch.bfloat16)
    wd = (torch.randn(4, Cin, Cout, device=dev) / Cin ** 0.5).to(torch.bfloat16)
    b4 = torch.zeros(4 * Cout, device=dev)
    out = torch.zeros(B, 2 * H, 2 * H, 2 * Cout, dtype=torch.bfloat16, device=dev)
    gy = torch.randn(B, 2 * H, 2 * H, Cout, device=dev).to(torch.bfloat16)
    gin = torch.empty(B, H, H, Cin, dtype=torch.bfloat16, device=dev)
    fl = 2.0 * B * H * H * Cin * 4 * Cout
    tf = timeit(lambda: ops.conv2d(ops.view(x), wf, b4, ops.View(out, Cout, Cout), 1, 1, pixel_shuffle=True))
    tb = timeit(lambda: ops.conv2d(ops.view(gy), wd, None, ops.view(gin), 2, 2, stride=2, pad=0))
    print(f"taps={os.environ.get('CVCS_CONV_TAPS', '1')} H={H} {Cin}->{Cout}: fwd {tf:.1f} us ({fl / tf / 1e6:.0f} TF)  dgrad {tb:.1f} us ({fl / tb / 1e6:.0f} TF)", flush=True)

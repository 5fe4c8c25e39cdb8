import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvcs_amd import ops
B, S, Cin, Cout = (int(v) for v in sys.argv[1:5])
dev = "cuda:0"
x = torch.randn(B, S, S, Cin, device=dev).to(torch.bfloat16)
dy = torch.randn(B, S, S, Cout, device=dev).to(torch.bfloat16)
dw = torch.empty(Cout, Cin, 3, 3, device=dev)
ws = torch.empty(ops.wgrad_workspace_floats(B, S, S, Cout, Cin, 3, 3, 1), device=dev)
for _ in range(3):
    ops.conv2d_wgrad(ops.view(x), ops.view(dy), dw, 3, 3, 1, 1, ws)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5):
    ops.conv2d_wgrad(ops.view(x), ops.view(dy), dw, 3, 3, 1, 1, ws)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / 5
print(f"wgrad B={B} S={S} Cin={Cin} Cout={Cout}: {us:.1f} us {2.0*B*S*S*Cin*Cout*9/us/1e6:.1f} TFLOP/s")

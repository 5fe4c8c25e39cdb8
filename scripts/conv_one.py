import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvcs_amd import ops
B, S, Cin, Cout = (int(v) for v in sys.argv[1:5])
dev = "cuda:0"
x = torch.randn(B, S, S, Cin, device=dev).to(torch.bfloat16)
w = (torch.randn(9, Cout, Cin, device=dev) / (9 * Cin) ** 0.5).to(torch.bfloat16)
out = torch.empty(B, S, S, Cout, dtype=torch.bfloat16, device=dev)
for _ in range(6):
    ops.conv2d(ops.view(x), w, None, ops.view(out), 3, 3, 1, 1)
torch.cuda.synchronize()

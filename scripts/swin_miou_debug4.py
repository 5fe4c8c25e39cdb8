"""running statistics accumulated over K train-mode forwards with FROZEN weights (no chaos): HIP vs the oracle from the same state"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import miou_parity as M
from oracle import swin_upernet_oracle as W
from oracle import unet_oracle as O

K, S = int(sys.argv[1]) if len(sys.argv) > 1 else 24, 128
torch.set_num_threads(16)
train = [O.synthetic_tiles(M.B, S, M.NC, seed=100 + i, structured=True) for i in range(8)]
_, _, net = M._setup("TSwin", "bf16", seed=3)
net.train()
p = {k: v.detach().float().cpu().clone() for k, v in net.state_dict().items() if k in W.init_params(M.NC, 0)}
for s in range(K):
    img, _ = train[s % len(train)]
    with torch.no_grad():
        net(img.to(M.DEV), None)
        W.forward(p, img.float(), train=True)
    if s in (0, 1, 3, 7, K - 1):
        torch.cuda.synchronize()
        sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
        rows = sorted(((sd[k] - p[k]).abs().max().item() / (p[k].abs().max().item() + 1e-12), k.replace("decode_head.", "")) for k in p if "running_" in k)[-3:]
        print(f"after {s + 1:2d} forwards: largest relative gaps {[(round(a, 4), k) for a, k in rows]}", flush=True)

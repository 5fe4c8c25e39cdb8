"""is a Swin-T + UPerNet training run bitwise reproducible?  (two identical runs in one process: loss trajectory and a parameter checksum)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import miou_parity as M
from cvcs_amd import utils
from oracle import unet_oracle as O

steps, S = int(sys.argv[1]) if len(sys.argv) > 1 else 30, 128
train = [O.synthetic_tiles(M.B, S, M.NC, seed=100 + i, structured=True) for i in range(8)]
out = []
for rep in range(2):
    _, _, net = M._setup("TSwin", "bf16", seed=3)
    crit = utils.CrossEntropyLoss(ignore_index=0)
    opt, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 1}, net)
    net.train()
    losses = []
    for s in range(steps):
        img, lab = train[s % len(train)]
        loss = crit(net(img.to(M.DEV), None), lab.to(M.DEV))
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(loss.item())
    sd = net.state_dict()
    out.append((losses, {k: v.detach().double().sum().item() for k, v in sd.items() if v.dtype.is_floating_point}))
same_loss = out[0][0] == out[1][0]
diff = [k for k in out[0][1] if out[0][1][k] != out[1][1][k]]
print(f"losses identical: {same_loss}; first differing step: {next((i for i, (a, b) in enumerate(zip(out[0][0], out[1][0])) if a != b), None)}")
print(f"{len(diff)} of {len(out[0][1])} tensors differ:", diff[:12])

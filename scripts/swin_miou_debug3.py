"""running-statistics increments of ONE step at a TRAINED state: HIP vs the oracle started from the HIP state (same weights, same running
statistics, same batch): rm' - rm = 0.1 (batch mean - rm) must agree layer by layer"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import miou_parity as M
from cvcs_amd import utils
from oracle import swin_upernet_oracle as W
from oracle import unet_oracle as O

seed, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3, int(sys.argv[2]) if len(sys.argv) > 2 else 100
S, model = 128, "TSwin"
torch.set_num_threads(16)
train = [O.synthetic_tiles(M.B, S, M.NC, seed=100 + i + 1000 * seed, structured=True) for i in range(8)]
_, _, net = M._setup(model, "bf16", seed=3 + seed)
crit = utils.CrossEntropyLoss(ignore_index=0)
opt, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 1}, net)
net.train()
for s in range(steps):
    img, lab = train[s % len(train)]
    loss = crit(net(img.to(M.DEV), None), lab.to(M.DEV))
    opt.zero_grad(); loss.backward(); opt.step()
before = {k: v.detach().float().cpu().clone() for k, v in net.state_dict().items()}
img, lab = train[steps % len(train)]
# the oracle from the HIP state: one forward in train mode updates its running statistics in place
p = {k: v.clone() for k, v in before.items() if k in W.init_params(M.NC, 0)}
with torch.no_grad():
    W.forward(p, img.float(), train=True)
loss = crit(net(img.to(M.DEV), None), lab.to(M.DEV))
torch.cuda.synchronize()
after = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
rows = []
for k in p:
    if "running_" not in k:
        continue
    inc_o, inc_h = p[k] - before[k], after[k] - before[k]
    rel = (inc_h - inc_o).abs().max().item() / (inc_o.abs().max().item() + 1e-12)
    rows.append((rel, k, inc_o.abs().max().item(), inc_h.abs().max().item()))
for rel, k, a, b in sorted(rows, reverse=True):
    print(f"{k:60s} increment max |oracle| {a:.4e} |HIP| {b:.4e}  max difference / max |oracle increment| {rel:.3e}")

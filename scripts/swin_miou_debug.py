"""diagnostic for a held-out mIoU gap of the Swin-T + UPerNet run: trains HIP and oracle side by side (loss trajectories), then evaluates the
HIP network's OWN trained state with the oracle's eval forward - separates 'training went elsewhere' from 'the eval path disagrees'"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import miou_parity as M
from cvcs_amd import utils
from oracle import unet_oracle as O

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
S, model, seed = 128, "TSwin", 0
torch.set_num_threads(16)
train = [O.synthetic_tiles(M.B, S, M.NC, seed=100 + i + 1000 * seed, structured=True) for i in range(8)]
held = [O.synthetic_tiles(M.B, S, M.NC, seed=900 + i, structured=True) for i in range(4)]
tr, fwd, net = M._setup(model, "bf16", seed=3 + seed)
crit = utils.CrossEntropyLoss(ignore_index=0)
opt, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 1}, net)
net.train()
for s in range(steps):
    img, lab = train[s % len(train)]
    lo = tr.step(img, lab)[0]
    loss = crit(net(img.to(M.DEV), None), lab.to(M.DEV))
    opt.zero_grad(); loss.backward(); opt.step()
    if s % 5 == 0 or s == steps - 1:
        print(f"step {s:3d}  oracle loss {lo:.4f}  hip loss {loss.item():.4f}", flush=True)
net.eval()
sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
p_o = {k: v.detach() for k, v in tr.p.items()}
worst = sorted(((sd[k] - p_o[k]).abs().max().item() / (p_o[k].abs().max().item() + 1e-12), k) for k in p_o if k in sd and sd[k].shape == p_o[k].shape)[-8:]
print("largest relative parameter differences HIP vs oracle:", [(round(a, 4), k) for a, k in worst])
with torch.no_grad():
    for img, lab in held[:2]:
        lh = net(img.to(M.DEV), None).float().cpu()
        lo_own = fwd({k: sd.get(k, v) for k, v in p_o.items()}, img.float())      # the oracle's eval forward on the HIP network's state
        lo_or = fwd(p_o, img.float())
        agree_own = (lh.argmax(1) == lo_own.argmax(1)).float().mean().item()
        agree_or = (lh.argmax(1) == lo_or.argmax(1)).float().mean().item()
        err = (lh - lo_own).abs().max().item() / lo_own.abs().max().item()
        print(f"eval: HIP logits vs oracle forward on HIP's own state: max err {err:.3e} of max|logit|, label agreement {agree_own:.4f}; vs the oracle's own run {agree_or:.4f}")

"""where does a held-out mIoU gap of the Swin-T + UPerNet run come from?  After training HIP and oracle side by side: the oracle's f32 forward on
(a) the oracle's state, (b) the HIP state, each in eval mode (running statistics) and in train mode (batch statistics of the held-out batch), and
the same with single groups of the HIP state swapped into the oracle's (running statistics only / decode head only / backbone only)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import miou_parity as M
from cvcs_amd import utils
from oracle import swin_upernet_oracle as W
from oracle import unet_oracle as O

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps, S, model = 100, 128, "TSwin"
torch.set_num_threads(16)
train = [O.synthetic_tiles(M.B, S, M.NC, seed=100 + i + 1000 * seed, structured=True) for i in range(8)]
held = [O.synthetic_tiles(M.B, S, M.NC, seed=900 + i, structured=True) for i in range(4)]
tr, fwd, net = M._setup(model, "bf16", seed=3 + seed)
crit = utils.CrossEntropyLoss(ignore_index=0)
opt, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 1}, net)
net.train()
for s in range(steps):
    img, lab = train[s % len(train)]
    tr.step(img, lab)
    loss = crit(net(img.to(M.DEV), None), lab.to(M.DEV))
    opt.zero_grad(); loss.backward(); opt.step()
net.eval()
sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
p_o = {k: v.detach().clone() for k, v in tr.p.items()}
p_h = {k: sd.get(k, v).clone() for k, v in p_o.items()}


def miou(p, train_mode):
    with torch.no_grad():
        conf = sum(O.confusion_matrix(O.predict_labels(W.forward({k: v.clone() for k, v in p.items()}, img.float(), train=train_mode)).numpy(), lab.numpy(), 16, ignore_index=0)
                   for img, lab in held)
    m = O.metrics(conf)
    return 100 * m["mIoU"]


def mix(pred):
    return {k: (p_h[k] if pred(k) else p_o[k]) for k in p_o}


print(f"seed {seed}: oracle state eval {miou(p_o, False):.3f} train-mode {miou(p_o, True):.3f} | HIP state eval {miou(p_h, False):.3f} train-mode {miou(p_h, True):.3f}")
print(f"oracle state with HIP's running statistics: {miou(mix(lambda k: 'running_' in k), False):.3f};  HIP state with the oracle's running statistics: {miou(mix(lambda k: 'running_' not in k), False):.3f}")
print(f"oracle state with HIP's decode head: {miou(mix(lambda k: k.startswith('decode_head')), False):.3f};  with HIP's backbone: {miou(mix(lambda k: not k.startswith('decode_head')), False):.3f}")
for grp in ("psp_modules", "bottleneck", "lateral_convs", "fpn_convs", "fpn_bottleneck", "classifier"):
    print(f"oracle state with HIP's decode_head.{grp}: {miou(mix(lambda k: k.startswith('decode_head.' + grp)), False):.3f}")


def refreshed(p, passes=4):
    """the same weights with running statistics re-accumulated by train-mode forwards over the training tiles (no weight update)"""
    q = {k: v.clone() for k, v in p.items()}
    with torch.no_grad():
        for _ in range(passes):
            for img, _lab in train:
                W.forward(q, img.float(), train=True)
    return q


for name, p in (("oracle", p_o), ("HIP", p_h)):
    q = refreshed(p)
    dev = sorted(((q[k] - p[k]).abs().max().item() / (q[k].abs().max().item() + 1e-12), k) for k in p if "running_" in k)[-4:]
    print(f"{name} state with re-accumulated running statistics: eval {miou(q, False):.3f}; accumulated-during-training vs re-accumulated, largest relative gaps: {[(round(a, 3), k.replace('decode_head.', '')) for a, k in dev]}")

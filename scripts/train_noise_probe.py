"""Probe (GPU): 3 SGD2 steps + eval forward; HIP fp32 and torch-CPU fp32 against a float64 oracle run."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvcs_amd import nets, utils
from oracle import unet_oracle as O

NC, B, S = 5, 2, 32
variant, opt = sys.argv[1] if len(sys.argv) > 1 else "Unetv2", sys.argv[2] if len(sys.argv) > 2 else "SGD2"
img, lab = O.synthetic_tiles(B, S, NC, seed=11)


def run_oracle(dtype):
    p = {k: v.to(dtype) for k, v in O.init_params(variant, NC, seed=3).items()}
    tr = O.OracleTrainer(variant, NC, opt=opt, epochs=4, ignore_index=0, params=p)
    losses = []
    orig = O.unet_forward
    for s in range(3):
        x = img.to(dtype)
        logits = O.unet_forward(tr.p, x, variant, train=True)
        loss = O.cross_entropy(logits, lab.long(), None, 0)
        grads = torch.autograd.grad(loss, [tr.p[k] for k in tr.trainable])
        tr.nstep += 1
        lr = tr.lr()
        with torch.no_grad():
            for k, g in zip(tr.trainable, grads):
                q = tr.p[k]
                if tr.cfg["kind"] == "sgd":
                    newp, buf = O.sgd_step(q, g, tr.state.get(k), lr, tr.cfg["momentum"], tr.cfg["weight_decay"], k not in tr.state)
                    tr.state[k] = buf
                else:
                    m, v = tr.state.get(k, (torch.zeros_like(q), torch.zeros_like(q)))
                    newp, m, v = O.adam_step(q, g, m, v, tr.nstep, lr, tr.cfg["betas"], tr.cfg["eps"], 0.0)
                    tr.state[k] = (m, v)
                q.copy_(newp)
        losses.append(loss.item())
    with torch.no_grad():
        ev = O.unet_forward(tr.p, img.to(dtype), variant, train=False)
    return losses, ev, tr.p


l64, e64, p64 = run_oracle(torch.float64)
l32, e32, p32 = run_oracle(torch.float32)
cls = nets.Urnetv2 if variant == "Unetv2" else nets.Urnet
net = cls(NC, "fp32")
net.load_state_dict(O.init_params(variant, NC, seed=3), strict=False)
net = net.to("cuda:0")
crit = utils.CrossEntropyLoss(ignore_index=0)
optim, _ = utils.load_optimizer({"opt": opt, "epochs": 4}, net)
net.train()
lh = []
for s in range(3):
    loss = crit(net(img.to("cuda:0").float()), lab.to("cuda:0"))
    optim.zero_grad(); loss.backward(); optim.step()
    lh.append(loss.item())
net.eval()
with torch.no_grad():
    eh = net(img.to("cuda:0").float()).cpu()
print("losses f64", l64, "\n       t32", l32, "\n       hip", lh)
sc = e64.abs().max().item()
print("eval logits scale %.3e: torch32 rel err %.2e, hip rel err %.2e" % (sc, (e32.double() - e64).abs().max() / sc, (eh.double() - e64).abs().max() / sc))
sd = net.state_dict()
for k in ["encode1.0.layer.1.running_mean", "encode1.0.layer.1.running_var", "encode5.2.layer.1.running_var", "decode_forward4.0.layer.5.running_var", "decode_forward4.0.layer.5.running_mean"]:
    r = p64[k]
    print(k, "t32 err %.2e hip err %.2e" % ((p32[k].double() - r).abs().max() / r.abs().max(), (sd[k].cpu().double() - r).abs().max() / r.abs().max()))

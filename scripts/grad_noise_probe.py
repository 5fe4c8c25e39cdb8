"""Probe (GPU): per-parameter gradient error of the HIP fp32 path and of torch-CPU fp32 against an fp64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvcs_amd import nets, utils
from oracle import unet_oracle as O

NC, B, S = 5, 2, 32
variant = sys.argv[1] if len(sys.argv) > 1 else "Unetv2"
img, lab = O.synthetic_tiles(B, S, NC, seed=11)


def oracle_grads(dtype):
    p = {k: v.to(dtype) for k, v in O.init_params(variant, NC, seed=3).items()}
    tr = [k for k in p if not O.is_buffer(k)]
    for k in tr:
        p[k].requires_grad_(True)
    logits = O.unet_forward(p, img.to(dtype), variant, train=True)
    loss = O.cross_entropy(logits, lab.long(), None, 0)
    gr = torch.autograd.grad(loss, [p[k] for k in tr])
    return dict(zip(tr, gr)), logits.detach(), loss.item()


g64, l64, loss64 = oracle_grads(torch.float64)
g32, l32, loss32 = oracle_grads(torch.float32)
net = (nets.Urnetv2 if variant == "Unetv2" else nets.Urnet)(NC, "fp32")
net.load_state_dict(O.init_params(variant, NC, seed=3), strict=False)
net = net.to("cuda:0")
crit = utils.CrossEntropyLoss(ignore_index=0)
net.train()
pred = net(img.to("cuda:0").float())
loss = crit(pred, lab.to("cuda:0"))
loss.backward()
print("loss f64 %.8f f32 %.8f hip %.8f" % (loss64, loss32, loss.item()))
print("logits: torch32 err %.2e hip err %.2e" % ((l32.double() - l64).abs().max() / l64.abs().max(), (pred.detach().cpu().double() - l64).abs().max() / l64.abs().max()))
for k, p in net.named_parameters():
    ref = g64[k]
    sc = ref.abs().max().item() + 1e-30
    e_t = (g32[k].double() - ref).abs().max().item() / sc
    e_h = (p.grad.cpu().double() - ref).abs().max().item() / sc
    print(f"{k:40s} max|g| {sc:.3e}  torch32 rel err {e_t:.2e}  hip rel err {e_h:.2e}")

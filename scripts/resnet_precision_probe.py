"""where does the f32 HIP path of the ResNet-UNet lose precision against the float64 oracle?  (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvcs_amd import nets, utils
from oracle import resnet_unet_oracle as R, unet_oracle as O

arch, B, S, NC = sys.argv[1] if len(sys.argv) > 1 else "resnet18", 2, int(sys.argv[2]) if len(sys.argv) > 2 else 64, 5
net = getattr(nets, "Resnet18Unet" if arch == "resnet18" else "Resnet50Unet")(NC, "fp32")
p = R.init_params(arch, NC, seed=3)
net.load_state_dict(p, strict=False)
net = net.to("cuda:0").train()
img, lab = O.synthetic_tiles(B, S, NC, seed=5, structured=True)
crit = utils.CrossEntropyLoss(ignore_index=0)
logits = net(img.cuda(), None)
loss = crit(logits, lab.cuda())
loss.backward()
torch.cuda.synchronize()
eng = net._engine
p64 = {k: v.double().clone() for k, v in p.items()}
names = [k for k in p64 if not R.is_buffer(k)]
for k in names:
    p64[k].requires_grad_(True)
# float64 and float32 CPU forward with every conv output captured
import torch.nn.functional as F
caps = {}
orig = F.conv2d
def run(pp, x):
    caps.clear()
    order = []
    def hook(inp, w, b=None, stride=1, padding=0, *a, **k):
        y = orig(inp, w, b, stride, padding, *a, **k)
        order.append(y)
        return y
    F.conv2d = hook
    try:
        out = R.forward(pp, x, arch, train=True)
    finally:
        F.conv2d = orig
    return out, order
out64, ys64 = run(p64, img.double())
out32, ys32 = run({k: v.clone() for k, v in p.items()}, img.float())
convs = [n for n, _, _, _, _, _ in R.conv_table(arch, NC, S)]
# execution order of the oracle: per block conv1, conv2, (conv3), downsample -> same as conv_table
rl2 = lambda a, b: ((a.double() - b).norm() / b.norm()).item()
print("conv outputs: rel-L2 vs float64   HIP f32 | torch-CPU f32")
for name, y64, y32 in zip(convs, ys64, ys32):
    if name not in eng.units:
        continue
    u, _ = eng.units[name]
    got = u.y.torch().float().cpu().permute(0, 3, 1, 2)
    print(f"  {name:34s} {rl2(got, y64.detach()):.2e} | {rl2(y32, y64.detach()):.2e}")
print("logits", rl2(logits.detach().cpu(), out64.detach()), "|", rl2(out32, out64.detach()))
l64 = O.cross_entropy(out64, lab.long(), None, 0)
g64 = dict(zip(names, torch.autograd.grad(l64, [p64[k] for k in names])))
print("gradients: rel-L2 vs float64 (HIP f32), forward order")
for k, prm in net.named_parameters():
    print(f"  {k:40s} {rl2(prm.grad.cpu(), g64[k]):.2e}")

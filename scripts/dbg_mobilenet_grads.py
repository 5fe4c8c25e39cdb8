import sys
import torch
sys.path.insert(0, "tests")
from cvcs_amd import nets, ops, utils
from oracle import mobilenet_oracle as M, unet_oracle as O
DEV = "cuda:0"
B, S, NC = 4, 64, 5
net = nets.DeepLabV3MobileNet(NC, "fp32", aspp_dropout=0.0)
sd = {net._to_reference(k): v for k, v in M.init_params(NC, seed=3).items()}
net.load_state_dict(sd, strict=False)
net = net.to(DEV)
img, lab = O.synthetic_tiles(B, S, NC, seed=5, structured=True)
crit = utils.CrossEntropyLoss(ignore_index=0)
net.train()
p = {net._from_reference(k): v.detach().cpu().clone() for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")}
names = [k for k in p if not M.is_buffer(k)]
for k in names: p[k].requires_grad_(True)
logits, acts = M.forward(p, img.float(), train=True, return_acts=True)
for t in (acts["aspp"], acts["last"], acts["features"]): t.retain_grad()
loss = O.cross_entropy(logits, lab.long(), None, 0)
loss.backward()
out = net(img.to(DEV), None)
l2 = crit(out, lab.to(DEV))
l2.backward()
torch.cuda.synchronize()
eng = net._engine
def buf(name):
    for (n, shape, dt), t in eng._bufs.items():
        if n == name: return t
    raise KeyError(name)
r = lambda a, b: ((a - b).norm() / b.norm()).item()
nchw = lambda t: t.float().cpu().permute(0, 3, 1, 2)
print("loss", loss.item(), l2.item())
print("a fwd", r(nchw(buf("aspp.out")), acts["aspp"].detach()), "d fwd", r(nchw(buf("head.a1")), acts["last"].detach()), "f5 fwd", r(nchw(buf("encoder.16.a")), acts["features"].detach()))
print("g_a", r(nchw(buf("head.g_aspp")), acts["aspp"].grad), "g_last", r(nchw(buf("head.gx")), acts["last"].grad))
ga, a = acts["aspp"].grad, acts["aspp"].detach()
print("oracle dbeta from its own g_a", r((ga * (a > 0)).sum(dim=(0, 2, 3)), p["aspp.project.1.bias"].grad))
g_eng, a_eng = nchw(buf("head.g_aspp")), nchw(buf("aspp.out"))
print("engine dbeta vs sum(g*mask) of its own buffers", r(eng.G["aspp.project.1.bias"].cpu(), (g_eng * (a_eng > 0)).sum(dim=(0, 2, 3))))
print("engine dbeta vs oracle", r(eng.G["aspp.project.1.bias"].cpu(), p["aspp.project.1.bias"].grad))
print("mask mismatches", int(((a_eng > 0) != (a > 0)).sum()))

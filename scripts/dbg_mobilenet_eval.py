import torch
from cvcs_amd import nets, ops, utils
from oracle import mobilenet_oracle as M, unet_oracle as O
DEV = "cuda:0"
B, S, NC = 3, 128, 16
net = nets.DeepLabV3MobileNet(NC, "fp32", aspp_dropout=0.0)
net.load_state_dict({net._to_reference(k): v for k, v in M.init_params(NC, seed=3).items()}, strict=False)
net = net.to(DEV)
img, lab = O.synthetic_tiles(B, S, NC, seed=5, structured=True)
crit = utils.CrossEntropyLoss(ignore_index=0)
optim, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 20}, net)
net.train()
for _ in range(3):
    loss = crit(net(img.to(DEV), None), lab.to(DEV)); optim.zero_grad(); loss.backward(); optim.step()
net.eval()
p = {net._from_reference(k): v.detach().cpu().clone() for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")}
with torch.no_grad():
    ev = net(img.to(DEV), None).cpu()
    want, acts = M.forward(p, img.float(), train=False, return_acts=True)
    want64, acts64 = M.forward({k: v.double() for k, v in p.items()}, img.double(), train=False, return_acts=True)
eng = net._engine
def buf(name):
    c = [t for (n, shape, dt), t in eng._bufs.items() if n == name]
    return c
nchw = lambda t: t.float().cpu().permute(0, 3, 1, 2)
r = lambda a, b: ((a - b).norm() / b.norm()).item()
print("logits HIP vs f64", r(ev.double(), want64), "torch f32 vs f64", r(want.double(), want64))
names = ["encoder.0.a"] + [f"encoder.{i + 1}.block.out" for i in range(15)] + ["encoder.16.a"]
for i, n in enumerate(names):
    c = buf(n)
    o64 = acts64["blocks"][i]
    if not c:
        print(n, "no buffer (eval writes the projection output in place)", "torch", r(acts["blocks"][i].double(), o64)); continue
    t = [x for x in c if x.shape[1] == o64.shape[2]][-1]
    print(n, "HIP", f"{r(nchw(t)[:, :o64.shape[1]].double(), o64):.2e}", "torch f32", f"{r(acts['blocks'][i].double(), o64):.2e}", "max", f"{o64.abs().max().item():.2e}")

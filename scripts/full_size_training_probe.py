"""Numerical sanity at the bench shape: 80 SGD2 steps on structured synthetic tiles, batch 32 x 512x512, bf16 path.
Prints the loss curve and the training-set mIoU (fused argmax + confusion) at the end."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvcs_amd import dataset, utils

dev = "cuda:0"
NC = 16
cfg = {"net": "Unetv2", "num_classes": NC - 1, "precision": "bf16", "loss": "CEL", "ignore_background": True, "opt": "SGD2", "epochs": 20}
torch.manual_seed(0)
net = utils.load_network(cfg, dev); crit = utils.load_loss(cfg, dev); opt, _ = utils.load_optimizer(cfg, net)
ld = dataset.make_loader("synthetic:4:16", 4, 512, NC, seed=0)
tiles = list(ld.get_iterable_chunk(0))
x = torch.stack([t[0] for t in tiles]).to(dev)
y = torch.stack([t[1] for t in tiles]).to(dev)
batches = [(x[i:i + 32], y[i:i + 32]) for i in range(0, 64, 32)]
net.train()
t0 = time.time()
for s in range(80):
    xb, yb = batches[s % 2]
    loss = crit(net(xb, None), yb)
    opt.zero_grad(); loss.backward(); opt.step()
    if s % 10 == 0 or s == 79:
        print(f"step {s:3d} loss {loss.item():.4f}", flush=True)
torch.cuda.synchronize()
print(f"80 steps in {time.time() - t0:.1f} s")
flat = utils.ConfusionMatrix(num_classes=16, ignore_index=0, device=dev)
net.eval()
for xb, yb in batches:
    flat.update_from_labels(net.predict_labels(xb), yb)
m = utils.print_metrics(flat.compute(), silent=True)
print(f"training-set mIoU {100 * m['mIoU']:.2f}  OA {100 * m['oa_score']:.2f}")

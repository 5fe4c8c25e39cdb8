"""How launch-bound is a step at the reference's default shape (batch 10 x 224x224, configs/train/server.yaml)?  Eager step
time vs the same step replayed from a captured HIP graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvcs_amd import utils

dev = "cuda:0"
B, S, NC = (int(v) for v in (sys.argv[1:4] + ["10", "224", "16"][len(sys.argv) - 1:]))
cfg = {"net": "Unetv2", "num_classes": NC - 1, "precision": "bf16", "loss": "CEL", "ignore_background": True, "opt": "SGD2", "epochs": 20}
torch.manual_seed(0)
net = utils.load_network(cfg, dev); crit = utils.load_loss(cfg, dev); opt, _ = utils.load_optimizer(cfg, net)
g = torch.Generator().manual_seed(1)
x = torch.randint(0, 256, (B, 3, S, S), dtype=torch.uint8, generator=g).to(dev)
y = torch.randint(0, NC, (B, S, S), dtype=torch.uint8, generator=g).to(dev)
net.train()

def step():
    loss = crit(net(x, None), y)
    opt.zero_grad(); loss.backward(); opt.step()
    return loss

for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 30
for _ in range(n): step()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / n
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print(f"eager: {1e3 * eager:.2f} ms/step  ({B / eager:.0f} tiles/s) at B={B} S={S}")
try:
    gph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): step()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(gph):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): gph.replay()
    torch.cuda.synchronize()
    gt = (time.perf_counter() - t0) / n
    print(f"graph: {1e3 * gt:.2f} ms/step  ({B / gt:.0f} tiles/s); loss {loss.item():.5f}")
except Exception as ex:
    print("graph capture failed:", repr(ex)[:300])

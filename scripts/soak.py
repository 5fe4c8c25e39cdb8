"""Soak run (GPU): 240 bf16 train steps of ResNet50-UNet, Swin-T + UPerNet and DeepLabV3-ResNet101 (16 x 256 x 256): parameters stay finite,
the loss falls and the allocated device memory does not move after the plans are recorded.  usage: python scripts/soak.py"""
import sys, torch, time
sys.path.insert(0, '.')
from cvcs_amd import utils
for name in ("Resnet50Unet", "TSwin", "Resnet101"):
    torch.manual_seed(0)
    net = utils.load_network({"net": name, "num_classes": 15, "precision": "bf16"}, "cuda:0")
    crit = utils.load_loss({"num_classes": 15, "loss": "CEL", "ignore_background": True}, "cuda:0")
    opt, sched = utils.load_optimizer({"opt": "SGD2", "epochs": 20}, net)
    g = torch.Generator().manual_seed(1)
    B, S = 16, 256
    imgs = [torch.randint(0, 256, (B, 3, S, S), dtype=torch.uint8, generator=g).cuda() for _ in range(4)]
    labs = [(im.float().mean(1) // 17).clamp(0, 15).to(torch.uint8) for im in imgs]     # learnable: label = brightness band
    net.train()
    mem0 = None
    t0 = time.time()
    for step in range(240):
        loss = crit(net(imgs[step % 4], None), labs[step % 4])
        opt.zero_grad(); loss.backward(); opt.step()
        if step % 60 == 0 or step == 239:
            torch.cuda.synchronize()
            m = torch.cuda.memory_allocated() / 2**20
            mem0 = mem0 or m if step else None
            print(name, step, f"loss {loss.item():.4f}", f"alloc {m:.0f} MiB", flush=True)
        if step == 60: mem0 = torch.cuda.memory_allocated()
    assert torch.isfinite(net.flat_parameters()[0]).all()
    assert torch.cuda.memory_allocated() == mem0, (torch.cuda.memory_allocated(), mem0)
    print(name, "ok", f"{time.time() - t0:.1f}s", flush=True)
    del net, opt, crit
    torch.cuda.empty_cache()

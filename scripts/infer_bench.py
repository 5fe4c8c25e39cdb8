"""Inference throughput of the hot path (not the headline metric): eval-mode forward + fused argmax on resident u8 tiles,
and the full border-corrected segment_image pipeline on one synthetic 6800 x 7200 image (the GID-15 image size)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvcs_amd import ops, utils
from cvcs_amd.inference import segment_image

dev = "cuda:0"
net = utils.load_network({"net": "Unetv2", "num_classes": 15, "precision": "bf16"}, dev)
net.eval()
B, S = 32, 512
g = torch.Generator().manual_seed(0)
x = torch.randint(0, 256, (B, 3, S, S), dtype=torch.uint8, generator=g).to(dev)
labels = torch.empty((B, S, S), dtype=torch.uint8, device=dev)
with torch.no_grad():
    for _ in range(3):
        ops.argmax_confusion(net(x, None), labels=labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        ops.argmax_confusion(net(x, None), labels=labels)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"eval forward + argmax, batch {B} x {S}x{S} bf16: {B * n / dt:.1f} tiles/s ({1e3 * dt / n:.2f} ms per batch)")
img = torch.randint(0, 256, (3, 6800, 7200), dtype=torch.uint8, generator=g).to(dev)
for p, bc in ((224, 256), (448, 512)):
    segment_image(net, img, p, bc, batch=32, want_rgb=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rgb, lab = segment_image(net, img, p, bc, batch=32, want_rgb=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nt = (6800 // p) * (7200 // p)
    print(f"segment_image 6800x7200, patch {p}, border_correction {bc}: {dt:.3f} s for {nt} tiles ({nt / dt:.0f} tiles/s, {rgb.shape[0]}x{rgb.shape[1]} px map)")

"""fp8 (block-scaled MFMA) against bf16 3x3 halo convolutions on the layer shapes of ResNet50-UNet that qualify for fp8
(channel counts multiples of 128): hot back-to-back launches, HIP events.  usage: python scripts/fp8_conv_bench.py [tile]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvcs_amd import ops, _lib

DEV = "cuda:0"
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = 32 if S == 512 else 8
SHAPES = [  # (H, Cin, Cout, name)
    (S // 8, 128, 128, "enc layer2 3x3"), (S // 16, 256, 256, "enc layer3 3x3"), (S // 32, 512, 512, "enc layer4 3x3"),
    (S // 16, 3072, 256, "dec0 conv1"), (S // 16, 256, 256, "dec0 conv2"), (S // 8, 768, 128, "dec1 conv1"), (S // 8, 128, 128, "dec1 conv2"),
]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3   # us


g = torch.Generator().manual_seed(0)
for H, Cin, Cout, name in SHAPES:
    x = torch.randn(B, H, H, Cin, generator=g).bfloat16().to(DEV)
    w = (torch.randn(9, Cout, Cin, generator=g) * 0.05).bfloat16().to(DEV)
    y = torch.empty(B, H, H, Cout, dtype=torch.bfloat16, device=DEV)
    rows = ops.fp8_stat_rows(ops.view(x))
    stats = tuple(torch.zeros(n, dtype=torch.float32, device=DEV) for n in (rows * Cout, rows * Cout, rows))
    slots = ops.Fp8Slots(DEV)
    ix, iw = slots.slot("x", _lib.E4M3), slots.slot("w", _lib.E4M3)
    x8, w8 = torch.zeros_like(x, dtype=torch.uint8), torch.zeros_like(w, dtype=torch.uint8)
    ops.quantize_fp8(ops.view(x), ops.view(x8), _lib.E4M3, slots, ix)
    ops.quantize_fp8(ops.View(w.view(1, 9 * Cout, 1, Cin), 0, Cin), ops.View(w8.view(1, 9 * Cout, 1, Cin), 0, Cin), _lib.E4M3, slots, iw)
    fl = 2.0 * B * H * H * Cin * Cout * 9
    t8 = timeit(lambda: ops.conv3x3_fp8(ops.view(x8), _lib.E4M3, w8, ops.view(y), slots, ix, iw, stats=stats))
    t16 = timeit(lambda: ops.conv2d(ops.view(x), w, None, ops.view(y), 3, 3, 1, 1, stats=stats))
    tq = timeit(lambda: ops.quantize_fp8(ops.view(x), ops.view(x8), _lib.E4M3, slots, ix))
    print(f"{name:16s} B{B} {H}x{H} {Cin:4d}->{Cout:3d}: fp8 {t8:7.1f} us {fl / t8 / 1e6:7.1f} TF | bf16 {t16:7.1f} us {fl / t16 / 1e6:7.1f} TF | quantize x {tq:6.1f} us "
          f"({x.numel() * 3 / tq / 1e6:.2f} TB/s)", flush=True)

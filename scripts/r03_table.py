"""markdown table of the round-3 bench lines in profiles/ """
import glob
import json
import os

ROWS = [("resnet50unet_b32_s512_bf16", "**ResNet50-UNet (headline, cfg 2)**"), ("resnet50unet_b32_s512_fp8", "ResNet50-UNet, mixed bf16 / fp8 convs (cfg 5's precision on the headline shape)"),
        ("deeplabv3plus_b32_s512_bf16", "DeepLabV3+ ResNet-50 (cfg 3)"), ("swintupernet_b32_s512_bf16", "Swin-T + UPerNet (cfg 4)"),
        ("swintupernet_b32_s512_fp8", "Swin-T + UPerNet, mixed bf16 / fp8 (the UPerNet head's five 3x3 convs on the fp8 kernel)"),
        ("segformermod_b32_s512_bf16", "SegformerMod (reference factory name; MiT-b0, new)"),
        ("segformermod_b3_b32_s512_bf16", "SegformerMod with the MiT-b3 encoder (the checkpoint architecture the reference's default fetches)"), ("unetv2_b32_s512_bf16", "Unetv2 (the reference's own U-Net)"), ("unetv2_b32_s512_fp8", "Unetv2, mixed bf16 / fp8 (levels 2-5 and their decoder stages on the fp8 kernel)"),
        ("resnet101_b32_s512_bf16", "DeepLabV3-ResNet101 (the reference's `Resnet101`)"),
        ("cfg5_resnet50unet_b8_s1024_nc21_fp8", "cfg 5 as named: ResNet50-UNet, 1024^2 tiles, 21 classes, batch 8, mixed bf16 / fp8"),
        ("cfg5_resnet50unet_b8_s1024_nc21_bf16", "cfg 5 shape in bf16"), ("cfg1_resnet18unet_b8_s256_nc4_bf16", "cfg 1 shape: ResNet18-UNet, 256^2 tiles, 4 classes, batch 8")]
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
print("| workload (full train step) | tiles/s | ms/step | `roofline` = dominant family (time, achieved, frac of peak) | MFMA share of step | CPU oracle tiles/s (threads) |")
print("|---|---|---|---|---|---|")
for tag, name in ROWS:
    f = os.path.join(here, f"r03_bench_{tag}.json")
    if not os.path.isfile(f):
        continue
    d = json.load(open(f))
    r = d.get("roofline", {})
    cb = d.get("cpu_baseline", {})
    fam = r.get("kernel", "?").split(" (")[0]
    print(f"| {name} | {d['value']:.0f} | {d['ms_per_step']:.1f} | {fam}: {r.get('ms_per_step', 0):.1f} ms, {r.get('achieved', 0):.0f} {r.get('unit', '')} = {r.get('frac', 0):.2f} | "
          f"{d.get('mfma_kernels_share_of_step', '-')} | {cb.get('value', '-')} ({cb.get('cores', '-')}) |")

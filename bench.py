#!/usr/bin/env python3
"""bench.py - tiles/s of the per-tile TRAIN step (forward + loss + backward + optimiser) on N MI355X.

Workload (BASELINE.json metric "512x512 tiles/sec (train fwd+bwd)", configs[1] "ResNet50-UNet bf16, 512x512 tiles, batch 32,
1xMI355X"): `Resnet50Unet` (ResNet-50 v1.5 encoder + bilinear-upsample U-Net decoder, NC = 15+1 classes as in the
reference's configs/train/server.yaml:36) through the reference's factory API (load_network / load_loss / load_optimizer,
source/scripts/utils.py:174-242), batch 32 per GPU, 512x512 u8 tiles, bf16 MFMA convolutions with f32 accumulation and f32
master weights, CrossEntropyLoss(ignore_index=0) + SGD2 - the exact step of source/scripts/train.py:121-126.  Synthetic u8
tiles and labels are resident in HBM before the timed region.
`--net Unetv2` measures the reference's own U-Net at the same shape (the round-1 headline; kept as a second record).

One process per GPU.  For N > 1 the driver launches `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(RCCL); a plain `python bench.py --gpus N` spawns exactly that as a CHILD process before anything touches the GPU.
"""
import argparse
import json
import os
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak of MI355X (MI355X_MICROARCH.md, chip-level parameters)
PEAK_FP8_TFLOPS = 5000.0   # dense fp8 peak (block-scaled v_mfma_scale_f32_16x16x128_f8f6f4; same table)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0      # HBM3E spec (6.3 TB/s is what a streaming copy reaches on this chip: MI355X_MICROARCH.md, HBM)
PMC_FILES = {"Resnet50Unet": "r04_pmc_traffic_resnet50unet_b32_s512_bf16.json", "DeepLabV3Plus": "r04_pmc_traffic_deeplabv3plus_b32_s512_bf16.json", "SwinTUperNet": "r02_pmc_traffic_swintupernet_b32_s512_bf16.json", "Unetv2": "r01_pmc_traffic_b32_s512_bf16.json",
             "SegformerMod": "r03_pmc_traffic_segformermod_b32_s512_bf16.json", "MobileNet": "r03_pmc_traffic_mobilenet_b32_s512_bf16.json"}
FAMILY_KERNEL = {
    "conv3x3_halo": "conv3x3_halo_kernel (3x3 / stride 1 forward and data-gradient launches; strided 3x3 data gradients run it on the zero-dilated gradient, counted at their algorithmic FLOPs)",
    "conv3x3_halo_bn_bwd": "conv3x3_halo_kernel + fused BatchNorm-backward reduce pass (conv FLOPs only)",
    "conv_taps": "conv_taps_kernel (non-overlapping taps: 1x1 forward / data gradient, ConvTranspose forward / data gradient)",
    "conv_igemm": "conv_igemm_kernel (generic gather: strided convs, the 7x1 virtual-pixel stem, 1x1 with < 128 output channels)",
    "wgrad": "wgrad_* kernels (all weight gradients, split-K reduce included)",
    "window_attention": "attn_fwd_kernel / attn_bwd_kernel (7x7-window attention, one wavefront per (window, head), VALU; scores recomputed in backward)",
    "wgrad_1x1": "wgrad_gemm_kernel (1x1 weight gradients as one transposed GEMM)",
    "conv3x3_fp8": "conv3x3_fp8_kernel (3x3 / stride 1 forward and data gradient of the layers with channel counts % 128 == 0 on v_mfma_scale_f32_16x16x128_f8f6f4: e4m3 x e4m3 forward, e5m2 x e4m3 data gradient)",
    "hbm_bn_bwd": "bn_bwd_kernel (BatchNorm backward: reduce pass reads y, g; apply pass reads y, g and writes dy - algorithmic bytes = 5 tensor passes)",
    "hbm_bn_act": "bn_act_kernel (BatchNorm apply + ReLU: read y, write the activation)",
    "hbm_residual": "bn_add_act_kernel / relu_bwd_sum_kernel (residual block tail forward and backward)",
    "hbm_resize": "upsample2x_fwd / upsample2x_bwd (bilinear x2 of the decoder)",
    "hbm_patches": "im2col / col2im / phase shuffle (patch matrices of the strided, sequence-reduction and transposed convolutions of SegformerMod)",
    "hbm_dwconv": "depthwise kernels (dwconv3x3 of SegformerMod's Mix-FFN / dwconv k x k of MobileNetV3: forward, data gradient, weight-gradient partials)",
    "hbm_se": "squeeze-excite passes of MobileNetV3 (se_scale forward / backward, image_dot)",
    "sr_attention": "sr_attn_* kernels (spatial-reduction attention forward / backward, VALU; FLOPs = 4 N Nk C forward, 10 N Nk C backward)",
    "hbm_quantize_fp8": "quantize_fp8_kernel (bf16 -> fp8 image + amax of the tensors the fp8 convolutions read)",
    "hbm_gram": "cvcs_gram (second-moment matrix of a bottleneck's conv3 input on the matrix cores: the BatchNorm statistics of the conv3 output without that output; algorithmic bytes = one read of the input)",
    "hbm_head_ce": "head_ce_kernel (1x1 head + softmax cross-entropy + their backward in one pass: reads the last decoder activation and the labels, writes its gradient)",
}


def host_cores():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # honour a cgroup CPU quota (the GPU box gives 16 CPUs of a 256-thread host)
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(p))))
    except Exception:
        pass
    return cores


def cpu_baseline(net, nc, tile, tiles, steps, variant="b0"):
    """the CPU oracle (torch fp32 restatement of the same train step) on a bounded sample of the same workload"""
    import torch
    from oracle import resnet_unet_oracle as R
    from oracle import unet_oracle as O
    torch.set_num_threads(min(host_cores(), 32))
    img, lab = O.synthetic_tiles(tiles, tile, nc, seed=1234)
    if net in ("Unetv2", "Unet"):
        tr, what = O.OracleTrainer(net, nc, opt="SGD2", ignore_index=0, seed=0), f"oracle.unet_oracle (torch-CPU fp32 {net}"
    elif net in ("SwinTUperNet", "SwinBUperNet"):
        from oracle import swin_upernet_oracle as W
        v = "tiny" if net == "SwinTUperNet" else "base"
        tr, what = W.OracleTrainer(nc, opt="SGD2", ignore_index=0, seed=0, variant=v), f"oracle.swin_upernet_oracle (torch-CPU fp32 Swin-{v[0].upper()} + UPerNet"
    elif net == "SegformerMod":
        from oracle import segformer_oracle as SO
        tr, what = SO.OracleTrainer(nc, opt="SGD2", ignore_index=0, seed=0, variant=variant), f"oracle.segformer_oracle (torch-CPU fp32 SegformerMod = MiT-{variant} + the reference's tail, stochastic layers off"
    elif net == "MobileNet":
        from oracle import mobilenet_oracle as MO
        tr, what = MO.OracleTrainer(nc, opt="SGD2", ignore_index=0, seed=0), "oracle.mobilenet_oracle (torch-CPU fp32 DeepLabV3 on MobileNetV3-large"
    elif net in ("DeepLabV3Plus", "Resnet101"):
        from oracle import deeplab_oracle as D
        plus = net == "DeepLabV3Plus"
        tr = D.OracleTrainer("resnet50" if plus else "resnet101", nc, opt="SGD2", ignore_index=0, seed=0, output_stride=16 if plus else 8, plus=plus)
        what = f"oracle.deeplab_oracle (torch-CPU fp32 {net}"
    else:
        arch = {"Resnet50Unet": "resnet50", "Resnet18Unet": "resnet18", "Resnet34Unet": "resnet34"}[net]
        tr, what = R.OracleTrainer(arch, nc, opt="SGD2", ignore_index=0, seed=0), f"oracle.resnet_unet_oracle (torch-CPU fp32 {net}"
    tr.step(img, lab)  # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(img, lab)
    dt = time.perf_counter() - t0
    return {"value": round(tiles * steps / dt, 4), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{what} NC={nc}, CE ignore_index=0, SGD2): {steps} train steps of {tiles} tiles {tile}x{tile} after 1 warm-up step"}


def parity_at_dtype(net, name, nc, tile, dev, variant="b0"):
    """eval-mode logits and labels of the benchmarked network AT THE BENCHMARKED PRECISION against the f32 CPU oracle evaluated at the same
    (current) parameters, on one structured synthetic tile of the benchmarked size - the figures tests/test_named_configs_gpu.py asserts.
    The north star's "1e-3 on logits, bit-exact argmax" is met by the f32 path only (that test); bf16 / fp8 storage cannot reach it."""
    import torch
    from oracle import unet_oracle as O
    back = getattr(net, "_from_reference", lambda k: k)      # (the torchvision-keyed wrappers export `model.backbone.*`; the oracle speaks the engine's names)
    p = {back(k): v.detach().cpu().float() if v.is_floating_point() else v.detach().cpu() for k, v in net.state_dict().items()}
    if name in ("Resnet50Unet", "Resnet18Unet", "Resnet34Unet"):
        from oracle import resnet_unet_oracle as R
        arch = {"Resnet50Unet": "resnet50", "Resnet18Unet": "resnet18", "Resnet34Unet": "resnet34"}[name]
        fwd = lambda x: R.forward(p, x, arch, train=False)          # noqa: E731
    elif name in ("DeepLabV3Plus", "Resnet101"):
        from oracle import deeplab_oracle as D
        plus = name == "DeepLabV3Plus"
        fwd = lambda x: D.forward(p, x, "resnet50" if plus else "resnet101", train=False, output_stride=16 if plus else 8, plus=plus)   # noqa: E731
    elif name in ("SwinTUperNet", "SwinBUperNet"):
        from oracle import swin_upernet_oracle as W
        fwd = lambda x: W.forward(p, x, train=False)   # noqa: E731
    elif name == "SegformerMod":
        from oracle import segformer_oracle as SO
        fwd = lambda x: SO.forward(p, x, train=False, variant=variant)   # noqa: E731
    elif name == "MobileNet":
        from oracle import mobilenet_oracle as MO
        fwd = lambda x: MO.forward(p, x, train=False)   # noqa: E731
    else:
        fwd = lambda x: O.unet_forward(p, x, name, train=False)   # noqa: E731
    img, _ = O.synthetic_tiles(1, tile, nc, seed=11, structured=True)
    net.eval()
    with torch.no_grad():
        want = fwd(img.float())
        got = net(img.to(dev), None).float().cpu()
    scale = want.abs().max().item()
    d = got - want
    lab, ref = got.argmax(1), want.argmax(1)
    top2 = want.topk(2, dim=1).values
    bad = lab != ref
    return {"against": "f32 CPU oracle at the same (initial) parameters, eval mode, 1 structured synthetic tile of the benchmarked size",
            "logit_err_max": round(d.abs().max().item() / scale, 6), "logit_err_rms": round(d.pow(2).mean().sqrt().item() / scale, 6),
            "unit": "fraction of max|logit|", "labels_differing": int(bad.sum()), "labels_total": int(lab.numel()),
            "largest_oracle_top2_margin_among_them": round(((top2[:, 0] - top2[:, 1])[bad].max().item() / scale) if bool(bad.any()) else 0.0, 6),
            "north_star_tolerance": "1e-3 on logits / bit-exact argmax: met by --precision fp32 (tests/test_named_configs_gpu.py), not reachable in bf16 / fp8 storage"}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a child (this process has not touched the
    GPU, and it does not exec - it waits and exits with the child's code)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="tiles per GPU per step")
    ap.add_argument("--tile", type=int, default=512)
    ap.add_argument("--classes", type=int, default=15, help="config num_classes (NC = classes + 1)")
    ap.add_argument("--net", default="Resnet50Unet", choices=["Resnet50Unet", "Resnet18Unet", "Resnet34Unet", "DeepLabV3Plus", "Resnet101", "SwinTUperNet", "SwinBUperNet", "SegformerMod", "MobileNet", "Unetv2", "Unet"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp8"],
                    help="fp8 = mixed bf16 / fp8 convolutions (BASELINE configs[4]; ResNet-UNets and the UPerNet head of Swin + UPerNet)")
    ap.add_argument("--segformer-variant", default="b0", choices=["b0", "b3"], help="SegformerMod: b0 = SegformerConfig() (the reference's pretrained=False branch), "
                    "b3 = the architecture of the checkpoint its default pretrained=True fetches")
    ap.add_argument("--decoder-channels", default=None, help="ResNet-UNets: the five decoder widths, e.g. 256,128,64,64,64 (the widths of rounds 1-2; "
                    "default 256,128,64,32,16 since round 3 - config key `decoder_channels`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    import torch
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    # rehearsal hooks for a one-GPU box (tests/test_bench_multirank_gpu.py): every rank on cuda:0, collectives over gloo
    if os.environ.get("CVCS_BENCH_ONE_DEVICE") == "1":
        local = 0
    backend = os.environ.get("CVCS_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend)

    from cvcs_amd import ops, utils
    cfg = {"net": a.net, "num_classes": a.classes, "precision": a.precision, "loss": "CEL", "ignore_background": True,
           "opt": "SGD2", "epochs": 20, "device": "gpu", "segformer_variant": a.segformer_variant}
    NC = a.classes + 1
    dec = None
    if a.decoder_channels:
        assert a.net.startswith("Resnet") and a.net.endswith("Unet"), "--decoder-channels is a key of the ResNet-UNets"
        dec = cfg["decoder_channels"] = [int(c) for c in a.decoder_channels.split(",")]
    torch.manual_seed(0)
    net = utils.load_network(cfg, dev)
    parity = None
    if world == 1 and not a.no_cpu_baseline:
        # at INITIALISATION, before the timed schedule trains the network on uniform-random labels (outside every timed region)
        parity = parity_at_dtype(net, a.net, NC, a.tile, dev, a.segformer_variant)
    crit = utils.load_loss(cfg, dev)
    opt, sched = utils.load_optimizer(cfg, net)
    if world > 1:
        from cvcs_amd.parallel import DataParallel
        net.flat_parameters()   # materialise the flat buffers / engine
        DataParallel(net, opt)
    g = torch.Generator().manual_seed(1234 + rank)
    img = torch.randint(0, 256, (a.batch, 3, a.tile, a.tile), dtype=torch.uint8, generator=g).to(dev)
    lab = torch.randint(0, NC, (a.batch, a.tile, a.tile), dtype=torch.uint8, generator=g).to(dev)
    net.train()

    def step():
        pred = net(img, None)
        loss = crit(pred, lab)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    probe = {}
    if not a.no_kernel_timers:
        # Every tagged launch bracketed by two HIP events costs the step ~2.5 % (600 launches): two untimed, fully instrumented steps give
        # the kernel-family table and name the dominant conv family; the TIMED region then brackets that family only (+ the 3x3 halo
        # convs for `roofline_encoder`) - `roofline` is measured live over exactly the timed steps, `kernel_families` over the probe steps
        ops.TIMERS = ops.KernelTimers()
        for _ in range(2):
            step()
        probe = ops.TIMERS.summary()
        pf = {}
        for kind, t in probe.items():          # EVERY tagged family competes: convolutions, weight gradients, the HBM-bound passes
            pf[kind.split(":")[0]] = pf.get(kind.split(":")[0], 0.0) + t["total_ms"]
        dom_family = max(pf, key=pf.get)
        ops.TIMERS = ops.KernelTimers(only={dom_family, "conv3x3_halo", "conv3x3_fp8"})
        fence()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        loss = step()
        marks[i + 1].record()
    fence()
    dt = time.perf_counter() - t0
    timers = ops.TIMERS.summary() if ops.TIMERS is not None else {}
    ops.TIMERS = None
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
    last_loss = loss.item()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    if rank == 0:
        peak = PEAK_F32_TFLOPS if a.precision == "fp32" else PEAK_BF16_TFLOPS
        widths = "/".join(str(c) for c in (dec or getattr(net, "decoder_channels", ())))
        model = {"Resnet50Unet": f"ResNet50-UNet (ResNet-50 v1.5 encoder, bilinear-upsample decoder {widths}, 1x1 head)",
                 "Resnet18Unet": f"ResNet18-UNet (decoder {widths})", "Resnet34Unet": f"ResNet34-UNet (decoder {widths})",
                 "DeepLabV3Plus": "DeepLabV3+ (ResNet-50 v1.5 at output stride 16, ASPP rates 6/12/18, 64-channel low-level branch)",
                 "Resnet101": "DeepLabV3-ResNet101 (the reference's factory name Resnet101, S/nets.py:234-257; output stride 8, ASPP 12/24/36)",
                 "SwinTUperNet": "Swin-T + UPerNet (BASELINE configs[3]: embed 96, depths 2/2/6/2, window 7; UPerNet hidden 512, pool scales 1/2/3/6)",
                 "SwinBUperNet": "Swin-B + UPerNet (the reference's documented `BSwin` name: embed 128, depths 2/2/18/2, window 7; UPerNet hidden 512)",
                 "SegformerMod": "SegformerMod (the reference's factory name, S/nets.py:313-349: transformers' Segformer " + ("MiT-b0 = SegformerConfig()" if a.segformer_variant == "b0" else "MiT-b3 (the checkpoint architecture its default fetches)") + ", NC-class head, ConvTranspose 8/2/3 + 4/2/1 + Conv3x3 tail; DropPath 0-0.1 and Dropout 0.1 active)",
                 "MobileNet": "MobileNet (the reference's factory name, S/nets.py:277-311: torchvision deeplabv3_mobilenet_v3_large = dilated MobileNetV3-large at output stride 16 + DeepLabHead, ASPP 12/24/36, Dropout 0.5 active)",
                 "Unetv2": "Unetv2 (the reference's own U-Net, source/scripts/nets.py:117-199)", "Unet": "Unet (reference)"}[a.net]
        out = {
            "metric": "512x512 tiles/sec (train fwd+bwd)" if a.tile == 512 else f"{a.tile}x{a.tile} tiles/sec (train fwd+bwd)",
            "value": round(world * a.batch * a.steps / dt, 3), "unit": "tiles/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 + fp8 (e4m3 activations / weights, e5m2 gradients; f32 accumulate)" if a.precision == "fp8" else a.precision,
            "data": "synthetic",
            "config": {"workload": f"{model}, NC={NC}, train step: fwd + CE(ignore_index=0) + bwd + SGD2, "
                                   f"{a.tile}x{a.tile} u8 tiles, batch {a.batch}/GPU, "
                                   + ("mixed bf16 / fp8 MFMA convs (fp8: 3x3 stride-1 layers with channel counts % 128 == 0, forward + data gradient)"
                                      if a.precision == "fp8" else f"{a.precision} MFMA convs") + ", f32 accumulate/master",
                       "global_batch": world * a.batch, "tile": a.tile, "num_classes": NC, "decoder_channels": list(dec or getattr(net, "decoder_channels", ())) or None,
                       "parallelism": f"dp{world}" if world > 1 else "single"},
            "loss": round(last_loss, 5),
            "median_ms_per_step": round(statistics.median(per_step), 3),
            "timed_region_s": round(dt, 3),
        }
        if timers:
            # kernel families (HIP-event brackets on the launch stream); scopes: enc / dec.  `live`: inside the timed region
            def families(tt):
                out_ = {}
                for kind, t in tt.items():
                    fam = kind.split(":")[0]
                    f = out_.setdefault(fam, dict(ms=0.0, flops=0.0, launches=0))
                    f["ms"] += t["total_ms"]; f["flops"] += t["flops"]; f["launches"] += t["launches"]
                return out_
            live, fams = families(timers), families(probe)
            psteps = 2

            def line(f, kernel, nsteps, fam=""):
                sec = f["ms"] * 1e-3
                if fam.startswith("hbm_"):     # `flops` of these families is their algorithmic BYTE count (ops._tag_hbm)
                    gbs = f["flops"] / sec / 1e9 if sec > 0 else 0.0
                    return {"kernel": kernel, "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": round(gbs / PEAK_HBM_GBS, 4), "launches_per_step": f["launches"] // nsteps,
                            "avg_launch_us": round(1e3 * f["ms"] / f["launches"], 2),
                            "algorithmic_mbytes_per_step": round(f["flops"] / nsteps / 1e6, 1), "ms_per_step": round(f["ms"] / nsteps, 3)}
                pk = PEAK_FP8_TFLOPS if fam == "conv3x3_fp8" else peak
                tf = f["flops"] / sec / 1e12 if sec > 0 else 0.0
                return {"kernel": kernel, "bound": "mfma", "achieved": round(tf, 2), "peak": pk, "unit": "TFLOP/s",
                        "frac": round(tf / pk, 4), "launches_per_step": f["launches"] // nsteps,
                        "avg_launch_us": round(1e3 * f["ms"] / f["launches"], 2),
                        "algorithmic_gflop_per_step": round(f["flops"] / nsteps / 1e9, 1),
                        "ms_per_step": round(f["ms"] / nsteps, 3)}
            dom = max(fams, key=lambda k: fams[k]["ms"])     # the family with the largest time over ALL tagged families
            out["roofline"] = line(live[dom], FAMILY_KERNEL.get(dom, dom), a.steps, dom)
            out["roofline"]["measured"] = f"HIP events on the launch stream around every launch of this family in the {a.steps} timed steps"
            out["roofline"]["traffic"] = None
            out["kernel_families"] = {k: line(f, FAMILY_KERNEL.get(k, k), psteps, k) for k, f in fams.items() if k != dom}
            out["kernel_families_measured"] = f"{psteps} fully instrumented steps before the timed region (bracketing all ~600 launches costs the step 2.5 %)"
            # the 3x3 ENCODER convolutions (the set the north star's >= 50 % target is written for), forward + data gradient
            enc = [t for kind, t in timers.items() if kind.startswith("conv3x3_halo") and kind.endswith(":enc")]
            if enc:
                e = dict(ms=sum(t["total_ms"] for t in enc), flops=sum(t["flops"] for t in enc), launches=sum(t["launches"] for t in enc))
                out["roofline_encoder"] = line(e, "conv3x3_halo_kernel, encoder 3x3 convolutions only (forward + data gradient)", a.steps)
            enc8 = [t for kind, t in timers.items() if kind.startswith("conv3x3_fp8") and kind.endswith(":enc")]
            if enc8:
                e = dict(ms=sum(t["total_ms"] for t in enc8), flops=sum(t["flops"] for t in enc8), launches=sum(t["launches"] for t in enc8))
                out["roofline_encoder_fp8"] = line(e, "conv3x3_fp8_kernel, encoder 3x3 convolutions only (forward + data gradient)", a.steps, "conv3x3_fp8")
            pmc = os.path.join(ROOT, "profiles", PMC_FILES.get(a.net, ""))
            if a.batch == 32 and a.tile == 512 and a.precision == "bf16" and os.path.isfile(pmc):
                kern = {"conv3x3_halo": "conv3x3_halo_kernel", "conv_taps": "conv_taps_kernel", "conv_igemm": "conv_igemm_kernel",
                        "hbm_bn_bwd": "bn_bwd_kernel", "wgrad": "wgrad", "hbm_residual": "relu_bwd_sum_bn_kernel", "hbm_dwconv": "dwconv"}.get(dom)
                pm = json.load(open(pmc)).get(kern)
                if pm:
                    out["roofline"]["traffic"] = round(pm["hbm_bytes_per_launch"])
                    out["roofline"]["traffic_unit"] = f"bytes per launch (PMC, profiles/{PMC_FILES[a.net]})"
            mf = {k: f for k, f in fams.items() if not k.startswith("hbm_")}
            tot = sum(f["ms"] for f in mf.values()) / psteps
            out["mfma_kernels_share_of_step"] = round(tot / (1e3 * dt / a.steps), 3)
            out["algorithmic_gflop_per_step"] = round(sum(f["flops"] for f in mf.values()) / psteps / 1e9, 1)
            # (carried in `config` too: tiles/s of two rounds compare only at equal work per tile - rounds 2 -> 3 changed the decoder widths)
            out["config"]["algorithmic_gflop_per_tile"] = round(sum(f["flops"] for f in mf.values()) / psteps / 1e9 / a.batch, 2)
            out["step_tflops"] = round(sum(f["flops"] for f in mf.values()) / psteps * a.steps / dt / 1e12, 1)
        if world == 1 and not a.no_cpu_baseline:
            out["parity_at_dtype"] = parity
            out["cpu_baseline"] = cpu_baseline(a.net, NC, a.tile, 2, 2, a.segformer_variant)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py - tiles/s of the per-tile TRAIN step (forward + loss + backward + optimiser) on N MI355X.

Workload (BASELINE.json metric "512x512 tiles/sec (train fwd+bwd)"): the reference's own model `Unetv2`
(source/scripts/nets.py:117-199; NC = 15+1 as in configs/train/server.yaml:36) at the shape of BASELINE
configs[1]: batch 32 per GPU, 512x512 tiles, bf16 MFMA convolutions with f32 accumulation / f32 master weights,
CrossEntropyLoss(ignore_index=0) + SGD2 - the exact step of source/scripts/train.py:121-126, through the same
factory API (load_network / load_loss / load_optimizer).  Synthetic u8 tiles and labels are resident in HBM before
the timed region.  BASELINE's "ResNet50-UNet" has no definition in the reference; see DESIGN.md.

One process per GPU; for N > 1 launch through `python -m torch.distributed.run --nproc-per-node N ...` (RCCL).
"""
import argparse
import json
import os
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak of MI355X (MI355X_MICROARCH.md, chip-level parameters)
PMC_FILE = os.path.join(ROOT, "profiles", "r01_pmc_traffic_b32_s512_bf16.json")  # scripts/pmc_traffic.py output


def cpu_baseline(nc, tile, tiles, steps):
    """the CPU oracle (torch fp32 restatement of the reference step) on a bounded sample of the same workload."""
    from oracle import unet_oracle as O
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # honour a cgroup CPU quota (the GPU box gives 16 CPUs of a 256-thread host)
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(p))))
    except Exception:
        pass
    torch.set_num_threads(min(cores, 32))
    img, lab = O.synthetic_tiles(tiles, tile, nc, seed=1234)
    tr = O.OracleTrainer("Unetv2", nc, opt="SGD2", ignore_index=0, seed=0)
    tr.step(img, lab)  # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(img, lab)
    dt = time.perf_counter() - t0
    return {"value": round(tiles * steps / dt, 4), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle OracleTrainer (torch-CPU fp32 Unetv2 NC={nc}, CE ignore_index=0, SGD2): {steps} train steps "
                      f"of {tiles} tiles {tile}x{tile} after 1 warm-up step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="tiles per GPU per step")
    ap.add_argument("--tile", type=int, default=512)
    ap.add_argument("--classes", type=int, default=15, help="config num_classes (NC = classes + 1)")
    ap.add_argument("--net", default="Unetv2")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
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
           "opt": "SGD2", "epochs": 20, "device": "gpu"}
    NC = a.classes + 1
    torch.manual_seed(0)
    net = utils.load_network(cfg, dev)
    crit = utils.load_loss(cfg, dev)
    opt, sched = utils.load_optimizer(cfg, net)
    if world > 1:
        from cvcs_amd.parallel import DataParallel
        net(torch.zeros(1, 3, 32, 32, dtype=torch.uint8, device=dev))  # materialise the flat buffers / engine
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
    if not a.no_kernel_timers:
        ops.TIMERS = ops.KernelTimers()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    timers = ops.TIMERS.summary() if ops.TIMERS is not None else {}
    ops.TIMERS = None
    last_loss = loss.item()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    if rank == 0:
        out = {
            "metric": "512x512 tiles/sec (train fwd+bwd)" if a.tile == 512 else f"{a.tile}x{a.tile} tiles/sec (train fwd+bwd)",
            "value": round(world * a.batch * a.steps / dt, 3), "unit": "tiles/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"{a.net} (reference U-Net, NC={NC}) train step: fwd + CE(ignore_index=0) + bwd + SGD2, "
                                   f"{a.tile}x{a.tile} u8 tiles, batch {a.batch}/GPU, {a.precision} MFMA convs, f32 accumulate/master",
                       "global_batch": world * a.batch, "tile": a.tile, "num_classes": NC,
                       "parallelism": f"dp{world}" if world > 1 else "single"},
            "loss": round(last_loss, 5),
        }
        if "conv3x3_halo" in timers:
            c = timers["conv3x3_halo"]
            out["roofline"] = {"bound": "mfma", "achieved": round(c["tflops"], 2), "peak": PEAK_BF16_TFLOPS if a.precision == "bf16" else 157.3,
                               "unit": "TFLOP/s", "frac": round(c["tflops"] / (PEAK_BF16_TFLOPS if a.precision == "bf16" else 157.3), 4),
                               "traffic": None, "kernel": "conv3x3_halo_kernel (every 3x3 forward and data-gradient launch that is a convolution only; "
                                                          "the launches that also carry a BatchNorm-backward reduce pass are reported as fused_conv)",
                               "launches_per_step": c["launches"] // a.steps, "avg_launch_us": round(c["avg_us"], 2),
                               "algorithmic_gflop_per_step": round(c["flops"] / a.steps / 1e9, 1)}
            if "conv3x3_halo_bn_bwd" in timers:   # same kernel, plus the fused first pass of the BatchNorm backward below it
                f = timers["conv3x3_halo_bn_bwd"]
                out["fused_conv"] = {"kernel": "conv3x3_halo_kernel + fused BatchNorm-backward reduce (64- and 128-channel blocks; conv FLOPs only)",
                                     "achieved": round(f["tflops"], 2), "unit": "TFLOP/s", "launches_per_step": f["launches"] // a.steps,
                                     "avg_launch_us": round(f["avg_us"], 2), "algorithmic_gflop_per_step": round(f["flops"] / a.steps / 1e9, 1)}
            if "conv_igemm" in timers:   # ConvTranspose forward / data gradient on the non-overlapping-taps kernel
                g = timers["conv_igemm"]
                out["other_conv"] = {"kernel": "conv_taps_kernel (ConvTranspose forward + data gradient: short-K GEMMs, the shallow levels HBM-bound)",
                                     "achieved": round(g["tflops"], 2), "unit": "TFLOP/s", "launches_per_step": g["launches"] // a.steps,
                                     "avg_launch_us": round(g["avg_us"], 2), "algorithmic_gflop_per_step": round(g["flops"] / a.steps / 1e9, 1)}
            if "wgrad" in timers:
                w = timers["wgrad"]
                out["roofline_wgrad"] = {"bound": "mfma", "achieved": round(w["tflops"], 2), "unit": "TFLOP/s",
                                         "frac": round(w["tflops"] / (PEAK_BF16_TFLOPS if a.precision == "bf16" else 157.3), 4),
                                         "launches_per_step": w["launches"] // a.steps, "avg_launch_us": round(w["avg_us"], 2),
                                         "algorithmic_gflop_per_step": round(w["flops"] / a.steps / 1e9, 1)}
            # HBM traffic of the same kernel family from the rocprofv3 PMC passes of this exact workload (FETCH_SIZE
            # doubled per the gfx950 correction, WRITE_SIZE exact), collected with scripts/pmc_traffic.py
            if a.batch == 32 and a.tile == 512 and a.precision == "bf16" and a.net == "Unetv2" and os.path.exists(PMC_FILE):
                pm = json.load(open(PMC_FILE)).get("conv3x3_halo_kernel")
                if pm:
                    out["roofline"]["traffic"] = round(pm["hbm_bytes_per_launch"])
                    out["roofline"]["traffic_unit"] = "bytes per launch (PMC, profiles/r01_pmc_traffic_b32_s512_bf16.json)"
            tot = sum(t["total_ms"] for t in timers.values())
            out["mfma_kernels_share_of_step"] = round(tot / (1e3 * dt), 3)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(NC, a.tile, 2, 2)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

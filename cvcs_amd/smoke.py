"""`__graft_entry__.smoke()`: one tiny train step of the hot path on cuda:0, checked against the CPU oracle.
(The only module of the package allowed to import `oracle/`: the oracle is the checker, never the product path.)"""
import torch


def run():
    from oracle import unet_oracle as O
    from . import nets, utils
    NC, B, S = 5, 2, 64     # 64x64 tiles: the deepest BatchNorm still averages over 2 x 4 x 4 values per channel
    dev = "cuda:0"
    p0 = O.init_params("Unetv2", NC, seed=3)
    img, lab = O.synthetic_tiles(B, S, NC, seed=11)
    tr = O.OracleTrainer("Unetv2", NC, opt="SGD2", ignore_index=0, seed=3)
    ref_loss, ref_logits, _ = tr.step(img, lab)
    # bf16 with random-init weights on noise tiles: torch-CPU run in bfloat16 itself is 0.21 (max) / 0.027 (mean) away from
    # its fp32 run on this fixture; the HIP bf16 path (f32 accumulation and statistics) measures 0.19 / 0.02
    for precision, tol, mean_tol in (("fp32", 2e-3, 2e-4), ("bf16", 3e-1, 5e-2)):
        net = nets.Urnetv2(NC, precision)
        net.load_state_dict(p0, strict=False)
        net = net.to(dev)
        crit = utils.CrossEntropyLoss(ignore_index=0)
        opt, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 1}, net)
        net.train()
        pred = net(img.to(dev).float(), None)
        loss = crit(pred, lab.to(dev))
        opt.zero_grad()
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        diff = (pred.detach().cpu() - ref_logits).abs()
        scale = max(1.0, ref_logits.abs().max().item())
        err, merr = diff.max().item() / scale, diff.mean().item() / scale
        assert err < tol and merr < mean_tol, f"{precision}: logits differ from the oracle by {err:.3e} max / {merr:.3e} mean (rel)"
        assert abs(loss.item() - ref_loss) < mean_tol * max(1.0, abs(ref_loss)), (precision, loss.item(), ref_loss)
        w = dict(net.named_parameters())["decode_forward4.1.weight"].detach().cpu()
        werr = (w - tr.p["decode_forward4.1.weight"].detach()).abs().max().item()
        assert werr < mean_tol, f"{precision}: post-step head weights differ by {werr:.3e}"
        print(f"smoke {precision}: loss {loss.item():.6f} (oracle {ref_loss:.6f}), rel logit err max {err:.2e} mean {merr:.2e}, "
              f"post-step head weight err {werr:.2e}")

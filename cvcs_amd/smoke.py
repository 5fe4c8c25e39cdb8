"""`__graft_entry__.smoke()`: one tiny train step of the hot path on cuda:0, checked against the CPU oracle.
(The only module of the package allowed to import `oracle/`: the oracle is the checker, never the product path.)"""
import torch


def run():
    from oracle import unet_oracle as O
    from . import nets, utils
    NC, B, S = 5, 2, 32
    dev = "cuda:0"
    p0 = O.init_params("Unetv2", NC, seed=3)
    img, lab = O.synthetic_tiles(B, S, NC, seed=11)
    tr = O.OracleTrainer("Unetv2", NC, opt="SGD2", ignore_index=0, seed=3)
    ref_loss, ref_logits, _ = tr.step(img, lab)
    for precision, tol in (("fp32", 2e-3), ("bf16", 1.5e-1)):
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
        err = (pred.detach().cpu() - ref_logits).abs().max().item() / max(1.0, ref_logits.abs().max().item())
        assert err < tol, f"{precision}: logits differ from the oracle by {err:.3e} (rel)"
        assert abs(loss.item() - ref_loss) < tol * max(1.0, abs(ref_loss)), (precision, loss.item(), ref_loss)
        w = dict(net.named_parameters())["decode_forward4.1.weight"].detach().cpu()
        werr = (w - tr.p["decode_forward4.1.weight"].detach()).abs().max().item()
        assert werr < tol, f"{precision}: post-step head weights differ by {werr:.3e}"
        print(f"smoke {precision}: loss {loss.item():.6f} (oracle {ref_loss:.6f}), max rel logit err {err:.2e}")

"""mIoU parity run (north star: "mIoU on a held-out synthetic set matching the CPU reference to +-0.1"):
train the SAME schedule on structured synthetic tiles with (a) the CPU oracle (torch fp32) and (b) the HIP path,
then evaluate both on a held-out set with the reference's mIoU definition (S/utils.py:311-364).
usage: python scripts/miou_parity.py [bf16|fp32] [steps] [tile] [Unetv2|Resnet18Unet|Resnet50Unet|DeepLabV3Plus|TSwin]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cvcs_amd import nets, utils  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402

NC, B, DEV = 5, 8, "cuda:0"


def _setup(model, precision, emulate=False, seed=3):
    """(oracle trainer, oracle eval-forward, HIP network) of one model family, same initial parameters.  emulate: the oracle TRAINS in its
    bf16-emulation mode (the yardstick for how far two correct bf16 runs of this schedule land from the f32 one)"""
    if model in ("Unetv2", "Unet"):
        tr = O.OracleTrainer(model, NC, opt="SGD2", ignore_index=0, seed=seed)
        fwd = lambda p, x: O.unet_forward(p, x, model, train=False)   # noqa: E731
        net, p0 = (nets.Urnetv2 if model == "Unetv2" else nets.Urnet)(NC, precision), O.init_params(model, NC, seed=seed)
    elif model in ("Resnet18Unet", "Resnet50Unet"):
        from oracle import resnet_unet_oracle as R
        arch = "resnet18" if model == "Resnet18Unet" else "resnet50"
        tr = R.OracleTrainer(arch, NC, opt="SGD2", ignore_index=0, seed=seed)
        fwd = lambda p, x: R.forward(p, x, arch, train=False)         # noqa: E731
        net, p0 = getattr(nets, model)(NC, precision), R.init_params(arch, NC, seed=seed)
    elif model == "DeepLabV3Plus":
        from oracle import deeplab_oracle as D
        tr = D.OracleTrainer("resnet50", NC, "SGD2", ignore_index=0, seed=seed, output_stride=16, plus=True)
        fwd = lambda p, x: D.forward(p, x, "resnet50", train=False, output_stride=16, plus=True)   # noqa: E731
        net, p0 = nets.DeepLabV3Plus(NC, precision), D.init_params("resnet50", NC, seed=seed, plus=True)
    elif model == "TSwin":
        from oracle import swin_upernet_oracle as W
        tr = W.OracleTrainer(NC, opt="SGD2", ignore_index=0, seed=seed, emulate_bf16=emulate)
        fwd = lambda p, x: W.forward(p, x, train=False, emulate_bf16=emulate)   # noqa: E731
        net, p0 = nets.SwinTUperNet(NC, precision), W.init_params(NC, seed=seed)
    else:
        raise ValueError(model)
    net.load_state_dict(p0, strict=False)
    return tr, fwd, (net.to(DEV) if torch.cuda.is_available() else net)


def oracle_only(steps, S, model, emulate, seed=0):
    """held-out metrics of the CPU oracle alone (f32, or trained and evaluated in its bf16-emulation mode); seed: the ensemble member of run()"""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    train = [O.synthetic_tiles(B, S, NC, seed=100 + i + 1000 * seed, structured=True) for i in range(8)]
    held = [O.synthetic_tiles(B, S, NC, seed=900 + i, structured=True) for i in range(4)]
    tr, fwd, _ = _setup(model, "bf16", emulate, seed=3 + seed)
    for s in range(steps):
        tr.step(*train[s % len(train)])
    p = {k: v.detach() for k, v in tr.p.items()}
    with torch.no_grad():
        conf = sum(O.confusion_matrix(O.predict_labels(fwd(p, img.float())).numpy(), lab.numpy(), 16, ignore_index=0) for img, lab in held)
    return O.metrics(conf)


def run(precision="bf16", steps=40, S=64, verbose=True, model="Unetv2", seed=0, epoch_steps=None, cross=False):
    """seed: another member of the ensemble - other initial parameters and other training tiles (the held-out set stays the same).
    epoch_steps: every so many steps count as one epoch of the reference's schedule (S/utils.py:213-221: SGD2 = PolynomialLR over 20 epochs,
    stepped once per epoch, S/train.py): with steps = 20 * epoch_steps the learning rate decays linearly to ~0 by the end of the run, as it
    does at the end of the reference's training; None: no scheduler step, constant learning rate"""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    train = [O.synthetic_tiles(B, S, NC, seed=100 + i + 1000 * seed, structured=True) for i in range(8)]
    held = [O.synthetic_tiles(B, S, NC, seed=900 + i, structured=True) for i in range(4)]
    tr, fwd, net = _setup(model, precision, seed=3 + seed)
    crit = utils.CrossEntropyLoss(ignore_index=0)
    epochs = 1 if epoch_steps is None else -(-steps // epoch_steps)
    opt, sched = utils.load_optimizer({"opt": "SGD2", "epochs": epochs}, net)
    net.train()
    for s in range(steps):
        img, lab = train[s % len(train)]
        lo = tr.step(img, lab)[0]
        loss = crit(net(img.to(DEV), None), lab.to(DEV))
        opt.zero_grad(); loss.backward(); opt.step()
        if epoch_steps is not None and (s + 1) % epoch_steps == 0:
            tr.epoch += 1
            sched.step()
        if verbose and (s % 10 == 0 or s == steps - 1):
            print(f"step {s:3d}  oracle loss {lo:.4f}  hip loss {loss.item():.4f}", flush=True)
    p = {k: v.detach() for k, v in tr.p.items()}
    with torch.no_grad():
        conf_o = sum(O.confusion_matrix(O.predict_labels(fwd(p, img.float())).numpy(), lab.numpy(), 16, ignore_index=0) for img, lab in held)

    def hip_eval(n):
        n.eval()
        cm = utils.ConfusionMatrix(16, ignore_index=0, device=DEV)
        with torch.no_grad():
            for img, lab in held:
                cm.update_from_logits(n(img.to(DEV), None), lab.to(DEV))
        return utils.print_metrics(cm.compute(), silent=True)
    m_h = hip_eval(net)
    if not cross:
        return O.metrics(conf_o), m_h
    # the ORACLE'S trained model on the HIP evaluation path: no trajectory in between - the held-out mIoU of the same weights and running
    # statistics through the reference's model.eval() + metric code (S/utils.py:311-364) must agree
    _, _, net2 = _setup(model, precision, seed=3 + seed)
    net2.load_state_dict({k: v.clone() for k, v in p.items()}, strict=False)
    return O.metrics(conf_o), m_h, hip_eval(net2)


if __name__ == "__main__":
    precision = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    model = sys.argv[4] if len(sys.argv) > 4 else "Unetv2"
    t0 = time.time()
    m_o, m_h = run(precision, steps, S, model=model)
    if len(sys.argv) > 5 and sys.argv[5] == "emu":
        m_e = oracle_only(steps, S, model, True)
        print(f"{model} oracle trained in bf16 emulation: mIoU {100 * m_e['mIoU']:.3f} ({100 * abs(m_e['mIoU'] - m_o['mIoU']):.3f} points from its f32 run)")
    print(f"{model} held-out mIoU: oracle {100 * m_o['mIoU']:.3f}  hip[{precision}] {100 * m_h['mIoU']:.3f}  "
          f"diff {100 * abs(m_o['mIoU'] - m_h['mIoU']):.3f} points; OA oracle {m_o['oa_score']:.4f} hip {m_h['oa_score']:.4f}; {time.time() - t0:.0f}s")

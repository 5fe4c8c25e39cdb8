"""mIoU parity run (north star: "mIoU on a held-out synthetic set matching the CPU reference to +-0.1"):
train the SAME schedule on structured synthetic tiles with (a) the CPU oracle (torch fp32) and (b) the HIP path,
then evaluate both on a held-out set with the reference's mIoU definition (S/utils.py:311-364).
usage: python scripts/miou_parity.py [bf16|fp32] [steps] [tile]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cvcs_amd import nets, utils  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402

NC, B, DEV = 5, 8, "cuda:0"


def run(precision="bf16", steps=40, S=64, verbose=True):
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    train = [O.synthetic_tiles(B, S, NC, seed=100 + i, structured=True) for i in range(8)]
    held = [O.synthetic_tiles(B, S, NC, seed=900 + i, structured=True) for i in range(4)]
    tr = O.OracleTrainer("Unetv2", NC, opt="SGD2", ignore_index=0, seed=3)
    net = nets.Urnetv2(NC, precision)
    net.load_state_dict(O.init_params("Unetv2", NC, seed=3), strict=False)
    net = net.to(DEV)
    crit = utils.CrossEntropyLoss(ignore_index=0)
    opt, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 1}, net)
    net.train()
    for s in range(steps):
        img, lab = train[s % len(train)]
        lo = tr.step(img, lab)[0]
        loss = crit(net(img.to(DEV), None), lab.to(DEV))
        opt.zero_grad(); loss.backward(); opt.step()
        if verbose and (s % 10 == 0 or s == steps - 1):
            print(f"step {s:3d}  oracle loss {lo:.4f}  hip loss {loss.item():.4f}", flush=True)
    p = {k: v.detach() for k, v in tr.p.items()}
    conf_o = sum(O.confusion_matrix(O.predict_labels(O.unet_forward(p, img.float(), "Unetv2", train=False)).numpy(),
                                    lab.numpy(), 16, ignore_index=0) for img, lab in held)
    net.eval()
    cm = utils.ConfusionMatrix(16, ignore_index=0, device=DEV)
    with torch.no_grad():
        for img, lab in held:
            cm.update_from_logits(net(img.to(DEV), None), lab.to(DEV))
    return O.metrics(conf_o), utils.print_metrics(cm.compute(), silent=True)


if __name__ == "__main__":
    precision = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    t0 = time.time()
    m_o, m_h = run(precision, steps, S)
    print(f"held-out mIoU: oracle {100 * m_o['mIoU']:.3f}  hip[{precision}] {100 * m_h['mIoU']:.3f}  "
          f"diff {100 * abs(m_o['mIoU'] - m_h['mIoU']):.3f} points; OA oracle {m_o['oa_score']:.4f} hip {m_h['oa_score']:.4f}; {time.time() - t0:.0f}s")

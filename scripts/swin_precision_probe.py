"""bf16 vs f32 parameter gradients of SwinTUperNet on the HIP path, tensor by tensor in model order (where does rounding noise enter?)"""
import sys
import torch
sys.path.insert(0, ".")
from cvcs_amd import nets, utils  # noqa: E402
from oracle import swin_upernet_oracle as W, unet_oracle as O  # noqa: E402

NC, B, S = 16, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 128
img, lab = O.synthetic_tiles(B, S, NC, seed=9, structured=True)
out = {}
for prec in ("fp32", "bf16"):
    net = nets.SwinTUperNet(NC, prec)
    sd = W.init_params(NC, seed=3)
    if len(sys.argv) > 2:     # gate off pyramid-pooling branches (their BatchNorm gamma = 0): "0" = the 1x1 branch, "0123" = all four
        for i in sys.argv[2]:
            sd[f"decode_head.psp_modules.{i}.1.batch_norm.weight"].zero_()
    net.load_state_dict(sd, strict=False)
    net = net.to("cuda:0")
    crit = utils.CrossEntropyLoss(ignore_index=0)
    net.train()
    loss = crit(net(img.to("cuda:0"), None), lab.to("cuda:0"))
    loss.backward()
    torch.cuda.synchronize()
    out[prec] = {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}
for k in out["fp32"]:
    a, b = out["bf16"][k], out["fp32"][k]
    print(f"{((a - b).norm() / b.norm().clamp_min(1e-30)).item():10.3e}  |f32| {b.norm().item():.3e}  |bf16| {a.norm().item():.3e}  {k}")

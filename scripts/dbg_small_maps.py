import torch, torch.nn.functional as F
from cvcs_amd import ops
DEV = "cuda:0"
def nhwc(x, dt): return x.permute(0, 2, 3, 1).contiguous().to(dt).to(DEV).contiguous()
def back(t): return t.float().cpu().permute(0, 3, 1, 2).contiguous()
g = torch.Generator().manual_seed(0)
for dt in (torch.float32, torch.bfloat16):
    for B, H in ((4, 4), (3, 8), (4, 2), (4, 6)):
        C = 256
        x = torch.randn(B, C, H, H, generator=g).to(dt).float()
        w = (torch.randn(C, C, 3, 3, generator=g) / 48).to(dt).float()
        ref = F.conv2d(x, w, None, 1, 1)
        wf, wd = ops.pack_conv_weight(w.to(DEV), C, dt)
        y = torch.empty(B, H, H, C, dtype=dt, device=DEV)
        ops.conv2d(ops.view(nhwc(x, dt)), wf, None, ops.view(y), 3, 3, 1, 1)
        go = torch.randn(B, C, H, H, generator=g).to(dt).float()
        dx_ref = torch.nn.grad.conv2d_input(x.shape, w, go, padding=1)
        dx = torch.empty(B, H, H, C, dtype=dt, device=DEV)
        ops.conv2d(ops.view(nhwc(go, dt)), wd, None, ops.view(dx), 3, 3, 1, 1, 1)
        torch.cuda.synchronize()
        e1 = ((back(y) - ref).norm() / ref.norm()).item()
        e2 = ((back(dx) - dx_ref).norm() / dx_ref.norm()).item()
        print(dt, B, H, "fwd", f"{e1:.2e}", "dgrad", f"{e2:.2e}")

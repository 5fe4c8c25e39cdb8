import torch, torch.nn.functional as F
from cvcs_amd import ops
DEV = "cuda:0"
def nhwc(x, dt): return x.permute(0, 2, 3, 1).contiguous().to(dt).to(DEV).contiguous()
def back(t): return t.float().cpu().permute(0, 3, 1, 2).contiguous()
g = torch.Generator().manual_seed(0)
for dt in (torch.float32,):
    for B, H, C, mode in ((4, 4, 256, 0), (3, 8, 256, 0), (4, 4, 960, 3), (4, 4, 96, 2), (4, 8, 256, 0), (4, 4, 128, 0), (4, 4, 512, 0)):
        y = (torch.randn(B, C, H, H, generator=g) * 2 + 0.5).requires_grad_(True)
        gamma = (0.5 + torch.rand(C, generator=g)).requires_grad_(True)
        beta = (torch.rand(C, generator=g) - 0.5).requires_grad_(True)
        z = F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5)
        out = F.relu(z) if mode == 0 else F.hardswish(z) if mode == 3 else z
        go = torch.randn(out.shape, generator=g)
        out.backward(go)
        M_ = B * H * H
        yd, gd = nhwc(y.detach(), dt), nhwc(go, dt)
        mean = y.detach().mean(dim=(0, 2, 3)); invstd = 1.0 / torch.sqrt(y.detach().var(dim=(0, 2, 3), unbiased=False) + 1e-5)
        scale = (gamma.detach() * invstd).to(DEV); shift = (beta.detach() - mean * gamma.detach() * invstd).to(DEV)
        rows = ops.bn_bwd_rows(M_)
        p0, p1, p2 = (torch.zeros(rows * C, device=DEV) for _ in range(3))
        md, isd = mean.to(DEV), invstd.to(DEV)
        ops.bn_bwd_reduce(ops.view(yd), ops.view(gd), None, scale, shift, md, isd, mode, p0, p1)
        dgamma, dbeta, ca, cb = (torch.empty(C, device=DEV) for _ in range(4))
        ops.bn_bwd_finalize(p0, p1, rows, M_, C, gamma.detach().to(DEV), isd, dgamma, dbeta, ca, cb)
        dy = torch.empty_like(yd)
        ops.bn_bwd_apply(ops.view(yd), ops.view(gd), None, scale, shift, md, isd, ca, cb, mode, ops.view(dy), p2)
        torch.cuda.synchronize()
        r = lambda a, b: ((a - b).norm() / b.norm()).item()
        print(B, H, C, mode, "rows", rows, "dgamma", f"{r(dgamma.cpu(), gamma.grad):.2e}", "dbeta", f"{r(dbeta.cpu(), beta.grad):.2e}", "dy", f"{r(back(dy), y.grad):.2e}")

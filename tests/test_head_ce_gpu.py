"""1x1 head + cross-entropy + their backward in ONE pass (cvcs_head_ce) and the lazy logits handle that lets the reference's loop
`logits = net(x); loss = criterion(logits, y); loss.backward()` (S/train.py:121-125) reach it: against float64 autograd of
nn.Conv2d(C, NC, 1) -> nn.CrossEntropyLoss(weight, ignore_index) (S/nets.py:172, S/utils.py:230,238) on the host."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cvcs_amd import nets, ops, utils  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16


def close(got, ref, rel, what=""):
    scale = max(1e-30, ref.abs().max().item())
    err = (got.double() - ref.double()).abs().max().item()
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.2e} > {rel})"


CASES = [
    # B, H, W, C, NC, ignore_index, class weights, u8 target
    (2, 32, 32, 16, 16, 0, False, True),
    (1, 7, 9, 16, 5, 0, False, False),        # 63 pixels: a ragged last 16-pixel tile, int64 target
    (2, 24, 16, 32, 21, -100, True, True),    # two class tiles, class weights, nothing ignored
    (1, 16, 48, 64, 16, 0, True, True),       # the reference U-Nets' 64-channel head
    (3, 20, 20, 16, 21, 3, True, False),
    (1, 64, 64, 32, 2, 0, False, True),
]


@pytest.mark.parametrize("case", CASES)
def test_head_and_cross_entropy_in_one_pass(case):
    B, H, W, C_, NC, ignore, weighted, u8 = case
    g = torch.Generator().manual_seed(C_ * 100 + NC)
    x = (torch.randn(B, H, W, C_, generator=g) * 2).to(BF).float()
    w = torch.randn(NC, C_, generator=g) / C_ ** 0.5
    b = torch.randn(NC, generator=g) * 0.1
    t = torch.randint(0, NC, (B, H, W), generator=g)
    cw = (torch.rand(NC, generator=g) + 0.5) if weighted else None
    # float64 autograd
    X = x.double().requires_grad_(True)
    Wt, Bt = w.double().requires_grad_(True), b.double().requires_grad_(True)
    logits = (X @ Wt.T + Bt).permute(0, 3, 1, 2)
    loss = F.cross_entropy(logits, t, weight=None if cw is None else cw.double(), ignore_index=ignore)
    loss.backward()
    # device
    xt = torch.zeros(B, H, W, C_ + 16, dtype=BF, device=DEV)
    xt[..., 8:8 + C_] = x.to(BF).to(DEV)
    xv = ops.View(xt, 8, C_)
    dx = ops.view(torch.zeros(B, H, W, C_, dtype=BF, device=DEV))
    P = B * H * W
    rows = ops.head_ce_rows(P)
    part = torch.empty(rows * (NC * C_ + NC), dtype=torch.float32, device=DEV)
    ws = torch.empty(ops.ce_workspace_floats(P), dtype=torch.float32, device=DEV)
    lo = torch.empty(1, dtype=torch.float32, device=DEV)
    tgt = (t.to(torch.uint8) if u8 else t).to(DEV)
    ops.head_ce(xv, w.to(DEV), b.to(DEV), tgt, None if cw is None else cw.to(DEV), ignore, 1.0, dx, part, ws, lo)
    gw = torch.empty(NC * C_ + NC, dtype=torch.float32, device=DEV)
    ops.colsum_finalize(part, rows, NC * C_ + NC, gw)
    torch.cuda.synchronize()
    assert abs(lo.item() - loss.item()) <= 2e-6 * max(1.0, abs(loss.item())), (lo.item(), loss.item())
    close(dx.torch().float().cpu(), X.grad, 2.0 ** -8, "dx")                       # (bf16 storage of the result)
    close(gw[:NC * C_].cpu().view(NC, C_), Wt.grad, 2e-5, "dW")
    close(gw[NC * C_:].cpu(), Bt.grad, 2e-5, "db")
    lo2, dx2, part2 = torch.empty_like(lo), ops.view(torch.zeros_like(dx.t)), torch.empty_like(part)
    ops.head_ce(xv, w.to(DEV), b.to(DEV), tgt, None if cw is None else cw.to(DEV), ignore, 1.0, dx2, part2, ws, lo2)
    assert torch.equal(lo, lo2) and torch.equal(dx.t, dx2.t) and torch.equal(part, part2), "not bitwise reproducible"


def test_all_pixels_ignored_gives_nan_loss_and_zero_gradient():
    """torch semantics (F.cross_entropy on an all-ignored target: NaN loss, zero gradient) - as cvcs_ce_fwd_bwd"""
    B, H, W, C_, NC = 1, 16, 16, 16, 5
    x = ops.view(torch.randn(B, H, W, C_, device=DEV).to(BF))
    dx = ops.view(torch.ones(B, H, W, C_, dtype=BF, device=DEV))
    P = B * H * W
    rows = ops.head_ce_rows(P)
    part = torch.ones(rows * (NC * C_ + NC), dtype=torch.float32, device=DEV)
    ws = torch.empty(ops.ce_workspace_floats(P), dtype=torch.float32, device=DEV)
    lo = torch.empty(1, dtype=torch.float32, device=DEV)
    ops.head_ce(x, torch.randn(NC, C_, device=DEV), torch.zeros(NC, device=DEV), torch.zeros(B, H, W, dtype=torch.uint8, device=DEV), None, 0, 1.0, dx, part, ws, lo)
    torch.cuda.synchronize()
    assert lo.item() != lo.item() and dx.t.abs().max().item() == 0 and part.abs().max().item() == 0


def _step(net, img, lab, crit):
    net.train()
    logits = net(img, None)
    loss = crit(logits, lab)
    net.zero_grad(set_to_none=True)
    loss.backward()
    torch.cuda.synchronize()
    return logits, loss.item(), net.flat_parameters()[1].clone()


@pytest.mark.parametrize("unit_grad", [True, False])
def test_lazy_logits_handle_through_the_reference_training_step(unit_grad, monkeypatch):
    """`logits = net(x); loss = criterion(logits, y); loss.backward()` on a ResNet-UNet in bf16: the handle stays unread (no head launch, no
    logits) until the criterion consumes it through cvcs_head_ce; loss and EVERY parameter gradient equal those of the eager route (head ->
    logits -> cross-entropy -> head backward) to f32 rounding; reading the logits afterwards still works; a scaled loss scales the gradients."""
    NC, B, S = 5, 2, 64
    img, lab = O.synthetic_tiles(B, S, NC, seed=3, structured=True)
    img, lab = img.to(DEV), lab.to(DEV)
    torch.manual_seed(0)
    net = nets.Resnet18Unet(NC, "bf16").to(DEV)
    crit = utils.CrossEntropyLoss(ignore_index=0, unit_grad=unit_grad)
    monkeypatch.setattr(nets, "LAZY_HEAD", False)
    logits_e, loss_e, grad_e = _step(net, img, lab, crit)
    assert type(logits_e) is torch.Tensor
    logits_e = logits_e.detach().clone()
    monkeypatch.setattr(nets, "LAZY_HEAD", True)
    net.train()
    logits = net(img, None)
    eng = net._engine
    assert type(logits) is nets.LazyLogits and eng._head_pending and tuple(logits.shape) == (B, NC, S, S) and logits.requires_grad
    assert eng._head_pending, "reading metadata ran the head"
    loss = crit(logits, lab)
    assert eng._head_done and eng._head_pending, "the criterion did not take the fused path"
    net.zero_grad(set_to_none=True)
    (loss * 1.0).backward() if unit_grad else (loss * 0.5).backward()
    torch.cuda.synchronize()
    grad = net.flat_parameters()[1].clone()
    k = 1.0 if unit_grad else 0.5
    assert abs(loss.item() - loss_e) <= 1e-6 * max(1.0, abs(loss_e))
    # every parameter gradient: the two routes differ by f32 summation order in the head only (and the bf16 rounding of the head's data gradient)
    num, den = (grad - k * grad_e).norm().item(), (k * grad_e).norm().item()
    assert num <= 2e-2 * den, (num, den)
    hw = eng._goff["segmentation_head.0.weight"]
    n_head = NC * eng.dec[-1] + NC
    close(grad[hw:hw + n_head].cpu(), k * grad_e[hw:hw + n_head].cpu(), 1e-4, "head dW | db")
    # the handle still materialises for a reader after the fused loss
    got = logits.detach().float().cpu()
    assert not eng._head_pending
    close(got, logits_e.cpu(), 1e-6, "logits read after the fused loss")
    with pytest.raises(RuntimeError):
        crit(logits, lab)                      # a second differentiable loss on a fused handle is refused, not silently wrong
    # eval mode and no_grad keep returning plain tensors
    net.eval()
    assert type(net(img, None)) is torch.Tensor


def test_folded_head_uses_separate_weights_for_the_data_gradient():
    """the reference U-Nets fold their last BatchNorm into the head (cvcs_head_fold): logits = (W diag(s)) r + (b + W t), the gradient handed
    on is the one w.r.t. the BatchNorm output z = s r + t, i.e. W^T dl with the REAL weights, dW | db come out w.r.t. the folded ones"""
    B, H, W, C_, NC = 2, 16, 24, 64, 16
    g = torch.Generator().manual_seed(9)
    r = torch.randn(B, H, W, C_, generator=g).relu().to(BF).float()
    w = torch.randn(NC, C_, generator=g) / 8
    b = torch.randn(NC, generator=g) * 0.1
    sc, sh = torch.rand(C_, generator=g) + 0.5, torch.randn(C_, generator=g) * 0.2
    t = torch.randint(0, NC, (B, H, W), generator=g)
    wf, bf = w * sc[None, :], b + w @ sh
    Z = (r.double() * sc.double() + sh.double()).requires_grad_(True)
    Wt, Bt = w.double().requires_grad_(True), b.double().requires_grad_(True)
    loss = F.cross_entropy((Z @ Wt.T + Bt).permute(0, 3, 1, 2), t, ignore_index=0)
    loss.backward()
    Wf, Bf = wf.double().requires_grad_(True), bf.double().requires_grad_(True)
    F.cross_entropy((r.double() @ Wf.T + Bf).permute(0, 3, 1, 2), t, ignore_index=0).backward()
    xv = ops.view(r.to(BF).to(DEV).contiguous())
    dx = ops.view(torch.zeros(B, H, W, C_, dtype=BF, device=DEV))
    P = B * H * W
    rows = ops.head_ce_rows(P)
    part = torch.empty(rows * (NC * C_ + NC), dtype=torch.float32, device=DEV)
    ws = torch.empty(ops.ce_workspace_floats(P), dtype=torch.float32, device=DEV)
    lo = torch.empty(1, dtype=torch.float32, device=DEV)
    ops.head_ce(xv, wf.to(DEV), bf.to(DEV), t.to(torch.uint8).to(DEV), None, 0, 1.0, dx, part, ws, lo, w_dx=w.to(DEV))
    gw = torch.empty(NC * C_ + NC, dtype=torch.float32, device=DEV)
    ops.colsum_finalize(part, rows, NC * C_ + NC, gw)
    torch.cuda.synchronize()
    assert abs(lo.item() - loss.item()) <= 2e-6 * max(1.0, abs(loss.item()))
    close(dx.torch().float().cpu(), Z.grad, 2.0 ** -8, "gradient w.r.t. the BatchNorm output")
    close(gw[:NC * C_].cpu().view(NC, C_), Wf.grad, 2e-5, "dW (folded)")
    close(gw[NC * C_:].cpu(), Bf.grad, 2e-5, "db (folded)")


def test_lazy_logits_handle_on_the_reference_unet():
    """the reference's own model (S/nets.py:117-199) through its own loop: Unetv2 in bf16, the folded head"""
    NC, B, S = 16, 2, 64
    img, lab = O.synthetic_tiles(B, S, NC, seed=3, structured=True)
    img, lab = img.to(DEV), lab.to(DEV)
    torch.manual_seed(0)
    net = nets.Urnetv2(NC, "bf16").to(DEV)
    crit = utils.CrossEntropyLoss(ignore_index=0)
    old = nets.LAZY_HEAD
    try:
        nets.LAZY_HEAD = False
        logits_e, loss_e, grad_e = _step(net, img, lab, crit)
        logits_e = logits_e.detach().clone()
        nets.LAZY_HEAD = True
        net.train()
        logits = net(img, None)
        eng = net._engine
        assert type(logits) is nets.LazyLogits and eng._head_pending
        loss = crit(logits, lab)
        assert eng._head_done
        net.zero_grad(set_to_none=True)
        loss.backward()
        torch.cuda.synchronize()
        grad = net.flat_parameters()[1].clone()
    finally:
        nets.LAZY_HEAD = old
    assert abs(loss.item() - loss_e) <= 1e-6 * max(1.0, abs(loss_e))
    assert (grad - grad_e).norm().item() <= 2e-2 * grad_e.norm().item()
    hw = eng._goff["decode_forward4.1.weight"]
    close(grad[hw:hw + NC * 64 + NC].cpu(), grad_e[hw:hw + NC * 64 + NC].cpu(), 1e-4, "head dW | db after un-folding")
    close(logits.detach().float().cpu(), logits_e.cpu(), 1e-6, "logits read after the fused loss")

"""-m gpu: the reference's `SegformerMod` (S/nets.py:313-349) on the HIP path against oracle/segformer_oracle.py, which is PINNED against
transformers' own SegformerForSemanticSegmentation on the CPU (tests/test_segformer_oracle_cpu.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from cvcs_amd import nets, ops, utils  # noqa: E402
from oracle import segformer_oracle as SO  # noqa: E402
from oracle import unet_oracle as U  # noqa: E402

DEV = "cuda:0"


def _net(NC, precision, seed=3, stochastic=False, gen_seed=0):
    net = nets.SegformerMod(NC, precision, stochastic=stochastic, seed=gen_seed)
    missing, unexpected = net.load_state_dict(SO.init_params(NC, seed=seed), strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    return net.to(DEV)


def _tiles(B, S, NC, seed=5):
    return U.synthetic_tiles(B, S, NC, seed=seed, structured=True)


def _rel(a, b):
    return (a - b).abs().max().item() / max(1e-6, b.abs().max().item())


@pytest.mark.parametrize("NC,S,B", [(5, 64, 2), (16, 128, 2), (21, 96, 1)])
def test_fp32_eval_logits_and_labels_match_the_oracle(NC, S, B):
    """eval mode (BatchNorm folded, no DropPath / Dropout): NC = 5 / 16 / 21 padded to 32 channels; S = 64 ... 128 gives 4 ... 16 keys"""
    net = _net(NC, "fp32")
    img, _ = _tiles(B, S, NC)
    net.eval()
    with torch.no_grad():
        got = net(img.to(DEV), None).cpu()
        want = SO.forward(SO.init_params(NC, seed=3), img.float(), train=False)
    assert got.shape == (B, NC, S, S)
    assert _rel(got, want) <= 2e-4
    lab = net.predict_labels(img.to(DEV)).cpu()
    ref = want.argmax(1)
    diff = lab.long() != ref
    if diff.any():      # only on near-ties of the oracle's own top two
        top2 = want.topk(2, dim=1).values
        assert ((top2[:, 0] - top2[:, 1])[diff] <= 4e-4 * want.abs().max()).all()


def test_fp32_train_step_every_gradient_matches_the_oracle():
    """one train step without the stochastic layers: logits, loss, all 208 parameter gradients (the ConvTranspose / padded-channel layers come
    back through the scatter table in the reference's layouts), BatchNorm running statistics"""
    NC, S, B = 5, 64, 4
    net = _net(NC, "fp32")
    img, lab = _tiles(B, S, NC)
    crit = utils.CrossEntropyLoss()
    net.train()
    logits = net(img.to(DEV), None)
    loss = crit(logits, lab.to(DEV).long())
    loss.backward()
    p = SO.init_params(NC, seed=3)
    names = [k for k in p if not SO.is_buffer(k)]
    for k in names:
        p[k].requires_grad_(True)
    z = SO.forward(p, img.float(), train=True)
    lo = U.cross_entropy(z, lab.long(), None, -100)
    grads = dict(zip(names, torch.autograd.grad(lo, [p[k] for k in names])))
    assert _rel(logits.detach().cpu(), z.detach()) <= 2e-4
    assert abs(loss.item() - lo.item()) <= 1e-5 * max(1.0, abs(lo.item()))
    got = {k: v.grad.detach().cpu() for k, v in net.named_parameters()}
    assert set(got) == set(names)
    gmax = max(g.abs().max().item() for g in grads.values())
    worst = ("", 0.0)
    for k in names:
        # (per tensor, with a floor: the biases in front of the BatchNorm and the key biases have mathematically zero gradients)
        e = (got[k] - grads[k]).abs().max().item() / max(grads[k].abs().max().item(), 1e-4 * gmax)
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] <= 2e-3, worst
    sd = net.state_dict()
    for k in ("segformer.decode_head.batch_norm.running_mean", "segformer.decode_head.batch_norm.running_var"):
        assert torch.allclose(sd[k].cpu(), p[k], atol=1e-5, rtol=1e-4), k


def test_fp32_three_steps_with_drop_path_and_dropout_replayed_on_the_host():
    """train mode as the reference runs it (net.train(): DropPath 0 ... 0.1, Dropout 0.1): the per-sample DropPath factors and the dropout mask
    of each step are replayed on the host from the generator's (seed, step) and handed to the oracle; three SGD2 steps"""
    NC, S, B, seed = 5, 64, 8, 1234
    net = _net(NC, "fp32", stochastic=True, gen_seed=seed)
    tr = SO.OracleTrainer(NC, "SGD2", seed=3)
    img, lab = _tiles(B, S, NC)
    crit = utils.CrossEntropyLoss()
    opt, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 20}, net)
    net.train()
    cfg = SO.CONFIGS["b0"]
    rates2 = [r for r in SO.drop_path_rates(cfg) for _ in (0, 1)]
    h1 = S // 4
    dropped = 0
    for step in range(3):
        sc = ops.drop_path_scales_host(seed, step, rates2, B)
        dp = [sc[i] for i, r in enumerate(rates2) if r > 0]
        dropped += sum(int((s == 0).sum()) for s in dp)
        keep = ops.dropout_mask_host(seed, step, B * h1 * h1 * cfg["decoder"], cfg["classifier_dropout"]).view(B, h1, h1, cfg["decoder"]).permute(0, 3, 1, 2)
        masks = dict(drop_path=dp, dropout=keep.float() / (1 - cfg["classifier_dropout"]))
        logits = net(img.to(DEV), None)
        loss = crit(logits, lab.to(DEV).long())
        opt.zero_grad()
        loss.backward()
        opt.step()
        lo, z, _ = tr.step(img, lab, masks=masks)
        tol = (2e-4, 1e-5) if step == 0 else (5e-3, 1e-3)
        assert _rel(logits.detach().cpu(), z) <= tol[0], (step, _rel(logits.detach().cpu(), z))
        assert abs(loss.item() - lo) <= tol[1] * max(1.0, abs(lo)), (step, loss.item(), lo)
    assert dropped > 0, "the schedule must actually drop some samples"
    assert net._engine.drop_state.tolist() == [seed, 3]
    worst = max(_rel(v.detach().cpu(), tr.p[k].detach()) for k, v in net.named_parameters())
    assert worst <= 5e-3, worst


def test_bf16_step_within_the_emulation_noise_floor():
    """bf16 storage: one train step against the f32 oracle, bounded by 2 x the distance of the oracle's own bf16 emulation from it"""
    NC, S, B = 16, 128, 4
    net = _net(NC, "bf16")
    img, lab = _tiles(B, S, NC)
    crit = utils.CrossEntropyLoss()
    net.train()
    logits = net(img.to(DEV), None)
    loss = crit(logits, lab.to(DEV).long())
    loss.backward()
    with torch.no_grad():
        z = SO.forward(SO.init_params(NC, seed=3), img.float(), train=True)
        ze = SO.forward(SO.init_params(NC, seed=3), img.float(), train=True, emulate_bf16=True)
    floor = _rel(ze, z)
    err = _rel(logits.detach().cpu(), z)
    assert err <= 2.0 * floor + 1e-3, (err, floor)
    lo = U.cross_entropy(z, lab.long(), None, -100).item()
    assert abs(loss.item() - lo) <= 3e-2 * max(1.0, abs(lo))
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())


def test_checkpoints_in_the_4x_module_layout_and_through_custom_load(tmp_path):
    """S/nets.py:351-356 + S/utils.py:282-299: a checkpoint the reference wrote (DataParallel `module.` prefix, transformers 4.x names) loads
    through `wrapper` / `custom_load`; what this package saves carries the 5.x names of the installed library"""
    NC = 4
    net = utils.load_network({"net": "SegformerMod", "num_classes": NC - 1, "precision": "bf16"}, DEV)
    assert net.wrapper and net.returns_logits and not net.requires_context
    sd = {k: v for k, v in net.state_dict().items()}
    back = {
        "segformer.segformer.stages.0.patch_embeddings.proj.weight": "segformer.segformer.encoder.patch_embeddings.0.proj.weight",
        "segformer.segformer.stages.1.blocks.1.attention.q_proj.bias": "segformer.segformer.encoder.block.1.1.attention.self.query.bias",
        "segformer.segformer.stages.0.blocks.0.attention.sequence_reduction.sequence_reduction.weight": "segformer.segformer.encoder.block.0.0.attention.self.sr.weight",
        "segformer.segformer.stages.2.blocks.0.mlp.fc1.weight": "segformer.segformer.encoder.block.2.0.mlp.dense1.weight",
        "segformer.segformer.stages.3.layer_norm.bias": "segformer.segformer.encoder.layer_norm.3.bias",
        "segformer.decode_head.linear_projections.2.proj.weight": "segformer.decode_head.linear_c.2.proj.weight",
    }
    assert all(k in sd for k in back) and tuple(sd["seq.0.weight"].shape) == (NC, NC, 8, 8) and tuple(sd["seq.4.weight"].shape) == (NC, NC, 3, 3)
    old = {}
    for k, v in sd.items():
        k4 = k
        for new, o in back.items():
            if k == new:
                k4 = o
        old["module." + k4] = (v.float() * 0 + 0.125 if v.dtype == torch.float32 else v).cpu()
    path = str(tmp_path / "checkpoint1")
    torch.save({"model_state_dict": old}, path)
    other = utils.load_network({"net": "SegformerMod", "num_classes": NC - 1, "precision": "bf16"}, DEV)
    utils.load_checkpoint({"load_checkpoint": path}, other)
    assert all((v == 0.125).all() for v in other.state_dict().values() if v.dtype == torch.float32)
    # and the loaded network runs
    img, _ = _tiles(2, 64, NC)
    other.eval()
    with torch.no_grad():
        out = other(img.to(DEV), None)
    assert out.shape == (2, NC, 64, 64) and torch.isfinite(out).all()


def test_mit_b3_the_architecture_of_the_reference_default_checkpoint():
    """`SegformerMod(classes)` with the reference's default pretrained=True fetches nvidia/segformer-b3 (S/nets.py:322-324): hidden 64 / 128 / 320 / 512,
    depths 3 / 4 / 18 / 3, 64-wide heads, decoder 768.  Same code, `variant="b3"`: f32 eval logits and one train step against the oracle; a bf16 step
    (the matrix-core attention kernels with two K-steps) inside the emulation floor"""
    NC, S, B = 6, 64, 2
    p0 = SO.init_params(NC, seed=4, variant="b3")
    net = nets.SegformerMod(NC, "fp32", stochastic=False, variant="b3")
    missing, unexpected = net.load_state_dict(p0, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    net = net.to(DEV)
    img, lab = _tiles(B, S, NC)
    net.eval()
    with torch.no_grad():
        got = net(img.to(DEV), None).cpu()
        want = SO.forward({k: v.clone() for k, v in p0.items()}, img.float(), train=False, variant="b3")
    assert _rel(got, want) <= 3e-4
    crit = utils.CrossEntropyLoss()
    net.train()
    logits = net(img.to(DEV), None)
    loss = crit(logits, lab.to(DEV).long())
    loss.backward()
    p = {k: v.clone() for k, v in p0.items()}
    names = [k for k in p if not SO.is_buffer(k)]
    for k in names:
        p[k].requires_grad_(True)
    z = SO.forward(p, img.float(), train=True, variant="b3")
    lo = U.cross_entropy(z, lab.long(), None, -100)
    grads = dict(zip(names, torch.autograd.grad(lo, [p[k] for k in names])))
    assert _rel(logits.detach().cpu(), z.detach()) <= 3e-4 and abs(loss.item() - lo.item()) <= 1e-5 * max(1.0, abs(lo.item()))
    gmax = max(g.abs().max().item() for g in grads.values())
    worst = max(((v.grad.detach().cpu() - grads[k]).abs().max().item() / max(grads[k].abs().max().item(), 1e-4 * gmax), k) for k, v in net.named_parameters())
    assert worst[0] <= 3e-3, worst
    # bf16 at a size whose key counts (S = 256: 64 keys) run the matrix-core kernels
    nb = nets.SegformerMod(NC, "bf16", stochastic=False, variant="b3")
    nb.load_state_dict(p0, strict=False)
    nb = nb.to(DEV)
    img2, lab2 = _tiles(2, 256, NC, seed=9)
    nb.train()
    lg = nb(img2.to(DEV), None)
    crit(lg, lab2.to(DEV).long()).backward()
    with torch.no_grad():
        z2 = SO.forward({k: v.clone() for k, v in p0.items()}, img2.float(), train=True, variant="b3")
        ze = SO.forward({k: v.clone() for k, v in p0.items()}, img2.float(), train=True, variant="b3", emulate_bf16=True)
    floor = _rel(ze, z2)
    assert _rel(lg.detach().cpu(), z2) <= 2.0 * floor + 1e-3, (_rel(lg.detach().cpu(), z2), floor)
    assert all(torch.isfinite(q.grad).all() for q in nb.parameters())


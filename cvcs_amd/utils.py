"""The reference's factory / evaluation surface (source/scripts/utils.py), re-implemented over the HIP path.

Same names, argument meaning and error behaviour as the reference for the functions on the hot path:
`load_device`, `load_network`, `load_loss`, `load_optimizer`, `count_params`, `mask_reshape`,
`validation_loss`, `eval_model`, the metric helpers and `print_metrics`.  Quirks 1-2 of SURVEY section 5
(`load_device` crashing on cpu, `SGD1` always raising) are fixed; the config keys are unchanged and three optional
keys are added: `precision` ('bf16' default | 'fp32'), `eval_batch_size`, `world_size` (informational).
"""
from __future__ import annotations

import os

import torch

from . import nets, ops
from .optim import FusedAdam, FusedSGD, PolynomialLR

labels = {
    0: "unlabeled", 1: "industrial land", 2: "urban residential", 3: "rural residential", 4: "traffic land",
    5: "paddy field", 6: "irrigated cropland", 7: "dry cropland", 8: "garden plot", 9: "arbor forest",
    10: "shrub land", 11: "natural grassland", 12: "artificial grassland", 13: "river", 14: "lake", 15: "pond",
}


# ---------------------------------------------------------------------------------------------------- factories
def load_device(config):
    """S/utils.py:273-280 ('gpu' -> 'cuda:0'); the cpu branch no longer calls torch.cuda.get_device_name."""
    if config["device"] == "gpu":
        assert torch.cuda.is_available(), "Notebook is not configured properly!"
        # one process per GPU under torch.distributed.run; CVCS_ONE_DEVICE=1 rehearses N ranks on a one-GPU box
        local = os.environ.get("LOCAL_RANK", "0") if os.environ.get("CVCS_ONE_DEVICE") != "1" else "0"
        device = f"cuda:{local}"
        print("Training network on {}".format(torch.cuda.get_device_name(device=device)))
    else:
        device = torch.device("cpu")
        print("Training network on cpu")
    return device


def load_network(config, device):
    """S/utils.py:174-195, every network on the HIP kernels: the reference's `Unet` / `Unetv2` and `Resnet101` (DeepLabV3 on a dilated
    ResNet-101, torchvision key names), `Ensemble` (votes over HIP member networks), BASELINE.json's `Resnet18Unet` / `Resnet34Unet` /
    `Resnet50Unet`, `DeepLabV3Plus` and `TSwin` (Swin-T + UPerNet), and the reference's `SegformerMod` (S/utils.py:191-192: transformers'
    Segformer + the reference's ConvTranspose tail; config keys `segformer_variant` "b0" | "b3", `stochastic` false = DropPath / Dropout off,
    `seed`) and `MobileNet` (S/utils.py:183-184: torchvision's DeepLabV3 on a dilated MobileNetV3-large, torchvision key names)."""
    netname = config["net"]
    classes = config["num_classes"] + 1
    precision = config.get("precision", "bf16")
    if netname == "Unet":
        return nets.Urnet(classes, precision).to(device)
    elif netname == "Unetv2":
        return nets.Urnetv2(classes, precision).to(device)
    elif netname in ("Resnet18Unet", "Resnet34Unet", "Resnet50Unet"):
        # BASELINE.json's ResNet-encoder U-Nets: new factory names behind the same seam (the reference has no such model)
        return getattr(nets, netname)(classes, precision, config.get("decoder_norm", "bn_relu"), config.get("decoder_channels")).to(device)
    elif netname == "Ensemble":
        try:
            return Ensemble(classes, device, config.get("ensemble_config"), precision=precision)
        except Exception:
            print("Some error occured when loading ensemble!")
            raise
    elif netname in ("TSwin", "SwinTUperNet"):
        # the reference documents `TSwin` (configs/train/README.txt:44) without a definition: Swin-T + UPerNet of BASELINE configs[3]
        return nets.SwinTUperNet(classes, precision).to(device)
    elif netname in ("BSwin", "SwinBUperNet"):
        return nets.SwinBUperNet(classes, precision).to(device)   # configs/train/README.txt:45
    elif netname == "DeepLabV3Plus":
        # BASELINE.json configs[2]: DeepLabV3+ (ResNet-50, ASPP 6/12/18) - a new factory name behind the same seam
        return nets.DeepLabV3Plus(classes, precision, config.get("aspp_dropout"), config.get("seed", 0)).to(device)
    elif netname == "Resnet101":
        # S/utils.py:180-181 -> nets.DeepLabv3Resnet101: DeepLabV3 on a dilated ResNet-101, on the HIP kernels (random-init:
        # the reference's COCO weights come from a network fetch)
        return nets.DeepLabv3Resnet101(classes, precision, config.get("aspp_dropout"), config.get("seed", 0)).to(device)
    elif netname == "SegformerMod":
        # S/utils.py:191-192 -> nets.SegformerMod (random-init MiT-b0 = the reference's pretrained=False branch; its default fetches b3 weights)
        return nets.SegformerMod(classes, precision, config.get("stochastic", True), config.get("seed", 0), config.get("segformer_variant")).to(device)
    elif netname == "MobileNet":
        # S/utils.py:183-184 -> nets.DeepLabV3MobileNet: torchvision's deeplabv3_mobilenet_v3_large on the HIP kernels (random-init: the
        # reference's COCO weights come from a network fetch)
        return nets.DeepLabV3MobileNet(classes, precision, config.get("aspp_dropout"), config.get("seed", 0)).to(device)
    else:
        print("Invalid network name.")
        raise Exception


def count_params(net):
    return sum(p.numel() for p in net.parameters() if p.requires_grad)


class _CEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, crit):
        B, NC = logits.shape[:2]
        P = logits.numel() // NC
        dev = logits.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        if crit._ws is None or crit._ws.numel() < ops.ce_workspace_floats(P) or crit._ws.device != dev:
            crit._ws = torch.empty(ops.ce_workspace_floats(P), dtype=torch.float32, device=dev)
        ctx.crit, ctx.lazy = crit, None
        lz = nets.lazy_state(logits)
        if lz is not None and crit.sync is None:
            # the logits are an unread handle (nets.LazyLogits): head + cross-entropy + their backward in ONE pass over the last decoder
            # activation - neither the logits nor their gradient are ever written
            w = crit.weight
            if w is not None and w.device != dev:
                w = crit.weight = w.to(dev)
            lz["fused"] = True
            lz["engine"].head_ce(target.contiguous(), w, crit.ignore_index, crit._ws, loss)
            ctx.lazy, ctx.shape = lz, tuple(logits.shape)
            return loss[0]
        if nets.lazy_fused(logits) and logits.requires_grad:
            raise RuntimeError("a second differentiable loss on logits whose first cross-entropy was fused with the head (nets.LazyLogits): "
                               "set CVCS_LAZY_HEAD=0 to keep eager logits")
        need_grad = logits.requires_grad
        if need_grad and (crit._dl is None or crit._dl.shape != logits.shape or crit._dl.device != dev):
            crit._dl = torch.empty_like(logits)
        w = crit.weight
        if w is not None and w.device != dev:
            w = crit.weight = w.to(dev)
        logits, target = logits.contiguous(), target.contiguous()
        if crit.sync is None:
            ops.ce_fwd_bwd(logits, target, w, crit.ignore_index, 1.0, loss, crit._dl if need_grad else None, crit._ws)
        else:
            # tiles of one batch on several ranks: the mean runs over the non-ignored pixels of ALL ranks.  The gradient
            # carries a factor world (the optimiser averages the summed gradients); the returned value is the whole
            # batch's loss, identical on every rank.
            ops.ce_weight_sum(target, B, NC, w, crit.ignore_index, crit._ws)
            crit.sync.all_reduce(crit._ws[:1])
            ops.ce_fwd_bwd(logits, target, w, crit.ignore_index, float(crit.sync.world), loss,
                           crit._dl if need_grad else None, crit._ws, external_denominator=True)
            crit.sync.all_reduce(loss)
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        if ctx.lazy is not None:
            # the head's gradients already exist inside the engine (head_ce); autograd only needs a tensor of the logits' shape to hand on
            if not ctx.crit.unit_grad:
                ctx.lazy["engine"].scale_head_grads(gout)
            return torch.zeros(1, dtype=torch.float32, device=gout.device).expand(ctx.shape), None, None
        dl = ctx.crit._dl   # d(loss)/d(logits) was produced by the forward launch
        if not ctx.crit.unit_grad:
            # chained through further autograd ops: scale by the incoming gradient - compared with 1 ON THE DEVICE (no host round trip)
            ops.scale_unless_one(dl, gout.detach().to(device=dl.device, dtype=torch.float32).reshape(1).contiguous())
        return dl, None, None


class CrossEntropyLoss:
    """nn.CrossEntropyLoss(weight, ignore_index) (S/utils.py:230,238): one fused HIP launch produces the loss and
    d(loss)/d(logits).  target: uint8 or int64 [B,H,W].
    unit_grad=True asserts that the loss is differentiated directly - `loss.backward()` with its implicit gradient of 1, as
    S/train.py:125 does - and skips the check; by default the incoming gradient is compared with 1 on the device and the
    logits-sized gradient is only rescaled (one extra pass) when the loss was scaled / combined by further autograd ops
    (`(loss / accum).backward()`), so a drop-in use can never silently produce unscaled gradients.  load_loss opts in to
    unit_grad for the training scripts of this repository."""

    def __init__(self, weight=None, ignore_index=-100, unit_grad=False):
        self.weight = None if weight is None else weight.float()
        self.ignore_index = ignore_index
        self.unit_grad = unit_grad
        self.sync = None    # parallel.SyncStats when the batch is sharded over ranks with exact=True
        self._ws = None
        self._dl = None

    def __call__(self, logits, target):
        if target.dtype not in (torch.uint8, torch.int64):
            target = target.long()
        return _CEFunction.apply(logits, target, self)


class MSELoss:
    """nn.MSELoss() (S/utils.py:239-240).  The reference builds it but cannot train with it (its loop hands it an int64
    [B,H,W] target against [B,C,H,W] logits, S/train.py:122); kept for the factory surface: mean squared error with torch's
    broadcasting rules, evaluated on the device by elementwise torch ops (not a hot-path kernel), differentiable through autograd."""

    def __call__(self, pred, target):
        return torch.nn.functional.mse_loss(pred, target.to(pred.dtype))


def load_loss(config, device, dataset=None):
    """S/utils.py:223-242: 'CEL' | 'wCEL' (class-balanced weights from the Loader) ; ignore_index 0 or -100."""
    classes = config["num_classes"] + 1
    name = config["loss"]
    ignore_background = config.get("ignore_background", False)
    ignore_index = 0 if ignore_background else -100
    if name == "CEL":
        return CrossEntropyLoss(ignore_index=ignore_index, unit_grad=True)   # S/train.py:125: loss.backward() directly
    elif name == "wCEL":
        print("Computing class weights, it might take several minutes...", flush=True)
        weights = dataset.get_class_weights(classes, ignore_background).to(device)
        for i, score in enumerate(weights):
            print(f"{labels.get(i, i):>22s} {score.item():.6f}")
        return CrossEntropyLoss(weight=weights, ignore_index=ignore_index, unit_grad=True)
    elif name == "MSE":
        return MSELoss()                         # S/utils.py:239-240 (nn.MSELoss(); the reference never trains with it)
    else:
        raise Exception


def load_optimizer(config, net):
    """S/utils.py:208-221 (SGD1 reachable again: the reference's if/if/elif chain always raised for it)."""
    optimizer = config["opt"]
    if optimizer == "SGD1":
        opt = FusedSGD(net, lr=0.001, momentum=0.9, weight_decay=0.00001)
        sched = PolynomialLR(opt)
    elif optimizer == "SGD2":
        opt = FusedSGD(net, lr=0.006, momentum=0.9, weight_decay=0.00001)
        sched = PolynomialLR(opt, total_iters=20)
    elif optimizer == "ADAM1":
        opt = FusedAdam(net, lr=0.005)
        sched = PolynomialLR(opt, total_iters=config["epochs"], power=2.0)
    else:
        raise ValueError("Optimizer name not valid.")
    return opt, sched


def save_model(epoch, net, opt, scheduler, train_loss, val_loss, macro_precision, weighted_precision, conf_flat,
               conf_normalized, batch_size, checkpoint_dir, optimizer):
    """S/utils.py:128-142: same dictionary keys and file name (`checkpoint<epoch+1>`); `model_state_dict` has the
    reference's parameter names / shapes, so either side loads the other's weights."""
    import os
    torch.save({
        "epoch": epoch, "model_state_dict": net.state_dict(), "optimizer_state_dict": opt.state_dict(),
        "scheduler_state_dict": scheduler.state_dict(), "training_loss_values": train_loss,
        "validation_loss_values": val_loss, "batch_size": batch_size, "macro_precision": macro_precision,
        "weighted_precision": weighted_precision, "conf_flat": conf_flat, "conf_normalized": conf_normalized,
        "optimizer": optimizer,
    }, os.path.join(checkpoint_dir, "checkpoint{}".format(epoch + 1)))


def load_checkpoint(config, net, load_confusion=False):
    """S/utils.py:282-299.  Accepts checkpoints written by the reference (torch modules, possibly under a
    DataParallel `module.` prefix, S/nets.py:252-257) or by save_model above."""
    if "load_checkpoint" in config.keys():
        checkpoint = torch.load(config["load_checkpoint"], map_location="cpu", weights_only=False)
        if getattr(net, "wrapper", False):      # S/utils.py:286-287: the wrapper modules translate their own key names
            net.custom_load(checkpoint)
        else:
            sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in checkpoint["model_state_dict"].items()}
            net.load_state_dict(sd)
        TL, VL = checkpoint.get("training_loss_values", []), checkpoint.get("validation_loss_values", [])
        mIoU, wIoU = checkpoint.get("macro_precision", []), checkpoint.get("weighted_precision", [])
        print("Loaded checkpoint {}".format(config["load_checkpoint"]), flush=True)
        if load_confusion:
            return TL, VL, mIoU, wIoU, checkpoint.get("conf_flat"), checkpoint.get("conf_normalized")
        return TL, VL, mIoU, wIoU


def mask_reshape(mask: torch.Tensor):
    """S/utils.py:557-567: a 4-D mask carries replicated channels; keep channel 0."""
    if len(mask.shape) == 4:
        return mask[:, 0, :, :]
    return mask


class Ensemble(torch.nn.Module):
    """S/utils.py:472-507: several trained networks vote per pixel.  `config_file` (under configs/ensemble/, or an
    absolute path) maps network names to checkpoints; each member runs its HIP forward in eval mode, its argmax label map
    is taken on the device (ties -> first class) and the maps are merged by a per-pixel majority vote (torch.mode: ties ->
    smallest label) in one HIP launch.  Returns int64 labels ([H,W] for a single tile, [B,H,W] otherwise);
    `returns_logits` is False, as in the reference."""

    def __init__(self, num_classes, device, config_file, precision="bf16", config_path=None):
        super().__init__()
        if not config_file:
            print("To use the ensemble you have to specify a config file.")
            print("Add the 'ensemble_config' entry in your evaluation configuration file.")
            raise Exception
        self.requires_context = False
        self.num_classes = num_classes
        self.wrapper = False
        self.returns_logits = False
        self.config_path = os.path.abspath(config_path or "configs/ensemble/")
        self.config_file = config_file
        self.device = device
        self.precision = precision
        self._init_models()

    def _init_models(self):
        import yaml
        path = self.config_file if os.path.isabs(self.config_file) else os.path.join(self.config_path, self.config_file)
        with open(path, "r") as file:
            checkpoints = yaml.safe_load(file)
        items = checkpoints.items() if isinstance(checkpoints, dict) else [tuple(d.items())[0] for d in checkpoints]
        models = []
        for key, value in items:
            config = {"net": key, "load_checkpoint": value, "device": "gpu", "num_classes": self.num_classes - 1,
                      "precision": self.precision}
            model = load_network(config, self.device)
            load_checkpoint(config, model)
            models.append(model)
        assert 1 <= len(models) <= 16, "the vote kernel takes 1..16 member networks"
        self.models = torch.nn.ModuleList(models)

    @torch.no_grad()
    def forward(self, x: torch.Tensor, context=None):
        B, _, H, W = x.shape
        stack = torch.empty((len(self.models), B, H, W), dtype=torch.uint8, device=x.device)
        for m, net in enumerate(self.models):
            net.eval()
            stack[m] = net.predict_labels(x)
        out = torch.empty((B, H, W), dtype=torch.uint8, device=x.device)
        ops.vote_labels(stack, out)
        out = out.long()
        return out[0] if B == 1 else out


# ---------------------------------------------------------------------------------------------------- evaluation
class ConfusionMatrix:
    """device-side stand-in for torchmetrics' MulticlassConfusionMatrix (S/utils.py:76-78): rows = target,
    cols = prediction, pixels whose target == ignore_index dropped; `compute()` -> int64 [K,K] on the host."""

    def __init__(self, num_classes=16, ignore_index=None, device="cuda:0"):
        self.K = num_classes
        self.ignore_index = -1 if ignore_index is None else ignore_index
        self.conf = torch.zeros(num_classes, num_classes, dtype=torch.int64, device=device)
        self._labels = None

    def update_from_labels(self, pred, target):
        """pred: integer label map from a network with returns_logits = False (S/utils.py:89-94)"""
        p8 = pred.to(torch.uint8).contiguous()
        t = target if target.dtype in (torch.uint8, torch.int64) else target.long()
        ops.label_confusion(p8.view(-1), t.contiguous().view(-1), self.ignore_index, self.K, self.conf)
        return p8

    def update_from_logits(self, logits, target):
        B, NC = logits.shape[:2]
        if self._labels is None or self._labels.numel() != logits.numel() // NC:
            self._labels = torch.empty((B,) + tuple(logits.shape[2:]), dtype=torch.uint8, device=logits.device)
        t = target if target.dtype in (torch.uint8, torch.int64) else target.long()
        ops.argmax_confusion(logits.contiguous(), self._labels, t.contiguous(), self.ignore_index, self.K, self.conf)
        return self._labels

    def compute(self):
        return self.conf.cpu()


def eval_model(net, Loader_validation, device, batch_size=1, show_progress=False, ignore_background=False,
               num_classes=16, shard=None):
    """S/utils.py:59-103 with the per-tile logits D2H + CPU argmax replaced by one fused argmax+histogram launch
    (any batch size).  Returns (flat, normalized): `flat.compute()` is the int64 confusion matrix.
    shard=(rank, world[, group]): every rank evaluates the batches i with i % world == rank and the count matrices are
    summed over ranks once at the end (integer counts: identical to the single-process matrix)."""
    net.eval()
    ignored_index = 0 if ignore_background else None
    flat = ConfusionMatrix(num_classes=num_classes, ignore_index=ignored_index, device=device)
    i = 0
    with torch.no_grad():
        for c in range(len(Loader_validation)):
            for x, y, _, context in chunk_batches(Loader_validation, c, batch_size, device):
                i += 1
                if shard is not None and (i - 1) % shard[1] != shard[0]:
                    continue
                x, y = x.to(device), mask_reshape(y.to(device))
                tgt = y.squeeze(1) if y.dim() == 4 else y
                if hasattr(net, "predict_labels"):   # HIP network: head + argmax fused, the logits never leave the chip
                    flat.update_from_labels(net.predict_labels(x), tgt)
                    continue
                y_pred = net(x, None)
                if getattr(net, "returns_logits", True):
                    flat.update_from_logits(y_pred, tgt)
                else:   # the network already returns label indices (S/utils.py:89)
                    flat.update_from_labels(y_pred.reshape(tgt.shape), tgt)
    if shard is not None:
        import torch.distributed as dist
        dist.all_reduce(flat.conf, op=dist.ReduceOp.SUM, group=shard[2] if len(shard) > 2 else None)
    return flat, NormalizedConfusion(flat)


class NormalizedConfusion:
    """row-normalised view (torchmetrics normalize='true') of a ConfusionMatrix."""

    def __init__(self, flat):
        self.flat = flat

    def compute(self):
        c = self.flat.compute().double()
        return c / c.sum(dim=1, keepdim=True).clamp_min(1)


def _takes_device(loader) -> bool:
    """does this loader's get_iterable_chunk offer the device-resident route (keywords device / batch_size / shard)?  Decided from
    the signature, not by catching TypeError: an error raised INSIDE the device route must surface, not fall back silently"""
    import inspect
    try:
        names = inspect.signature(loader.get_iterable_chunk).parameters
    except (TypeError, ValueError):
        return False
    return all(k in names for k in ("device", "batch_size", "shard"))


def chunk_batches(loader, c, batch_size, device, random_tps=None, shard=None):
    """batches of chunk c on `device`.  On the GPU the loader keeps the chunk's decoded images (or tiles) resident and
    produces every batch with one gather launch (dataset.DeviceChunk / DeviceTiles - SURVEY section 8 f1); otherwise (and
    for foreign loaders without the `device` keyword) the reference's route: per-tile iterable -> DataLoader -> `.to(device)`
    (S/train.py:107-115).  shard=(rank, world): this rank's slice of every global batch of `batch_size` tiles."""
    if torch.device(device).type == "cuda" and _takes_device(loader):
        return loader.get_iterable_chunk(c, random_tps, device=device, batch_size=batch_size, shard=shard)
    ds = loader.get_iterable_chunk(c, random_tps) if random_tps is not None else loader.get_iterable_chunk(c)
    dl = torch.utils.data.DataLoader(ds, batch_size=batch_size, drop_last=shard is not None)
    if shard is None:
        return dl
    r, w = shard
    per = batch_size // w
    return ((b[0][r * per:(r + 1) * per], b[1][r * per:(r + 1) * per], b[2], b[3]) for b in dl)


def validation_loss(net, Loader_validation, crit, device, bs, show_progress=False):
    """S/utils.py:106-126."""
    loss_values = []
    net.eval()
    with torch.no_grad():
        for c in range(len(Loader_validation)):
            for image, index_mask, _, context in chunk_batches(Loader_validation, c, bs, device):
                image, mask = image.to(device), mask_reshape(index_mask.to(device))
                mask_pred = net(image, None)
                loss = crit(mask_pred, mask)
                loss_values.append(loss.item())
    return loss_values


# ---------------------------------------------------------------------------------------------------- metrics
def _get_class_scores(confusion, formula, ignore_condition):
    """S/utils.py:319-341."""
    scores, excluded = [], []
    _, classes = list(confusion.shape)
    for i in range(classes):
        tp = confusion[i, i].item()
        fp = (torch.sum(confusion[:, i]) - tp).item()
        fn = (torch.sum(confusion[i, :]) - tp).item()
        if ignore_condition(tp, fp, fn):
            scores.append(0)
            excluded.append(i)
        else:
            scores.append(formula(tp, fp, fn))
    return torch.tensor(scores), excluded


def _mean_excluding(scores, excluded):
    """S/utils.py:343-346."""
    inc = torch.tensor([x for i, x in enumerate(scores) if i not in excluded])
    return torch.mean(inc).item()


def _score(confusion, formula, ignore_condition, mean, return_excluded):
    scores, excluded = _get_class_scores(confusion, formula, ignore_condition)
    m = _mean_excluding(scores, excluded)
    if mean:
        return (m, excluded) if return_excluded else m
    return (scores, excluded) if return_excluded else m


def precision(confusion, macro=False, return_excluded=False):
    return _score(confusion, lambda tp, fp, fn: tp / (tp + fp), lambda tp, fp, fn: tp + fp == 0, macro, return_excluded)


def recall(confusion, macro=False, return_excluded=False):
    return _score(confusion, lambda tp, fp, fn: tp / (tp + fn), lambda tp, fp, fn: tp + fn == 0, macro, return_excluded)


def IoU(confusion, mean=False, return_excluded=False):
    return _score(confusion, lambda tp, fp, fn: tp / (tp + fn + fp), lambda tp, fp, fn: tp + fn == 0, mean, return_excluded)


def F1(confusion, mean=False, return_excluded=False):
    return _score(confusion, lambda tp, fp, fn: (2 * tp) / (2 * tp + fn + fp), lambda tp, fp, fn: tp + fn == 0, mean,
                  return_excluded)


def accuracy(confusion):
    _, classes = list(confusion.shape)
    return sum(confusion[i, i].item() for i in range(classes)) / torch.sum(confusion).item()


def print_metrics(confusion, silent=False):
    """S/utils.py:375-403 (plain-text tables: prettytable is not a dependency here)."""
    out = {"mIoU": IoU(confusion, mean=True), "precision_score": precision(confusion, macro=True),
           "recall_score": recall(confusion, macro=True), "dice_score": F1(confusion, mean=True),
           "oa_score": accuracy(confusion)}
    values, excluded = IoU(confusion, mean=False, return_excluded=True)
    out["perclass_IoU"] = values.tolist()
    if not silent:
        for k, name in (("mIoU", "mIoU"), ("precision_score", "mPrec"), ("recall_score", "mRec"),
                        ("dice_score", "Dice"), ("oa_score", "OA")):
            print(f"{name:>6s} {out[k]:.6f}")
        print(f"Excluded classes (not in target): {list(excluded)}")
        for i, s in enumerate(out["perclass_IoU"]):
            print(f"{labels.get(i, str(i)):>22s} {s:.6f}", flush=True)
    return out


def display_configs(configs):
    for key, value in configs.items():
        print(f"{key:>28s} : {value}", flush=True)

"""Probe (GPU): BatchNorm-backward apply pass (reads y, g; writes dy) at the level-1 shape, product build vs an ablation build
with the arithmetic removed (scripts/libbn_probe_copy.so: hipcc ... -DCVCS_PROBE_BN_COPY elementwise.hip api.hip), and a plain
torch copy of the same bytes for reference."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvcs_amd import ops, _lib
dev = "cuda:0"
here = os.path.dirname(os.path.abspath(__file__))
libs = {"product": _lib.lib()}
for name in ("libbn_probe_copy.so",):
    if os.path.exists(os.path.join(here, name)):
        h = C.CDLL(os.path.join(here, name)); h.cvcs_bn_bwd_apply.restype = C.c_int; h.cvcs_bn_bwd_apply.argtypes = _lib.SIGNATURES["cvcs_bn_bwd_apply"][1]
        h.cvcs_bn_bwd_reduce.restype = C.c_int; h.cvcs_bn_bwd_reduce.argtypes = _lib.SIGNATURES["cvcs_bn_bwd_reduce"][1]
        libs[name] = h
st = torch.cuda.current_stream().cuda_stream
def timeit(fn):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 100
for (B, S, Cc) in ((32, 512, 64), (32, 256, 128), (32, 128, 256)):
    y = torch.randn(B, S, S, Cc, device=dev).to(torch.bfloat16)
    g = torch.randn(B, S, S, Cc, device=dev).to(torch.bfloat16)
    dy = torch.empty_like(y)
    v = [torch.rand(Cc, device=dev) + 0.5 for _ in range(6)]
    rows = _lib.lib().cvcs_bn_bwd_rows(B * S * S)
    part, part2 = torch.empty(rows * Cc, device=dev), torch.empty(rows * Cc, device=dev)
    nb = y.numel() * 2
    for name, h in libs.items():
        ap = lambda: h.cvcs_bn_bwd_apply(y.data_ptr(), Cc, g.data_ptr(), Cc, None, 0, B, S, S, Cc, *[t.data_ptr() for t in v], 0, dy.data_ptr(), Cc, part.data_ptr(), 1, st)
        rd = lambda: h.cvcs_bn_bwd_reduce(y.data_ptr(), Cc, g.data_ptr(), Cc, None, 0, B, S, S, Cc, *[t.data_ptr() for t in v[:4]], 0, part.data_ptr(), part2.data_ptr(), 1, st)
        ta, tr = timeit(ap), timeit(rd)
        print(f"{name:22s} S={S} C={Cc}: apply {ta:.1f} us ({3 * nb / ta / 1e6:.2f} TB/s)  reduce {tr:.1f} us ({2 * nb / tr / 1e6:.2f} TB/s)", flush=True)
    tc = timeit(lambda: dy.copy_(y))
    print(f"torch copy             S={S} C={Cc}: {tc:.1f} us ({2 * nb / tc / 1e6:.2f} TB/s)", flush=True)

"""Probe (GPU): head backward / forward / cross entropy at B=32, S=512, NC=16 (per-launch time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctypes as C
from cvcs_amd import ops, _lib
dev = "cuda:0"
_pl = os.environ.get("PROBE_LIB")   # an ablation build of head_loss.hip (scripts/libhead_probe_*.so), same C-ABI
plib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), _pl)) if _pl else _lib.lib()
if _pl:
    plib.cvcs_head_bwd.restype = C.c_int
    plib.cvcs_head_bwd.argtypes = _lib.SIGNATURES["cvcs_head_bwd"][1]
B, S, NC = 32, 512, 16
x = torch.randn(B, S, S, 64, device=dev).to(torch.bfloat16)
dl = torch.randn(B, NC, S, S, device=dev) * 1e-3
w = torch.randn(NC, 64, device=dev) * 0.1
dx = torch.empty_like(x)
rows = _lib.lib().cvcs_head_bwd_rows(B * S * S)
part = torch.empty(rows * (NC * 64 + NC), device=dev)
st = torch.cuda.current_stream().cuda_stream
def bwd():
    _lib.check(plib.cvcs_head_bwd(x.data_ptr(), 64, dl.data_ptr(), B, S, S, 64, w.data_ptr(), NC, dx.data_ptr(), 64, part.data_ptr(), 1, st))
for _ in range(3): bwd()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): bwd()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 100
print(f"{_pl or 'product'} rows={rows} head_bwd {us:.1f} us  ({(x.numel() * 2 * 2 + dl.numel() * 4) / us / 1e6:.2f} TB/s)", flush=True)

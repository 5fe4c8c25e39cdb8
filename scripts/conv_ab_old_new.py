"""Same-box A/B of the conv kernels: the current library vs scripts/libconv_old.so (conv_igemm.hip of an older commit,
built by hand: `git show <commit>:cvcs_amd/csrc/{conv_igemm.hip,common.h,api.hip}` + the matching include/cvcs_hip.h into a
scratch directory, `hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared conv_igemm.hip api.hip -o scripts/libconv_old.so`).
Box-to-box spread is +-3 %, so kernel changes of a few per cent can only be judged inside one process.  Data-gradient-like launches (no bias, no statistics) of the five U-Net levels, interleaved."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvcs_amd import ops, _lib

old = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libconv_old.so"))

class OldDesc(C.Structure):
    _fields_ = _lib.ConvDesc._fields_ if os.environ.get("OLD_SAME_ABI") == "1" else _lib.ConvDesc._fields_[:24]   # [:24]: the descriptor before the eval-fold / Cin_valid fields

old.cvcs_conv2d.restype = C.c_int
old.cvcs_conv2d.argtypes = [C.POINTER(OldDesc), C.c_void_p]
dev = "cuda:0"

def desc(cls, x, w, out, B, S, Cin, Cout):
    d = cls()
    d.in_, d.in_ld, d.B, d.H, d.W, d.Cin = x.data_ptr(), Cin, B, S, S, Cin
    d.wt, d.bias, d.out, d.out_ld, d.Ho, d.Wo, d.Cout = w.data_ptr(), None, out.data_ptr(), Cout, S, S, Cout
    d.KH, d.KW, d.stride, d.pad, d.dil, d.relu, d.pixel_shuffle, d.dtype = 3, 3, 1, 1, 1, 0, 0, 1
    return d

def timeit(fn, reps=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps

new = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
for (B, S, Cin, Cout) in ((32, 512, 64, 64), (32, 256, 128, 128), (32, 128, 256, 256), (32, 64, 512, 512), (32, 32, 1024, 1024)):
    x = torch.randn(B, S, S, Cin, device=dev).clamp_(min=0).to(torch.bfloat16)
    w = (torch.randn(9, Cout, Cin, device=dev) / (9 * Cin) ** 0.5).to(torch.bfloat16)
    out = torch.empty(B, S, S, Cout, dtype=torch.bfloat16, device=dev)
    dn, do = desc(_lib.ConvDesc, x, w, out, B, S, Cin, Cout), desc(OldDesc, x, w, out, B, S, Cin, Cout)
    res = []
    for rnd in range(3):
        res.append((timeit(lambda: old.cvcs_conv2d(C.byref(do), st)), timeit(lambda: new.cvcs_conv2d(C.byref(dn), st))))
    print(f"S={S} {Cin}->{Cout}: old/new us " + "  ".join(f"{a:.1f}/{b:.1f}" for a, b in res), flush=True)

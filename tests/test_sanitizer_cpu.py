"""Host side of the C-ABI under AddressSanitizer + UBSan (SURVEY section 5 "race detection / sanitizers"; CPU container only):
`make -C cvcs_amd/csrc asan` builds libcvcs_hip_asan.so with the host code instrumented, and a child Python process (ASan runtime
preloaded) drives every host-only path: descriptor validation of each entry point with hostile arguments, the launch / split-K
planners over a sweep of layer shapes, the sizing queries.  Any heap / stack error or undefined behaviour aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_RT = "/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so"

CHILD = r'''
import ctypes as C, itertools, sys
sys.path.insert(0, %(root)r)
from cvcs_amd import _lib
h = C.CDLL(%(lib)r)
for name, (res, args) in _lib.SIGNATURES.items():
    fn = getattr(h, name); fn.restype = res; fn.argtypes = args
assert h.cvcs_abi_version() == _lib.ABI_VERSION
assert h.cvcs_sizeof_conv_desc() == C.sizeof(_lib.ConvDesc) and h.cvcs_sizeof_wgrad_desc() == C.sizeof(_lib.WgradDesc)
n = 0
# planners / sizing queries over a sweep of shapes (pure host code)
for B, S, cin, cout, k, stride in itertools.product((1, 2, 32), (8, 17, 64, 512), (16, 64, 192, 3072), (64, 128, 2048), (1, 2, 3), (1, 2)):
    r = h.cvcs_wgrad_slices(B, S, S, cout, cin, k, k, stride); n += 1
    assert r >= 1, (B, S, cin, cout, k, stride, r)
    d = _lib.WgradDesc()
    d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad, d.dtype = B, S, S, cin, cout, k, k, stride, k // 2, 1
    d.Ho = (S + 2 * (k // 2) - k) // stride + 1; d.Wo = d.Ho
    assert h.cvcs_wgrad_workspace_floats(C.byref(d)) > 0
    for dil in (2, 6, 18):
        d.dil = dil
        assert h.cvcs_wgrad_workspace_floats(C.byref(d)) > 0
    c = _lib.ConvDesc()
    c.B, c.H, c.W, c.Cin, c.Cout, c.KH, c.KW, c.stride, c.pad, c.dil, c.dtype = B, S, S, cin, cout, k, k, stride, k // 2, 1, 1
    c.Ho = d.Ho; c.Wo = d.Ho
    assert h.cvcs_conv_stat_rows(C.byref(c)) > 0
for M in (1, 63, 64, 65, 1 << 20, 1 << 31):
    assert h.cvcs_bn_bwd_rows(M) >= 1 and h.cvcs_head_bwd_rows(M) >= 1 and h.cvcs_linear_head_bwd_rows(M) >= 1 and h.cvcs_ce_workspace_floats(M) > 2
    assert h.cvcs_bn_finalize_workspace_floats(min(M, 1 << 20), 64) >= 0
# every launch entry point with a zeroed / hostile argument list must refuse on the host (no HIP call is reached)
bad = 0
for name, (res, args) in _lib.SIGNATURES.items():
    if name in _lib._QUERIES:
        continue
    vals = []
    for a in args:
        if a is C.c_void_p: vals.append(None)
        elif a in (C.c_int, C.c_int64): vals.append(0)
        elif a is C.c_float: vals.append(0.0)
        else: vals.append(a())          # POINTER(desc): NULL
    rc = getattr(h, name)(*vals)
    assert rc < 0, (name, rc)
    assert len(h.cvcs_last_error()) > 0
    bad += 1
# descriptors with inconsistent geometry / misaligned views / bad channel counts
for mut in (dict(Cin=33), dict(Cout=80), dict(Ho=7), dict(in_ld=3), dict(dtype=9), dict(stride=0), dict(Cin_valid=5)):
    c = _lib.ConvDesc()
    c.in_, c.wt, c.out = 4096, 4096, 4096
    c.B, c.H, c.W, c.Cin, c.Cout, c.KH, c.KW, c.stride, c.pad, c.dil, c.dtype, c.Ho, c.Wo, c.in_ld, c.out_ld = 1, 8, 8, 64, 64, 3, 3, 1, 1, 1, 1, 8, 8, 64, 64
    for k_, v in mut.items(): setattr(c, k_, v)
    assert h.cvcs_conv2d(C.byref(c), None) < 0, mut
print("asan child ok:", n, "shapes,", bad, "entry points refused hostile arguments")
'''


@pytest.mark.skipif(not os.path.exists(ASAN_RT), reason="no ASan runtime in this image")
def test_host_side_under_asan_ubsan(tmp_path):
    lib = os.path.join(ROOT, "cvcs_amd", "libcvcs_hip_asan.so")
    srcs = [os.path.join(ROOT, "cvcs_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "cvcs_amd", "csrc")) if f.endswith((".hip", ".h"))]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(s) for s in srcs):
        subprocess.run(["make", "-C", os.path.join(ROOT, "cvcs_amd", "csrc"), "asan"], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=ASAN_RT, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, lib=lib)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "asan child ok" in r.stdout, (r.stdout[-800:], r.stderr[-3000:])

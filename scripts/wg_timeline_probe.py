"""Probe (GPU): where a workgroup of the 3x3 halo kernel spends its life.  Needs scripts/libconv_probe.so:
  cd cvcs_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCVCS_PROBE conv_igemm.hip api.hip -o ../../scripts/libconv_probe.so
The probe build stamps s_memtime at workgroup entry, after the prologue's DMA issue, after its wait, after the main loop and at exit."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvcs_amd import _lib
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("PROBE_LIB", "libconv_probe.so")))
lib.cvcs_conv2d.restype = C.c_int
lib.cvcs_conv2d.argtypes = [C.POINTER(_lib.ConvDesc), C.c_void_p]
lib.cvcs_probe_read.argtypes = [C.c_void_p, C.c_size_t]
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
SHAPES = ((32, 128, 64, 64, 1), (32, 64, 128, 128, 1), (32, 32, 256, 256, 1), (32, 16, 512, 512, 1)) if os.environ.get("PROBE_SHAPES") == "resnet" else ((32, 512, 64, 64, 0), (32, 512, 64, 64, 1), (32, 512, 128, 64, 1), (32, 256, 128, 128, 0), (32, 64, 512, 512, 0))
for (B, S, Cin, Cout, stats) in SHAPES:
    x = torch.randn(B, S, S, Cin, device=dev).clamp_(min=0).to(torch.bfloat16)
    w = (torch.randn(9, Cout, Cin, device=dev) / (9 * Cin) ** 0.5).to(torch.bfloat16)
    out = torch.empty(B, S, S, Cout, dtype=torch.bfloat16, device=dev)
    bias = torch.zeros(Cout, device=dev)
    rows = B * (S // 16) ** 2
    ss, sq, sc = torch.empty(rows * Cout, device=dev), torch.empty(rows * Cout, device=dev), torch.empty(rows, device=dev)
    d = _lib.ConvDesc()
    d.in_, d.in_ld, d.B, d.H, d.W, d.Cin = x.data_ptr(), Cin, B, S, S, Cin
    d.wt, d.bias, d.out, d.out_ld, d.Ho, d.Wo, d.Cout = w.data_ptr(), (bias.data_ptr() if stats else None), out.data_ptr(), Cout, S, S, Cout
    d.KH, d.KW, d.stride, d.pad, d.dil, d.relu, d.pixel_shuffle, d.dtype = 3, 3, 1, 1, 1, 0, 0, 1
    if stats: d.stat_sum, d.stat_m2, d.stat_cnt = ss.data_ptr(), sq.data_ptr(), sc.data_ptr()
    for _ in range(3): assert lib.cvcs_conv2d(C.byref(d), st) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); assert lib.cvcs_conv2d(C.byref(d), st) == 0; e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    n = min(rows, 32768)
    buf = np.zeros(8 * 32768, dtype=np.uint64)
    assert lib.cvcs_probe_read(buf.ctypes.data, buf.nbytes) == 0
    t = buf.reshape(-1, 8)[:n].astype(np.int64)
    # columns 0-4: s_memtime (a per-XCD counter: only differences inside one workgroup mean anything); 5, 6: s_memrealtime
    # (100 MHz, chip-wide) at entry and exit -> the tick of s_memtime and the global timeline
    real_us = (t[:, 6] - t[:, 5]) * 0.01
    tick_us = real_us.sum() / (t[:, 4] - t[:, 0]).sum()
    span_us = (t[:, 6].max() - t[:, 5].min()) * 0.01 * (rows / n if rows > n else 1.0)
    ph = np.diff(t[:, :5], axis=1) * tick_us
    life = (t[:, 4] - t[:, 0]) * tick_us
    mid = slice(n // 4, 3 * n // 4)   # steady state
    conc = life.sum() / span_us
    us = span_us
    h0, h1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    h0.record()
    for _ in range(10): lib.cvcs_conv2d(C.byref(d), st)
    h1.record(); torch.cuda.synchronize()
    hot_us = h0.elapsed_time(h1) * 100
    print(f"[hot: {hot_us:.0f} us/launch over 10] ", end="")
    print(f"S={S} {Cin}->{Cout} stats={stats}: kernel {us:.0f} us, tick {tick_us*1e3:.2f} ns, WG life {life[mid].mean():.2f} us "
          f"(setup {((t[mid,7]-t[mid,0])*tick_us).mean():.2f}, DMA issue {((t[mid,1]-t[mid,7])*tick_us).mean():.2f}, wait for the first slice {ph[mid,1].mean():.2f}, main loop {ph[mid,2].mean():.2f}, epilogue {ph[mid,3].mean():.2f}); resident WGs {conc:.0f} "
          f"of grid {rows}x{Cout // (128 if Cout % 128 == 0 else 64)}", flush=True)

"""window-attention launches alone at the Swin-T stage shapes of the benchmarked step (batch 32, 512 x 512 tiles): ms per launch and the
HBM rate of the algorithmic bytes (forward: read qkv, write out; backward: read qkv + dout, write dqkv)"""
import sys
import torch
sys.path.insert(0, ".")
from cvcs_amd import ops  # noqa: E402

dev = "cuda:0"
B = 32
dtype = torch.bfloat16 if len(sys.argv) < 2 else getattr(torch, sys.argv[1])
es = 2 if dtype == torch.bfloat16 else 4
for H, heads, shift in ((128, 3, 3), (64, 6, 3), (32, 12, 0), (16, 24, 3)):
    C_ = heads * 32
    Hp = -(-H // 7) * 7
    T = B * Hp * Hp
    qkv = ops.view((torch.randn(1, T, 1, 3 * C_, device=dev) * 0.5).to(dtype))
    go = ops.view(torch.randn(1, T, 1, C_, device=dev).to(dtype))
    out = ops.view(torch.empty(1, T, 1, C_, dtype=dtype, device=dev))
    dqkv = ops.view(torch.empty(1, T, 1, 3 * C_, dtype=dtype, device=dev))
    table = torch.randn(169, heads, device=dev) * 0.1
    dtable = torch.empty(169, heads, device=dev)
    ws = torch.empty(ops.window_attention_bwd_workspace(B, H, H, heads), device=dev)
    res = []
    for name, fn, nbytes in (("fwd", lambda: ops.window_attention_fwd(qkv, B, H, H, heads, shift, table, out), T * 4 * C_ * es),
                             ("bwd", lambda: ops.window_attention_bwd(qkv, go, B, H, H, heads, shift, table, dqkv, dtable, ws), T * 8 * C_ * es)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        res.append(f"{name} {ms:8.3f} ms  {nbytes / ms / 1e6:7.1f} GB/s")
    print(f"tokens {H}x{H} (padded {Hp}), heads {heads}, shift {shift}: " + " | ".join(res), flush=True)

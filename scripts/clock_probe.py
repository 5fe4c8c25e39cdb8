"""Sample rocm-smi (sclk, power) while one kernel family runs back-to-back for a few seconds."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvcs_amd import ops
dev = "cuda:0"
which, B, S, Cin, Cout = sys.argv[1], *(int(v) for v in sys.argv[2:6])
data = os.environ.get("DATA", "randn")
x = torch.randn(B, S, S, Cin, device=dev).to(torch.bfloat16)
dy = torch.randn(B, S, S, Cout, device=dev).to(torch.bfloat16)
if data == "zeros":
    x.zero_(); dy.zero_()
w = (torch.randn(9, Cout, Cin, device=dev) / (9 * Cin) ** 0.5).to(torch.bfloat16)
out = torch.empty(B, S, S, Cout, dtype=torch.bfloat16, device=dev)
dw = torch.empty(Cout, Cin, 3, 3, device=dev)
ws = torch.empty(ops.wgrad_workspace_floats(B, S, S, Cout, Cin, 3, 3, 1), device=dev)
def run():
    if which == "conv":
        ops.conv2d(ops.view(x), w, None, ops.view(out), 3, 3, 1, 1)
    elif which == "copy":
        out.copy_(dy)
    else:
        ops.conv2d_wgrad(ops.view(x), ops.view(dy), dw, 3, 3, 1, 1, ws)
samples = []
stop = False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            samples.append(" | ".join(l.strip() for l in o.splitlines() if ("sclk" in l or "Power" in l or "mclk" in l or "fclk" in l)))
        except Exception as e:
            samples.append(repr(e))
        time.sleep(0.3)
for _ in range(3): run()
torch.cuda.synchronize()
t = threading.Thread(target=sampler); t.start()
t0 = time.time(); n = 0
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
while time.time() - t0 < 4.0:
    for _ in range(50): run()
    n += 50
    torch.cuda.synchronize()
e.record(); torch.cuda.synchronize()
stop = True; t.join()
us = s.elapsed_time(e) * 1e3 / n
print(f"{which} {data} B={B} S={S} {Cin}->{Cout}: {us:.1f} us  {2.0*B*S*S*Cin*Cout*9/us/1e6:.1f} TFLOP/s")
for smp in samples[2:8]: print("   ", smp)

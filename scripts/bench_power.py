"""Run bench.py as a child and sample rocm-smi (sclk, socket power) while it trains."""
import subprocess, sys, time
p = subprocess.Popen([sys.executable, "bench.py", "--steps", "60", "--warmup", "3", "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
samples = []
while p.poll() is None:
    try:
        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        sclk = [l.split("(")[-1].split(")")[0] for l in o.splitlines() if "sclk" in l]
        pw = [l.split(":")[-1].strip() for l in o.splitlines() if "Power (W)" in l]
        samples.append((round(time.time() % 1000, 1), sclk[0] if sclk else "?", pw[0] if pw else "?"))
    except Exception as e:
        samples.append(repr(e))
    time.sleep(0.25)
out = p.stdout.read()
print([l for l in out.splitlines() if l.startswith("{")][-1][:200])
for s in samples: print(s)

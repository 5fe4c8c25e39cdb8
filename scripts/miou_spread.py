"""How far does the bf16 path's held-out mIoU land from the f32 CPU oracle's for NEARBY schedules of the same task?  (the yardstick for
tests/test_model_gpu.py::test_heldout_miou_after_training_matches_cpu_reference: a chaotic trajectory's end point)
usage: python scripts/miou_spread.py [model] [precision] [S] steps..."""
import importlib.util
import os
import sys

spec = importlib.util.spec_from_file_location("miou_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "miou_parity.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
model, precision, S = sys.argv[1], sys.argv[2], int(sys.argv[3])
args = sys.argv[4:]
seeds = [0]
epoch_steps = None
if "epoch_steps" in args:    # ... epoch_steps 10: the reference's per-epoch PolynomialLR with one "epoch" every 10 steps
    i = args.index("epoch_steps")
    epoch_steps = int(args[i + 1])
    args = args[:i] + args[i + 2:]
if "seeds" in args:          # ... steps ... seeds 0 1 2 3 4: the ensemble over initial parameters / training tiles
    i = args.index("seeds")
    args, seeds = args[:i], [int(a) for a in args[i + 1:]]
import time
for steps in map(int, args):
    d = []
    for seed in seeds:
        t0 = time.time()
        m_o, m_h = mod.run(precision, steps=steps, S=S, verbose=False, model=model, seed=seed, epoch_steps=epoch_steps)
        d.append(100 * (m_h["mIoU"] - m_o["mIoU"]))
        print(f"{model} {precision} {steps} steps seed {seed}: oracle mIoU {100 * m_o['mIoU']:.3f}  HIP {100 * m_h['mIoU']:.3f}  difference {d[-1]:+.3f} points  ({time.time() - t0:.0f}s)", flush=True)
    if len(d) > 1:
        mean = sum(d) / len(d)
        sd = (sum((x - mean) ** 2 for x in d) / (len(d) - 1)) ** 0.5
        print(f"{model} {precision} {steps} steps: mean difference {mean:+.3f} points over {len(d)} seeds, standard deviation {sd:.3f}", flush=True)

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
for steps in map(int, sys.argv[4:]):
    m_o, m_h = mod.run(precision, steps=steps, S=S, verbose=False, model=model)
    print(f"{model} {precision} {steps} steps: oracle mIoU {100 * m_o['mIoU']:.3f}  HIP {100 * m_h['mIoU']:.3f}  difference {100 * (m_h['mIoU'] - m_o['mIoU']):+.3f} points", flush=True)

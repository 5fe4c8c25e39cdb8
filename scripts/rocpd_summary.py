"""per-kernel summary of a rocprofv3 sqlite result (rocpd): python scripts/rocpd_summary.py results.db [steps] [group-by-grid]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    by_grid = len(sys.argv) > 3
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name = "display_name" if "display_name" in cols else "kernel_name"
    q = f"select s.{name}, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id"
    agg = {}
    for n, gx, gy, gz, dt in cur.execute(q):
        key = (n[:100], gx, gy, gz) if by_grid else n[:110]
        a = agg.setdefault(key, [0, 0])
        a[0] += 1
        a[1] += dt
    tot = sum(a[1] for a in agg.values())
    print(f"total kernel time {tot / 1e6:.2f} ms = {tot / 1e6 / steps:.3f} ms/step over {steps} steps")
    for key, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"{t / 1e6 / steps:8.3f} ms/step {100 * t / tot:5.1f}%  n/step={n / steps:7.1f} avg={t / n / 1e3:8.1f}us  {key}")


if __name__ == "__main__":
    main()

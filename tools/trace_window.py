"""Aggregate the LAST part of a rocprofv3 kernel trace (CSV): per kernel calls / total / average, and how busy the GPU was.

    python tools/trace_window.py <dir with *_kernel_trace.csv> [fraction of the time range to keep = 0.25] [steps in it]

For time-stepping cases whose early steps are not representative (more Krylov iterations while the flow starts up).
"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    t0, t1 = rows[0][0], rows[-1][1]
    cut = t1 - int(frac * (t1 - t0))
    sel = [r for r in rows if r[0] >= cut]
    busy = sum(e - s for s, e, _ in sel)
    win = sel[-1][1] - sel[0][0]
    agg = {}
    for s, e, n in sel:
        a = agg.setdefault(n, [0, 0])
        a[0] += 1
        a[1] += e - s
    print(f"# window {win / 1e6:.2f} ms, {len(sel)} launches, kernels busy {busy / 1e6:.2f} ms = {busy / win:.2f} of it"
          + (f"; per step: {win / 1e3 / steps:.1f} us wall, {len(sel) / steps:.1f} launches, {busy / 1e3 / steps:.1f} us of kernels" if steps else ""))
    # idle time by the pair of kernels around it (gaps of more than 10 us)
    gaps = {}
    last_end = sel[0][1]
    last_name = sel[0][2]
    for s0, e0, n0 in sel[1:]:
        g = s0 - last_end
        if g > 10000:
            a = gaps.setdefault((last_name[:60], n0[:60]), [0, 0])
            a[0] += 1
            a[1] += g
        if e0 > last_end:
            last_end, last_name = e0, n0
    tot_gap = sum(v[1] for v in gaps.values())
    print(f"# gaps > 10 us: {tot_gap / 1e6:.2f} ms in all" + (f" = {tot_gap / 1e3 / steps:.1f} us per step" if steps else ""))
    for (a, b2), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"   {t / 1e6:8.2f} ms  {c:6d} x {t / c / 1e3:8.1f} us   after {a}  before {b2}")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        per = f" {c / steps:6.1f}/step" if steps else ""
        print(f"{t / busy * 100:5.1f}%  {c:7d}{per}  avg {t / c / 1e3:7.2f} us  {n[:110]}")


if __name__ == "__main__":
    main()

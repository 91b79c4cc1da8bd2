"""Timeline of ONE iteration out of a rocprofv3 kernel trace (CSV): the dispatches between the last two launches of a marker
kernel (default: k_spmv_lds), each with its start offset, duration and the idle gap before it.

    python tools/trace_sequence.py <dir with *_kernel_trace.csv> [marker substring] [how many markers from the end = 2]
"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "k_spmv_lds"
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    a, b = marks[-back - 1], marks[-back]
    t0 = rows[a][0]
    busy = 0
    last_end = rows[a][0]
    print(f"# {b - a} dispatches between two launches of {marker}: {1e-3 * (rows[b][0] - t0):.1f} us")
    print("| start us | dur us | gap us | kernel |")
    print("|---|---|---|---|")
    for s, e, n in rows[a:b]:
        name = n.replace("void pib::", "").replace("pib::", "")
        name = name[: name.index("(")] if "(" in name else name
        print(f"| {1e-3 * (s - t0):8.1f} | {1e-3 * (e - s):7.1f} | {1e-3 * (s - last_end):6.1f} | {name} |")
        busy += e - s
        last_end = e
    print(f"# kernels busy {1e-3 * busy:.1f} us of {1e-3 * (rows[b][0] - t0):.1f}")


if __name__ == "__main__":
    main()

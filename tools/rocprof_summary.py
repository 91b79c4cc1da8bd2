#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace [--stats] [--pmc ...] CSV output directory into one small
per-kernel table (calls, total/avg/min/max duration, share, optional PMC means) for profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_dir [--out profiles/r1_xxx.md] [--title ...]
"""
import argparse
import collections
import csv
import glob
import os
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--out", default=None)
    ap.add_argument("--title", default="")
    ap.add_argument("--cmd", default="")
    a = ap.parse_args()
    traces = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    if not traces:
        sys.exit(f"no *kernel_trace.csv under {a.dir}")
    agg = collections.OrderedDict()
    for f in traces:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3  # us
            e = agg.setdefault(name, {"n": 0, "tot": 0.0, "min": 1e30, "max": 0.0, "vgpr": r.get("VGPR_Count", ""),
                                      "lds": r.get("LDS_Block_Size", ""), "grid": r.get("Grid_Size", ""),
                                      "wg": r.get("Workgroup_Size", "")})
            e["n"] += 1
            e["tot"] += d
            e["min"] = min(e["min"], d)
            e["max"] = max(e["max"], d)
    # counters: the mean over all dispatches of a kernel and -- the multigrid launches one kernel on every level -- the
    # values of its LONGEST dispatch (the fine level), matched through the dispatch id
    dur = {}
    for f in traces:
        for r in csv.DictReader(open(f)):
            dur[r.get("Dispatch_Id", "")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    top = {}
    for f in glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            pmc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            d = dur.get(r.get("Dispatch_Id", ""), 0.0)
            t = top.setdefault(r["Kernel_Name"], {"us": -1.0, "id": None, "c": {}})
            if d > t["us"]:
                t.update(us=d, id=r.get("Dispatch_Id"), c={})
            if r.get("Dispatch_Id") == t["id"]:
                t["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    counters = sorted({c for k in pmc.values() for c in k})
    total = sum(e["tot"] for e in agg.values())
    lines = []
    if a.title:
        lines.append(f"# {a.title}\n")
    if a.cmd:
        lines.append(f"command: `{a.cmd}`\n")
    lines.append(f"total kernel time {total / 1e3:.2f} ms over {sum(e['n'] for e in agg.values())} dispatches\n")
    hdr = ["kernel", "calls", "total_ms", "share_%", "avg_us", "min_us", "max_us", "vgpr", "lds_B", "wg"] + \
          [f"mean_{c}" for c in counters] + ([f"longest_{c}" for c in counters] if counters else [])
    lines.append("| " + " | ".join(hdr) + " |")
    lines.append("|" + "---|" * len(hdr))
    for name, e in sorted(agg.items(), key=lambda kv: -kv[1]["tot"]):
        short = name if len(name) < 100 else name[:97] + "..."
        row = [f"`{short}`", str(e["n"]), f"{e['tot'] / 1e3:.3f}", f"{100 * e['tot'] / total:.1f}",
               f"{e['tot'] / e['n']:.1f}", f"{e['min']:.1f}", f"{e['max']:.1f}", e["vgpr"], e["lds"], e["wg"]]
        for c in counters:
            v = pmc.get(name, {}).get(c)
            row.append(f"{sum(v) / len(v):.4g}" if v else "")
        for c in counters:
            v = top.get(name, {}).get("c", {}).get(c)
            row.append(f"{v:.4g}" if v is not None else "")
        lines.append("| " + " | ".join(row) + " |")
    lines.append("\n`vgpr` is rocprofv3's VGPR_Count as printed; on gfx950 it is HALF the code object's `.vgpr_count` (72 here is 144 "
                 "32-bit registers per lane: three waves per SIMD, not seven).")
    text = "\n".join(lines) + "\n"
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        open(a.out, "w").write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()

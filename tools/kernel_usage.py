"""Register / LDS / occupancy table of every kernel of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage; runs in
the build container, no GPU).  python tools/kernel_usage.py petibm_amd/csrc/gmg.hip [-DFOO ...] [--grep k_prolong]"""
import os
import re
import subprocess
import sys
import tempfile


def usage(src, extra=()):
    with tempfile.TemporaryDirectory() as d:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-x", "hip", "-c", src,
               "-o", os.path.join(d, "x.o"), "-Rpass-analysis=kernel-resource-usage", *extra]
        t = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True, cwd=d).stderr
    rows = []
    for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
        name = b.split("\n")[0].strip()

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        rows.append((name, g("VGPRs"), g("AGPRs"), g("VGPR Spill"), g("ScratchSize \[bytes/lane\]"), g("LDS Size \[bytes/block\]"),
                     g("Occupancy \[waves/SIMD\]")))
    return rows


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True)
    return p.stdout.split("\n")


if __name__ == "__main__":
    args = sys.argv[1:]
    pat = None
    if "--grep" in args:
        i = args.index("--grep")
        pat = args[i + 1]
        del args[i:i + 2]
    rows = usage(os.path.abspath(args[0]), args[1:])
    names = demangle([r[0] for r in rows])
    print(f"{'kernel':80s} {'vgpr':>5} {'agpr':>5} {'spill':>5} {'scratch':>7} {'lds':>7} {'waves/SIMD':>10}")
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        if pat and pat not in n:
            continue
        print(f"{n[:80]:80s} {r[1]:5d} {r[2]:5d} {r[3]:5d} {r[4]:7d} {r[5]:7d} {r[6]:10d}")

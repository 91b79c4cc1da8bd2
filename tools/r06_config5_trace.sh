#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r06q; mkdir -p $O
P=$O/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o t -- python tools/config5_heaving_plate.py --steps 15 > $O/prof.log 2>&1
python tools/rocprof_summary.py $P --out $O/config5_kernel_trace.md --title "tools/config5_heaving_plate.py --steps 15, rocprofv3 --kernel-trace --stats (round 6 closing source)" || true
rm -rf $P
head -45 $O/config5_kernel_trace.md | cut -c1-170

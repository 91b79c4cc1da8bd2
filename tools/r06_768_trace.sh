#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r06m; mkdir -p $O
P=$O/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o t -- python bench.py --grid 768 --steps 2 --warmup 1 --no-cpu --no-secondary --pmc off > $O/prof.log 2>&1
python tools/rocprof_summary.py $P --out $O/kernel_trace_768.md --title "bench.py --grid 768 --steps 2 --warmup 1, rocprofv3 --kernel-trace --stats" || true
rm -rf $P
head -22 $O/kernel_trace_768.md | cut -c1-190

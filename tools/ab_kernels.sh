#!/bin/bash
# tools/ab_kernels.sh "PATTERN" VARIANT...: per-kernel times (rocprofv3 kernel trace of the 512^3 bench) of the kernels matching
# PATTERN with petibm_amd/lib/var_<V>.so in place of the library, in turn; the solve time of each run
export TMPDIR=/tmp
PAT=$1; shift
cp petibm_amd/lib/libpetibm_amd.so /tmp/keep.so
for v in "$@"; do
  cp petibm_amd/lib/var_$v.so petibm_amd/lib/libpetibm_amd.so
  P=/tmp/abk_$v; rm -rf $P
  rocprofv3 --kernel-trace --stats --output-format csv -d $P -o t -- python bench.py --steps 3 --warmup 1 --no-cpu --no-secondary --pmc off --kernel-reps 2 > /tmp/abk_$v.log 2>&1
  python tools/rocprof_summary.py $P --out /tmp/abk_$v.md --title "$v" > /dev/null
  echo "== $v: $(grep -h 'ms_per_step' /tmp/abk_$v.log | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms per solve/')"
  grep "$PAT" /tmp/abk_$v.md | awk -F'|' '{printf "   %-50s calls %s avg %s max %s\n", substr($2,1,50), $3, $6, $8}'
done
cp /tmp/keep.so petibm_amd/lib/libpetibm_amd.so

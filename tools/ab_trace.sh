#!/bin/bash
# tools/ab_trace.sh VARIANT...: per-kernel times of the 512^3 bench with petibm_amd/lib/var_<V>.so in place of the library
export TMPDIR=/tmp
cp petibm_amd/lib/libpetibm_amd.so /tmp/keep.so
for v in "$@"; do
  cp petibm_amd/lib/var_$v.so petibm_amd/lib/libpetibm_amd.so
  P=/tmp/abtrace_$v; rm -rf $P
  rocprofv3 --kernel-trace --stats --output-format csv -d $P -o t -- python bench.py --steps 3 --warmup 1 --no-cpu --no-secondary --kernel-reps 2 > /tmp/ab_$v.log 2>&1
  python tools/rocprof_summary.py $P --out /tmp/ab_$v.md --title "$v" > /dev/null
  echo "== $v: $(grep 'total kernel time' /tmp/ab_$v.md)"
  grep "k_level_march\|k_prolong_smooth\|k_presmooth2\|k_restrict_march" /tmp/ab_$v.md | awk -F'|' '{printf "   %-60s calls %s avg %s max %s\n", substr($2,1,60), $3, $6, $8}'
done
cp /tmp/keep.so petibm_amd/lib/libpetibm_amd.so

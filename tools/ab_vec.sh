#!/bin/bash
# tools/ab_vec.sh VARIANT...: as ab_trace.sh, printing the Krylov vector kernels and the SpMV
export TMPDIR=/tmp
cp petibm_amd/lib/libpetibm_amd.so /tmp/keep.so
for v in "$@"; do
  cp petibm_amd/lib/var_$v.so petibm_amd/lib/libpetibm_amd.so
  P=/tmp/abtrace_$v; rm -rf $P
  rocprofv3 --kernel-trace --stats --output-format csv -d $P -o t -- python bench.py --steps 3 --warmup 1 --no-cpu --no-secondary --kernel-reps 2 > /tmp/ab_$v.log 2>&1
  python tools/rocprof_summary.py $P --out /tmp/ab_$v.md --title "$v" > /dev/null
  echo "== $v: $(grep 'total kernel time' /tmp/ab_$v.md) $(python -c "import json,sys; print(json.loads(open('/tmp/ab_$v.log').read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null)"
  grep "k_vec\|k_spmv_lds" /tmp/ab_$v.md | awk -F'|' '{printf "   %-70s calls %s avg %s max %s\n", substr($2,1,70), $3, $6, $8}'
done
cp /tmp/keep.so petibm_amd/lib/libpetibm_amd.so

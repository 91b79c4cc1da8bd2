#!/bin/bash
# tools/ab_literal.sh VARIANT...: the V(1,1)-literal 512^3 solve (pib_sweep_pairs=0, presweeps = postsweeps = 1: k_presmooth2<1>, k_restrict_march,
# k_prolong_smooth<1>) and the 256^3 V(2,2) solve with petibm_amd/lib/var_<V>.so in place of the library, in turn
export TMPDIR=/tmp
cp petibm_amd/lib/libpetibm_amd.so /tmp/keep.so
for v in "$@"; do
  cp petibm_amd/lib/var_$v.so petibm_amd/lib/libpetibm_amd.so
  a=$(python bench.py --steps 3 --warmup 1 --no-cpu --no-secondary --pmc off --kernel-reps 2 --presweeps 1 --postsweeps 1 2>/dev/null | sed 's/.*"ms_per_step": \([0-9.]*\).*"iters_per_solve": \([0-9.]*\).*/\1 ms \2 it/')
  b=$(python bench.py --grid 256 --steps 5 --warmup 1 --no-cpu --no-secondary --pmc off --kernel-reps 2 2>/dev/null | sed 's/.*"ms_per_step": \([0-9.]*\).*"iters_per_solve": \([0-9.]*\).*/\1 ms \2 it/')
  echo "== $v: 512^3 V(1,1): $a | 256^3 V(2,2): $b"
done
cp /tmp/keep.so petibm_amd/lib/libpetibm_amd.so

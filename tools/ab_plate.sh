#!/bin/bash
# tools/ab_plate.sh VARIANT...: the config-5-size heaving plate step (Chebyshev velocity file) and the V(1,1)-literal 512^3 solve with
# petibm_amd/lib/var_<V>.so in place of the library, in turn
export TMPDIR=/tmp
cp petibm_amd/lib/libpetibm_amd.so /tmp/keep.so
for v in "$@"; do
  cp petibm_amd/lib/var_$v.so petibm_amd/lib/libpetibm_amd.so
  a=$(python tools/config5_heaving_plate.py --chebyshev 2>/dev/null | tail -2 | tr '\n' ' ' | sed 's/.*steps: \([0-9.]*\) ms per step.*solvePoisson \([0-9.]*\).*/\1 ms per step, solvePoisson \2/')
  b=$(python bench.py --steps 3 --warmup 1 --no-cpu --no-secondary --pmc off --kernel-reps 2 --presweeps 1 --postsweeps 1 2>/dev/null | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/')
  echo "== $v: plate $a | 512^3 V(1,1) $b"
done
cp /tmp/keep.so petibm_amd/lib/libpetibm_amd.so

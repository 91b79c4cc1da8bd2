#!/bin/bash
# tools/ab_velocity.sh VARIANT...: the 256^3 velocity solves (BiCGStab + Jacobi, Chebyshev + Jacobi) and the product's own time with
# petibm_amd/lib/var_<V>.so in place of the library, in turn
export TMPDIR=/tmp
cp petibm_amd/lib/libpetibm_amd.so /tmp/keep.so
for v in "$@"; do
  cp petibm_amd/lib/var_$v.so petibm_amd/lib/libpetibm_amd.so
  a=$(python bench.py --system velocity --grid 256 --steps 3 --warmup 1 --kernel-reps 10 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms %g it, product %.1f us' % (d['ms_per_step'], d['iters_per_solve'], 1e3*d['roofline']['ms_per_launch']))")
  b=$(python bench.py --system velocity --grid 256 --steps 3 --warmup 1 --kernel-reps 10 --no-cpu --velocity-solver CHEBYSHEV --velocity-tol 1e-8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms %g it' % (d['ms_per_step'], d['iters_per_solve']))")
  echo "== $v: BiCGStab $a | Chebyshev $b"
done
cp /tmp/keep.so petibm_amd/lib/libpetibm_amd.so

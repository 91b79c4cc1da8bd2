#!/bin/bash
# tools/vel_ab.sh KEY=VALUE...: the 256^3 velocity line of bench.py with each extra solver-config line (and without any)
for kv in "" "$@"; do
  python bench.py --system velocity --grid ${GRID:-256} --steps 5 --warmup 2 --extra-config "$kv" 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s solve %.2f ms, plain product %.1f us, its %.0f, |r| %.2e' % ('$kv' or '(default)', b['ms_per_step'], 1e3*b['roofline']['ms_per_launch'], b['iters_per_solve'], b['true_abs_residual']))"
done

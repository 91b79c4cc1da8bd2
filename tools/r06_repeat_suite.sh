#!/bin/bash
# the GPU suite N times in a row on one box (flakiness count before the round closes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r06rep; mkdir -p $O
for k in 1 2 3; do
  timeout 1500 python -m pytest tests -m gpu -q -x > $O/run$k.log 2>&1; echo "run $k rc=$? $(grep -E 'passed|failed' $O/run$k.log | tail -1)"
done | tee $O/summary.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $O/summary.txt

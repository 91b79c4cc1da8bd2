#!/bin/bash
# BASELINE configs 4 and 5 at their sizes on one GPU, round 6 closing source
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r06p; mkdir -p $O
(echo "== tools/config4_cylinder_re550.py"; timeout 600 python tools/config4_cylinder_re550.py 2>&1 | grep -v amdgpu.ids | tail -6
 echo "== tools/config5_heaving_plate.py"; timeout 900 python tools/config5_heaving_plate.py 2>&1 | grep -v amdgpu.ids | tail -5
 echo "== tools/config5_heaving_plate.py --chebyshev"; timeout 900 python tools/config5_heaving_plate.py --chebyshev 2>&1 | grep -v amdgpu.ids | tail -5) | tee $O/configs45.txt

#!/bin/bash
# round 6 closing set, one gpurun call: the GPU suite, the default bench, rocprofv3 kernel trace + separate PMC passes (tools/gpu_round.sh),
# then the 512^3 / 256^3 time-step stages and a kernel trace of the 512^3 step
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
TAG=${1:-r06h}
bash tools/gpu_round.sh $TAG
export TMPDIR=/tmp
O=gpurun_out/$TAG
(echo "== 256^3"; timeout 600 python tools/cavity3d_stages.py 256 | grep -v amdgpu.ids; echo "== 512^3"; timeout 600 python tools/cavity3d_stages.py 512 --steps 6 | grep -v amdgpu.ids; echo "== 512^3, Chebyshev velocity file"; timeout 600 python tools/cavity3d_stages.py 512 --steps 6 --chebyshev | grep -v amdgpu.ids) > $O/time_step_stages.txt 2>&1
tail -12 $O/time_step_stages.txt
P=$O/prof_step
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o t -- python tools/cavity3d_stages.py 512 --steps 3 > $O/prof_step.log 2>&1
python tools/rocprof_summary.py $P --out $O/time_step_kernel_trace.md --title "$TAG tools/cavity3d_stages.py 512 --steps 3 (5 + 3 steps), rocprofv3 --kernel-trace --stats" || true
rm -rf $P
head -20 $O/time_step_kernel_trace.md | cut -c1-200

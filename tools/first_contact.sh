#!/bin/bash
# First contact with a multi-GPU node (nothing with N > 1 has ever run on RCCL / xGMI: DESIGN.md 5).  Runs, in this order and
# each under its own timeout so that one hang does not lose the rest:
#   1. what a collective costs between two DIFFERENT GPUs: RCCL send/recv + all-reduce next to both peer-transport flavours
#   2. bench.py --gpus 2 on RCCL, on the peer transport, and as the driver runs it (--transport auto: both timed, the faster kept)   (256^3: seconds)
#   3. the forced-failure drill: RCCL's bootstrap "fails", bench.py must fall back to the peer transport and say so
#   4. bench.py --gpus N (default 8) on RCCL and on the peer transport at the headline size, with the per-rank counters
#   5. the scaling sweep 1 / 2 / 4 / N the driver makes (default flags: transport and CG recurrence tuned on untimed solves)
# usage: tools/first_contact.sh [N]      output: gpurun_out/first_contact/*.json|txt (copy what matters to profiles/)
N=${1:-8}
OUT=gpurun_out/first_contact
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$(dirname "$0")/.."
ndev=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
echo "GPUs visible: $ndev" | tee $OUT/summary.txt
if [ "$ndev" -lt 2 ]; then echo "needs >= 2 GPUs" | tee -a $OUT/summary.txt; exit 2; fi
run() { # name timeout cmd...
  local name=$1 t=$2; shift 2
  echo "== $name: $*" | tee -a $OUT/summary.txt
  timeout $t "$@" > $OUT/$name.out 2> $OUT/$name.err; local rc=$?
  echo "   rc=$rc  $(grep -h '^{' $OUT/$name.out | tail -1 | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','ms_per_step','iters_per_solve','true_rel_residual','notes')}, d.get('config',{}).get('transport'), d.get('config',{}).get('cg_recurrence'))
except Exception as e: print('')" 2>/dev/null)" | tee -a $OUT/summary.txt
}
run latency_two_gpus 600 python tools/comm_latency.py --two-gpus
cat $OUT/latency_two_gpus.out >> $OUT/summary.txt
run bench2_rccl_256 600 python bench.py --gpus 2 --grid 256 --steps 5 --warmup 2 --no-cpu --no-secondary --transport rccl
run bench2_peer_256 600 python bench.py --gpus 2 --grid 256 --steps 5 --warmup 2 --no-cpu --no-secondary --transport peer
run bench2_auto_256 600 python bench.py --gpus 2 --grid 256 --steps 5 --warmup 2 --no-cpu --no-secondary
PIB_FORCE_RCCL_FAIL=1 run bench2_forced_rccl_failure 600 python bench.py --gpus 2 --grid 256 --steps 2 --warmup 1 --no-cpu --no-secondary
if [ "$ndev" -ge "$N" ]; then
  run bench${N}_rccl_512 900 python bench.py --gpus $N --steps 10 --warmup 3 --no-cpu --no-secondary --transport rccl
  run bench${N}_peer_512 900 python bench.py --gpus $N --steps 10 --warmup 3 --no-cpu --no-secondary --transport peer
  run bench${N}_rccl_512_standard 900 python bench.py --gpus $N --steps 10 --warmup 3 --no-cpu --no-secondary --transport rccl --extra-config "pib_cg_single_reduction=0"
  run bench${N}_rccl_512_single_reduction 900 python bench.py --gpus $N --steps 10 --warmup 3 --no-cpu --no-secondary --transport rccl --extra-config "pib_cg_single_reduction=1"
fi
for g in 1 2 4 $N; do
  [ "$ndev" -ge "$g" ] && run scale_$g 900 python bench.py --gpus $g --steps 10 --warmup 3 --no-cpu --no-secondary --pmc off
done
echo "done: $OUT/summary.txt"

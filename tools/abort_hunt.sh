#!/bin/bash
# tools/abort_hunt.sh TAG: one cold-start run of the subset that has died inside pib_destroy now and then (docs/history/round4.md), with core
# dumps on; when it dies, the native backtraces of all threads (rocgdb on the core) go to gpurun_out/abort_hunt/TAG.bt.txt
TAG=${1:-run}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/abort_hunt; mkdir -p $O
ulimit -c unlimited
echo "PIB_TORCH_FIRST=${PIB_TORCH_FIRST:-1}"
echo "core_pattern: $(cat /proc/sys/kernel/core_pattern)" > $O/$TAG.log
rm -f /tmp/core* core*
timeout 800 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_periodic.py tests/test_gpu_multirank_loopback.py -x -q -v -s >> $O/$TAG.log 2>&1
rc=$?; echo "rc=$rc" >> $O/$TAG.log
if [ $rc -ne 0 ]; then
  C=$(ls -t core* /tmp/core* /var/lib/apport/coredump/* 2>/dev/null | head -1)
  echo "core file: $C $(ls -la $C 2>/dev/null)" >> $O/$TAG.log
  if [ -n "$C" ]; then
    timeout 300 /opt/rocm/bin/rocgdb -batch -ex "thread apply all bt 40" $(which python) $C > $O/$TAG.bt.txt 2>&1
  fi
fi
grep -E "passed|failed|Fatal|what\(\)" $O/$TAG.log | tail -3; echo "rc=$rc"

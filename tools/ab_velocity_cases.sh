#!/bin/bash
cp petibm_amd/lib/libpetibm_amd.so /tmp/keep.so
for v in old new old new; do
  cp petibm_amd/lib/var_$v.so petibm_amd/lib/libpetibm_amd.so
  echo "== $v"
  python bench.py --system velocity --grid 256 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  velocity_256 %.2f ms' % b['ms_per_step'])"
  timeout 800 python tools/config5_heaving_plate.py --steps 25 2>&1 | tail -1
  python examples/python/taylor_green_3d.py --nt 300 --every 100 2>&1 | tail -1
done
cp /tmp/keep.so petibm_amd/lib/libpetibm_amd.so

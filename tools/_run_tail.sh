cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/tail4
for v in base new base new; do
  cp petibm_amd/lib/var_$v.so petibm_amd/lib/libpetibm_amd.so
  echo "== $v"
  timeout 300 python tools/slab_probe.py 8 2>&1 < /dev/null | tail -2
  timeout 300 python tools/config4_cylinder_re550.py --nt 600 2>&1 < /dev/null | tail -2
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu --no-secondary 2>/dev/null < /dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('bench ms_per_step', d['ms_per_step'], 'iters', d.get('config', {}).get('iterations'), 'roofline_solve', d.get('roofline_solve', {}).get('frac'))"
done
cp petibm_amd/lib/var_new.so petibm_amd/lib/libpetibm_amd.so

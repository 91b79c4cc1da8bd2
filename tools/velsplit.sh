export TMPDIR=/tmp
P=gpurun_out/velsplit; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --output-format csv -d $P/a -o t -- python bench.py --system velocity --grid 256 --steps 2 --warmup 1 --kernel-reps 10 --no-cpu --extra-config "pib_fuse_velocity_product=0\npib_fuse_bicgstab_dots=0\npib_lean_bicgstab=0" > $P/a.log 2>&1
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/velsplit/a/**/*kernel_trace.csv',recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'k_vel' in n:
        agg[(n[:60], r.get('Grid_Size_X', r.get('Grid_Size','')))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(agg.items()):
    print(k, len(v), 'avg %.1f us min %.1f'%(sum(v)/len(v), min(v)))
PY
rm -rf $P/a

#!/bin/bash
# tools/build_asan.sh: petibm_amd/lib/var_asan.so = the library with its HOST code under AddressSanitizer (device code as ever); run with
#   LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 \
#   PIB_LIBRARY=petibm_amd/lib/var_asan.so python -m pytest ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/asan
for f in $ROOT/petibm_amd/csrc/*.cpp $ROOT/petibm_amd/csrc/*.hip; do
  b=$(basename $f); b=${b%.*}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -g -std=c++17 -ffp-contract=off -fPIC -fsanitize=address -fno-gpu-sanitize -Wno-unused-function -x hip -c $f -o /tmp/asan/$b.o ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fno-gpu-sanitize -shared-libsan -o $ROOT/petibm_amd/lib/var_asan.so /tmp/asan/*.o -lrccl
echo built $ROOT/petibm_amd/lib/var_asan.so

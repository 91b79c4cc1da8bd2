#!/bin/bash
# tools/build_variant.sh NAME SOURCE.hip [extra hipcc flags...]: petibm_amd/lib/var_NAME.so = the library with SOURCE's object
# rebuilt from the CURRENT file with the extra flags (tools/ab_trace.sh, tools/ab_*.sh swap such variants in)
set -e
NAME=$1; SRC=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/petibm_amd/lib/obj
BASE=$(basename $SRC .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function "$@" -x hip -c $ROOT/petibm_amd/csrc/$SRC -o /tmp/var_${NAME}_$BASE.o
OBJS=$(ls $OBJ/*.o | grep -v "/$BASE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/petibm_amd/lib/var_$NAME.so $OBJS /tmp/var_${NAME}_$BASE.o -lrccl
echo "built petibm_amd/lib/var_$NAME.so"

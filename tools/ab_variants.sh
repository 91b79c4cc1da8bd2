#!/bin/bash
# tools/ab_variants.sh "CMD" VARIANT...: run CMD (prints one line) with petibm_amd/lib/var_<V>.so in place of the library, in turn
CMD=$1; shift
cp petibm_amd/lib/libpetibm_amd.so /tmp/keep.so
for v in "$@"; do
  cp petibm_amd/lib/var_$v.so petibm_amd/lib/libpetibm_amd.so
  echo "== $v: $(bash -c "$CMD" 2>/dev/null | tail -1)"
done
cp /tmp/keep.so petibm_amd/lib/libpetibm_amd.so

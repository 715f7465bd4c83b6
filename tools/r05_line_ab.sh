#!/bin/bash
# On the GPU box: the headline line (no extras) with the base library and each variant under ab/.
cd ${GRAFT_REPO_ROOT:-.}
cp compression_amd/libtfc_hip.so /tmp/libtfc_hip.keep
line() { timeout 300 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_in_flight']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'enc', k['enc_kernel'], 'dec', k['dec_kernel'], json.dumps(k['stages']))"; }
echo "== base"; line; line
for v in ab/*/; do
  [ -f $v/libtfc_hip.so ] || continue
  cp $v/libtfc_hip.so compression_amd/libtfc_hip.so
  echo "== $(basename $v)"; line; line
done
cp /tmp/libtfc_hip.keep compression_amd/libtfc_hip.so

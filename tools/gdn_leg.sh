#!/bin/bash
# On the GPU box: the gdn leg's figures (forward cold / warm, backward passes, the (alpha, epsilon) variants), twice.
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2; do
timeout 300 python bench.py --leg gdn 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=d['gdn_fwd']
print('fwd', g['kernel_ms'], g['frac'], 'warm', g['warm']['kernel_ms'], 'copy cold/warm', g['copy_reference']['cold_ms'], g['copy_reference']['warm_ms'], 'bwd', g['backward']['kernel_ms'], g['backward']['passes_ms'])
for k,v in g.get('variants',{}).items():
    if isinstance(v,dict): print('  ', k, {kk:v[kk] for kk in v if kk in ('fwd_ms','kernel_ms','frac','warm_ms')}, v.get('backward',{}).get('passes_ms'))
"
done

#!/bin/bash
# On the GPU box: SQ counters for the conv kernels of tools/conv_probe.py (one pass, 8 SQ slots).
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmc_conv
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS \
  --output-format csv -d /tmp/pmc_conv -- python $R/tools/conv_probe.py > /tmp/pmc_conv.log 2>&1
tail -2 /tmp/pmc_conv.log | cut -c1-200
python - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_conv/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv_bf16_kernel' in r['Kernel_Name'] or 'conv_kernel' in r['Kernel_Name']:
            key = (r['Kernel_Name'][:60], r['Grid_Size'])
            rows[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, c in rows.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    wc = m.get('SQ_WAVE_CYCLES', 1)
    print(key[0], 'grid', key[1], 'launches', len(c['SQ_WAVE_CYCLES']))
    print('   ', ' '.join(f"{k.replace('SQ_', '')}={v / wc:.3f}" for k, v in sorted(m.items())), f'wave_cycles={wc:.3g}')
PY

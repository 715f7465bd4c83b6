#!/bin/bash
# On the GPU box: the whole GPU test tier, then the headline line without extras.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r05e_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05e_pytest.log
timeout 300 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r05e_line.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05e_line.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], json.dumps(d.get("kernels_ms_in_flight")))
PY

cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  echo "== bls2017"
  timeout 300 python bench.py --workload bls2017 --steps 32 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"
  echo "== bmshj2018"
  timeout 300 python bench.py --workload bmshj2018 --steps 128 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"
done
timeout 600 python -m pytest tests/test_models_gpu.py -x -q -m gpu 2>&1 | tail -3

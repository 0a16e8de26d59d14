cd $GRAFT_REPO_ROOT
python -m pytest tests/test_round4_gpu.py -q -x 2>&1 | tail -3
python -m pytest tests/test_config_golden.py tests/test_model_gpu.py tests/test_determinism_gpu.py -q -x -m gpu 2>&1 | tail -3
for i in 1 2; do
SPE_LINEAR_GROUP=0 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group off', r['ms_per_step'])"
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group on ', r['ms_per_step'], r.get('hbm_peak_allocated_gb'))"
done

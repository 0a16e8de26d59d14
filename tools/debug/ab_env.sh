# same-box A/B of one environment switch: bash tools/debug/ab_env.sh SPE_QKV_FUSED   (off = "0", on = default), two interleaved rounds
cd $GRAFT_REPO_ROOT
V=$1
for i in 1 2; do
env $V=0 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=0', round(r['ms_per_step'],2))"
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V on', round(r['ms_per_step'],2), r.get('hbm_peak_allocated_gb'))"
done

for r in 1 2 3; do
python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unpinned', round(r['ms_per_step'],2))"
taskset -c 0,1 python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2 cores ', round(r['ms_per_step'],2))"
taskset -c 0-3 python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4 cores ', round(r['ms_per_step'],2))"
done
nproc; lscpu | grep -i "numa\|model name\|socket" | head
python tools/debug/attn_time.py 2 4150 8 48 0.05 2>&1 | tail -7

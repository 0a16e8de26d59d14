# quick A/B pass on the GPU box: the attention launches in isolation, the attention tests, one bench line -> gpurun_out/<tag>/
O=gpurun_out/${1:-r06n}; mkdir -p $O
python tools/debug/attn_time.py > $O/attn_time.txt 2>&1
python -m pytest tests/test_round6_gpu.py tests/test_round5_gpu.py tests/test_round2_gpu.py -m gpu -q -x 2>&1 | tail -5 > $O/tests.txt
python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err
cat $O/attn_time.txt $O/tests.txt; python -c "
import json;r=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(r['value'],r['ms_per_step'],r['roofline']['avg_ms'])"

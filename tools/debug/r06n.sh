O=gpurun_out/r06n; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_data.hip -o /tmp/mfma_data && /tmp/mfma_data > $O/mfma_data.txt 2>&1
python tools/debug/mm_peak.py > $O/mm_peak.txt 2>&1
python tools/debug/attn_time.py > $O/attn_time.txt 2>&1
python -m pytest tests/test_round6_gpu.py tests/test_round5_gpu.py -m gpu -q -x 2>&1 | tail -5 > $O/tests.txt
python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err
cat $O/mfma_data.txt $O/mm_peak.txt $O/attn_time.txt $O/tests.txt; python -c "
import json;r=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(r['value'],r['ms_per_step'],r['roofline']['avg_ms'])"

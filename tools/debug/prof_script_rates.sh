# rocprofv3 kernel statistics of the launch scripts' configuration (3 encoder layers, 300 queries, drop_path 0.2, attn_drop 0.05, dropout 0.07)
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/psr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psr -- python $GRAFT_REPO_ROOT/bench.py --enc-layers 3 --queries 300 --drop-path 0.2 --attn-drop 0.05 --backbone-drop 0.07 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/psr.log 2>&1
cp $(find /tmp/psr -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/script_rates_kernel_stats.csv
tail -1 /tmp/psr.log | cut -c1-300

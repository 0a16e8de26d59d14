export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/dwp -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/dwp.log 2>&1
f=$(find /tmp/dwp -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/debug/decoder_window.py $f
python $GRAFT_REPO_ROOT/tools/step_windows.py $f

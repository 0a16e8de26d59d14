# Flash attention kernel variants (tools/ab.py builds, -DSPE_ABLATE) timed under rocprofv3 at cfg2 shapes: the committed evidence for the
# one-wave-per-SIMD / 512-register prototype (FLF_NW=4: accumulators in the AccVGPR half) and the barrier / mix-order choices.
#   python tools/ab.py base "" nw4 "-DFLF_NW=4" noskew "-DFLF_SKEW=0" mixh0 "-DFLF_MIXH=0" ; gpurun -- bash tools/debug/flash_variants.sh
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base nw4 noskew mixh0; do
  [ -f $R/build_ab/$v.so ] || continue
  rm -rf /tmp/fv_$v
  (cd /tmp && SPE_HIP_LIB=$R/build_ab/$v.so REP=5 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fv_$v -- python $R/tools/debug/flash_only.py > /tmp/fv_$v.log 2>&1)
  f=$(find /tmp/fv_$v -name "*kernel_stats.csv" | head -1)
  echo "== variant $v"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "talking_flash" in r["Name"] or "flash_merge" in r["Name"]:
        kv = "dV pass " if "true>" in r["Name"].split("(")[0][-8:] else ("forward " if "talking_flash" in r["Name"] else "merge   ")
        print("   %s %-70s calls %3s  avg %8.1f us  min %8.1f us" % (kv, r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done

# same-box A/B of two builds of the library: bash tools/debug/ab_lib.sh <old.so> [bench flags...]   (new = the tree's libspe_hip.so)
cd $GRAFT_REPO_ROOT
OLD=$1; shift
for i in 1 2; do
SPE_HIP_LIB=$GRAFT_REPO_ROOT/$OLD python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', round(r['ms_per_step'],2))"
python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', round(r['ms_per_step'],2))"
done

cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_rows.py -m gpu -q -x 2>&1 | tail -3
for T in 64 128 256; do
echo "== item size $T"
BDF_ITEM_SIZE=$T python bench.py --steps 40 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['test_rmse'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
done
echo "== all split 128"
BDF_ALL_SPLIT=1 BDF_ITEM_SIZE=128 python bench.py --steps 40 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['test_rmse'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
export TMPDIR=/tmp
for mode in normal allsplit; do
if [ $mode = allsplit ]; then export BDF_ALL_SPLIT=1; fi
(cd /tmp && BDF_ITEM_SIZE=128 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 20 --no-cpu-baseline > /tmp/log 2>&1)
python3 - <<PY
import csv,glob
print("$mode")
for f in glob.glob('/tmp/prof_$mode/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if float(r['Percentage'])>0.5: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
done

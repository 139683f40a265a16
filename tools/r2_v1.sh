# Large name spaces and the scratch bound: parity tests of the first-generation path, then 65 536 / 16 384 names at
# 1e9 pairs and config 4's slice with the default dispatch, and config 3 cut (default) against uncut (cap lifted).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r2v1}; mkdir -p $OUT; cd $R
(timeout 900 python -m pytest tests/test_gpu_twolevel.py tests/test_gpu_hot.py tests/test_gpu_fuzz.py tests/test_gpu_options.py tests/test_gpu_part2.py -x -q) > $OUT/pytest_v1.log 2>&1
tail -2 $OUT/pytest_v1.log
run() { timeout 300 python tools/sweep.py --samples $1 --pairs $2 --reps 5 --dists lognormal $3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('n=$1 names=$2 $3', 'avg_ms', round(j['avg_ms'],3), 'min_ms', round(j['min_ms'],3))" | tee -a $OUT/v1.txt; }
run 1e9 65536; run 1e9 16384; run 1.25e8 65536
run 1e9 65536 "--opt 6=1610612736"
for rep in 1 2; do run 1e9 1024; run 1e9 1024 "--opt 6=17179869184 --opt 7=1073741824"; done

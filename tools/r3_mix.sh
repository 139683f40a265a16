# Round 3: third-generation timings after a change + C3 with one sub-launch.  usage: bash tools/r3_mix.sh <tag>
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r3mix}; mkdir -p $OUT; cd $R
(timeout 600 python -m pytest tests/test_gpu_part3.py tests/test_gpu_part2.py -x -q) > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log
bash tools/r3_v3.sh $1 notest quick 2>&1 | grep -E "avg_ms|k_scatter|k_split|k_part_hist3|k_survey"
for o in "" "--opt 7=1073741824 --opt 6=4294967296"; do
timeout 300 python tools/sweep.py --samples 1e9 --pairs 1024 --reps 5 --dists lognormal $o 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('c3 1e9 names=1024 [$o]', 'avg_ms', round(j['avg_ms'],3), 'min_ms', round(j['min_ms'],3), 'scratch', j.get('scratch_bytes'))" | tee -a $OUT/c3.txt
done

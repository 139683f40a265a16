# Round 4: K1 on the exponential stream (the all-distribution guard showed it 20 % above lognormal at 2e8 samples)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-r4p}; mkdir -p $OUT
python tools/sweep.py --samples 1e9 --reps 8 --dists lognormal,lognormal,exponential,uniform,exponential,lognormal 2>&1 | cut -c1-150 | tee $OUT/k1_exp.txt
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for D in exponential lognormal; do
rm -rf /tmp/kc; rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d /tmp/kc -o t -- python $R/tools/sweep.py --samples 1e9 --reps 3 --dists $D > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py pmc /tmp/kc/t_results.db k_ingest_single | python -c "
import json,sys; j=json.load(sys.stdin); print('$D', {k:(round(v['avg']),round(v['avg_duration_us_profiled'])) for k,v in j['counters'].items()})" | tee -a $R/$OUT/k1_exp.txt
done
true

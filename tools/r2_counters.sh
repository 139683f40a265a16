# PMC counters of the second-generation mixed-ingest kernels at 1 024 Zipf names, 1e9 pairs: one rocprofv3 pass per set
# usage: bash tools/r2_counters.sh OUTDIR SHAPE
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUTD=$R/gpurun_out/${1:-r2g}; SH=${2:-0}; mkdir -p $OUTD; OUT=$OUTD/counters_shape$SH.txt; : > $OUT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pc$i
  timeout 300 rocprofv3 --pmc $set -d /tmp/pc$i -o t -- python $R/tools/sweep.py --samples 1e9 --pairs 1024 --reps 2 --opt 11=$SH --dists lognormal > /dev/null 2>&1
  for k in k_scatter k_part_hist2; do
    echo "== pmc $k [$set]" >> $OUT
    python $R/profiles/summarize_rocpd.py pmc /tmp/pc$i/t_results.db $k | grep -E '"[A-Z_]+": \{|"avg"|avg_duration' | tr -d '\n' | sed 's/},/\n/g' | sed 's/  */ /g' >> $OUT; echo >> $OUT
  done
done
cat $OUT

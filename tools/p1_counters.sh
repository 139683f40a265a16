# PMC counters of the mixed-ingest kernels at 1 024 Zipf names, 1e9 samples (hot-name windows on): one pass per set
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/p1_counters.txt; : > $OUT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pc$i
  timeout 300 rocprofv3 --pmc $set -d /tmp/pc$i -o t -- python $R/tools/sweep.py --samples 1e9 --pairs 1024 --reps 3 --dists lognormal > /dev/null 2>&1
  for k in k_scatter_samples k_part_hist; do
    echo "== pmc $k [$set]" >> $OUT
    python $R/profiles/summarize_rocpd.py pmc /tmp/pc$i/t_results.db $k | grep -E '"[A-Z_]+": \{|"avg"|avg_duration' | tr -d '\n' | sed 's/},/\n/g' | sed 's/  */ /g' >> $OUT; echo >> $OUT
  done
done
cat $OUT

# Tuning helper (GPU box): ablation timings + SQ counters of the partitioned mixed ingest.
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; OUT=$R/gpurun_out/abl.txt; : > $OUT
for f in ${ABL_FLAGS:-0 1 2 3}; do
  LH_DEBUG_FLAGS=$f timeout 300 rocprofv3 --kernel-trace -d /tmp/abl_$f -o t -- python $R/tools/abl.py > /dev/null 2>&1
  echo "== LH_DEBUG_FLAGS=$f" >> $OUT
  python $R/profiles/summarize_rocpd.py stats /tmp/abl_$f/t_results.db | grep -E "lh::k_p" | cut -c1-140 >> $OUT
done
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d /tmp/abl_pmc$i -o t -- python $R/tools/abl.py > /dev/null 2>&1
  for k in k_part_scatter k_part_hist; do
    echo "== pmc $k" >> $OUT
    python $R/profiles/summarize_rocpd.py pmc /tmp/abl_pmc$i/t_results.db $k | grep -E '"[A-Z_]+": \{|"avg"|avg_duration' | tr -d '\n' | sed 's/},/\n/g' >> $OUT; echo >> $OUT
  done
done
cat $OUT

# SQ counters of the third-generation kernels, 65 536 names, 1e9 pairs (one rocprofv3 --pmc pass per counter set).
# usage: bash tools/r3_pmc.sh <tag>
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r3pmc}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN"; do
  i=$((i+1)); rm -rf /tmp/pc$i
  timeout 300 rocprofv3 --pmc $set -d /tmp/pc$i -o t -- python $R/tools/sweep.py --samples 1e9 --pairs 65536 --reps 2 --dists lognormal > /dev/null 2>&1
  for k in k_scatter4 k_split_waves k_part_hist3; do
    echo "== pmc $k [$set]" >> $OUT/pmc.txt
    python $R/profiles/summarize_rocpd.py pmc /tmp/pc$i/t_results.db $k | grep -E '"[A-Z_]+": \{|"avg"|avg_duration' | tr -d '\n' | sed 's/},/\n/g' | sed 's/  */ /g' >> $OUT/pmc.txt; echo >> $OUT/pmc.txt
  done
done
cat $OUT/pmc.txt

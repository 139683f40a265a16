# Ablations + PMC of the third-generation kernels (tuning build), 65 536 names.  usage: bash tools/r3_abl.sh <tag>
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r3abl}; mkdir -p $OUT; cd $R
LIBT=$R/loghisto_amd/build/liblhgpu_tuning.so
cd /tmp; export TMPDIR=/tmp
for sz in 1.25e8 1e9; do
for bits in 0 1 2 4 8 12; do
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace -d /tmp/pk -o t -- python $R/tools/sweep.py --lib $LIBT --samples $sz --pairs 65536 --reps 3 --dists lognormal --opt 100=$bits > /dev/null 2>&1
echo "== n=$sz dbg=$bits" | tee -a $OUT/abl.txt
python $R/profiles/summarize_rocpd.py stats /tmp/pk/t_results.db | grep -E "k_scatter4|k_part_hist3|k_split" | cut -c1-130 | tee -a $OUT/abl.txt
done; done
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pc$i
  timeout 300 rocprofv3 --pmc $set -d /tmp/pc$i -o t -- python $R/tools/sweep.py --samples 1e9 --pairs 65536 --reps 2 --dists lognormal > /dev/null 2>&1
  for k in k_scatter4 k_split_records k_part_hist3; do
    echo "== pmc $k [$set]" >> $OUT/pmc.txt
    python $R/profiles/summarize_rocpd.py pmc /tmp/pc$i/t_results.db $k | grep -E '"[A-Z_]+": \{|"avg"|avg_duration' | tr -d '\n' | sed 's/},/\n/g' | sed 's/  */ /g' >> $OUT/pmc.txt; echo >> $OUT/pmc.txt
  done
done
cat $OUT/pmc.txt

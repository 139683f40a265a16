# NOTE: the ablation bits of k_split_waves / k_part_hist3 were removed after the measurements (commit a835df4 has
# them: profiles/r03_level1_experiments.txt records the results); lh_kernels_part3.h carries no ablation hooks.
# Ablations of k_split_waves / k_part_hist3 (tuning build), 65 536 names.  usage: bash tools/r3_abl23.sh <tag> <samples> [bits...]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r3abl23}; mkdir -p $OUT; cd $R; SZ=${2:-1.25e8}; shift; shift
LIBT=$R/loghisto_amd/build/liblhgpu_tuning.so
cd /tmp; export TMPDIR=/tmp
for bits in ${@:-0 1048576 2097152 4194304 8388608 33554432 67108864 134217728}; do
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace -d /tmp/pk -o t -- python $R/tools/sweep.py --lib $LIBT --samples $SZ --pairs 65536 --reps 3 --dists lognormal --opt 100=$bits > /dev/null 2>&1
echo "== n=$SZ dbg=$bits" | tee -a $OUT/abl.txt
python $R/profiles/summarize_rocpd.py stats /tmp/pk/t_results.db | grep -E "k_split|k_part_hist3|k_survey_count_h" | cut -c1-130 | tee -a $OUT/abl.txt
done

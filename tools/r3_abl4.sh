# NOTE: the ablation bits of k_scatter4 / k_scatter5 this script drives lived in the working tree of round 3 only (see
# profiles/r03_level1_experiments.txt for what they measured); lh_kernels_part3.h carries no ablation hooks.
# Ablations of k_scatter4 (tuning build), 65 536 names, 1e9 pairs.  usage: bash tools/r3_abl4.sh <tag> [bits...]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r3abl4}; mkdir -p $OUT; cd $R; shift
LIBT=$R/loghisto_amd/build/liblhgpu_tuning.so
cd /tmp; export TMPDIR=/tmp
for bits in ${@:-0 16 32 64 80 96 128 144 160}; do
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace -d /tmp/pk -o t -- python $R/tools/sweep.py --lib $LIBT --samples 1e9 --pairs 65536 --reps 3 --dists lognormal --opt 100=$bits > /dev/null 2>&1
echo "== dbg=$bits" | tee -a $OUT/abl.txt
python $R/profiles/summarize_rocpd.py stats /tmp/pk/t_results.db | grep -E "k_scatter[45]" | cut -c1-130 | tee -a $OUT/abl.txt
done

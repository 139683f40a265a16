# Round 4: host-fed lane launches in the lanes' own scratch blocks (LH_OPT_LANE_SCRATCH_BLOCKS) -- tests, then rates
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-r4r}; mkdir -p $OUT
python -m pytest tests/test_gpu_lane_blocks.py tests/test_gpu_options.py tests/test_gpu_pairs16.py -x -q 2>&1 | tail -4 | tee $OUT/pytest.log
python -m pytest tests/test_gpu_parity.py tests/test_cpp_host.py tests/test_gpu_part3.py -x -q 2>&1 | tail -3 | tee -a $OUT/pytest.log
for args in "16 8e8 1024 1048576" "16 8e8 65536 1048576" "8 4e8 1024 1048576"; do
  echo "== hostfed_native $args" | tee -a $OUT/hostfed.txt
  loghisto_amd/build/hostfed_native $args 2>&1 | tee -a $OUT/hostfed.txt
done

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/bisect
for lib in loghisto_amd/liblhgpu.so loghisto_amd/build/liblhgpu_tuning_visatom.so loghisto_amd/build/liblhgpu_tuning_tabexact.so; do
for i in 1 2 3 4; do
echo "== $lib run $i" | tee -a gpurun_out/bisect/log.txt
timeout 300 python tools/run_tests_with_lib.py $lib tests/test_gpu_part3.py -k "bad_ids or exact" 2>&1 | grep -E "passed|failed|AssertionError: " | tee -a gpurun_out/bisect/log.txt
done; done

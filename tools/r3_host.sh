# Round 3: merge / staging / counters tests, then the host-fed forms.  usage: gpurun -- 'bash tools/r3_host.sh <tag>'
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r3host}; mkdir -p $OUT; cd $R
(timeout 1200 python -m pytest tests/test_gpu_merge.py tests/test_gpu_counters.py tests/test_cpp_host.py "tests/test_gpu_parity.py" -x -q -m gpu) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
timeout 600 python -c "
import json, torch, bench, loghisto_amd as la
torch.cuda.set_device(0)
print(json.dumps(bench.run_hostfed(la)))" 2>&1 | tail -3 | tee $OUT/hostfed.json

R=$GRAFT_REPO_ROOT; cd $R
(timeout 600 python -m pytest tests/test_gpu_options.py tests/test_gpu_counters.py tests/test_gpu_twolevel.py -x -q) 2>&1 | tail -4
python tools/hostfed.py --pairs 1024 --threads 1,4,16 2>/dev/null | cut -c1-300
python tools/hostfed.py --threads 16 2>/dev/null | cut -c1-300
loghisto_amd/build/latency 300 4194304 65536 1
loghisto_amd/build/latency 300 4194304 65536 0
loghisto_amd/build/latency 1000 4194304 1024 1
loghisto_amd/build/latency 1000 4194304 1024 0

R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; OUT=$R/gpurun_out/r6M; mkdir -p $OUT
{ echo "== defaults (round 6): lognormal values, Zipf(1) names"
  timeout 900 python tools/call_size.py 1024,8192,65536 12 26 2>&1 | grep names
  echo "== the thresholds until round 6 (OPTS 17=131072,10=33554432,13=262144): first generation from 2^17, second from 2^25, third from 2^18"
  OPTS=17=131072,10=33554432,13=262144 timeout 900 python tools/call_size.py 1024,8192,65536 17 25 2>&1 | grep names
  echo "== defaults, streams that fall into few cells"
  for c in "constant zipf" "constant zero" "lognormal sorted"; do set -- $c; DIST=$1 IDS=$2 timeout 600 python tools/call_size.py 1024,65536 12 22 2>&1 | grep names; done
} | sed -e "s/'samples_fallback': 0}//" | cut -c1-230 | tee $OUT/small_calls.txt

# Where the region scatter's time goes: L2-resident input (compute only), counters, other distributions.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r2s}; mkdir -p $OUT; cd $R
run() { # lib shape dbg dist extra
  timeout 300 python tools/sweep.py --lib $1 --samples 1e9 --pairs ${5:-1024} --reps 3 --opt 9=1 --opt 11=$2 --opt 100=$3 --dists $4 ${6:-} 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('shape $2 dbg $3 $4 names ${5:-1024} ${6:-} avg_ms', round(j['avg_ms'],3))" | tee -a $OUT/ablate.txt
}
LIB=loghisto_amd/build/liblhgpu_tuning.so
for SH in 2 3; do for D in 0 512 516 532 534; do run $LIB $SH $D lognormal; done; done
for DIST in uniform exponential normal lognormal25; do run $LIB 2 0 $DIST; done
run $LIB 2 0 lognormal 1024 "--ids uniform"
for M in 64 256 4096 8192; do run $LIB 2 0 lognormal $M; run $LIB 0 0 lognormal $M; done
bash tools/r2_counters.sh ${1:-r2s} 2 > /dev/null 2>&1
cat $OUT/counters_shape2.txt | cut -c1-400

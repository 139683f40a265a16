R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r2u}; mkdir -p $OUT; cd $R
run() { # lib shape dbg dist
  timeout 300 python tools/sweep.py --lib $1 --samples 1e9 --pairs 1024 --reps 3 --opt 9=1 --opt 11=$2 --opt 100=$3 --dists $4 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('shape $2 dbg $3 $4 avg_ms', round(j['avg_ms'],3))" | tee -a $OUT/ablate.txt
}
LIB=loghisto_amd/build/liblhgpu_tuning.so
for D in 4 1028 36 1060 6 70; do run $LIB 2 $D lognormal; done

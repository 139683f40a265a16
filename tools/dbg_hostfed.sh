# Host-fed pairs, current library against another build on the same box (A/B after a suspected regression).
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
python tools/hostfed.py --pairs 1024 --threads 16 --samples 8e8 2>/dev/null | cut -c1-220
python tools/hostfed.py --pairs 1024 --threads 16 --samples 8e8 --lib loghisto_amd/build/liblhgpu_old.so 2>/dev/null | cut -c1-220
done
python tools/hostfed.py --threads 16 --samples 8e8 2>/dev/null | cut -c1-220

R=$GRAFT_REPO_ROOT; cd $R
bash tools/round.sh tests f_tests
bash tools/round.sh cells32 f_cells32
bash tools/round.sh p3spill f_p3spill
bash tools/round.sh profile f_profile

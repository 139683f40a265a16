R=$GRAFT_REPO_ROOT; cd $R
bash tools/round.sh tests f_tests
bash tools/round.sh profile f_profile

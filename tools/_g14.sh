cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 bash tools/profile_round.sh r04 > gpurun_out/profile_round_r04.log 2>&1; echo "profile_round rc=$?"; tail -2 gpurun_out/profile_round_r04.log
bash tools/final_check.sh

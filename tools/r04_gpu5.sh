# r04 GPU call 5: 2^24-row proofs - proofs in flight and virtual linear columns (same-box A/B)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/ab.sh gpurun_out/r04_ab24 2 "infl2||--log-rows 24 --steps 4 --warmup 1 --inflight 2" "infl3||--log-rows 24 --steps 4 --warmup 1 --inflight 3" "infl4||--log-rows 24 --steps 3 --warmup 1 --inflight 4" "virt2|MS_LDE_VIRTUAL=1|--log-rows 24 --steps 4 --warmup 1 --inflight 2" "virt3|MS_LDE_VIRTUAL=1|--log-rows 24 --steps 4 --warmup 1 --inflight 3"

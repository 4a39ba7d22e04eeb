# r04 GPU call 8: proofs in flight at 2^24 rows (4 / 6 / 8) and for BabyBear at 2^20 rows (8 / 12), same-box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/ab.sh gpurun_out/r04_ab24b 2 "infl4||--log-rows 24 --steps 3 --warmup 1 --inflight 4" "infl6||--log-rows 24 --steps 3 --warmup 1 --inflight 6" "infl8||--log-rows 24 --steps 2 --warmup 1 --inflight 8"
bash tools/ab.sh gpurun_out/r04_abbb 2 "bb8||--field 1 --steps 15 --warmup 2 --inflight 8" "bb12||--field 1 --steps 10 --warmup 2 --inflight 12" "bb6||--field 1 --steps 20 --warmup 2 --inflight 6"

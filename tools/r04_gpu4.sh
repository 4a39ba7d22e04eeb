# r04 GPU call 4: the exchange lab of VERDICT r3 #6, the rank probe with the replicated time broken down by kernel class, then the round's profile passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_gpu4
mkdir -p $O
step() { name=$1; shift; echo "== $name"; "$@"; rc=$?; echo "$name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ] || [ $rc -eq 139 ]; then echo "stopping after $name"; exit $rc; fi; }
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ntt_xchg_lab.hip -o /tmp/ntt_xchg_lab 2> $O/xchg_build.err
step xchg_lab timeout -k 10 120 /tmp/ntt_xchg_lab > $O/ntt_xchg_lab.log 2>&1
cat $O/ntt_xchg_lab.log
step rank_probe_24 timeout -k 10 600 python3 tools/shard_rank_probe.py --log-rows 24 --worlds 8 --dist 1 > $O/rank_probe_2p24.log 2>&1
cut -c1-1500 $O/rank_probe_2p24.log
step profile_round timeout -k 10 1100 bash tools/profile_round.sh r04 > $O/profile_round.log 2>&1
tail -5 $O/profile_round.log

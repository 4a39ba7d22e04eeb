cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_exp20
mkdir -p $O
for i in 1 2 3; do
timeout -k 10 300 python3 bench.py --steps 60 --warmup 4 --no-cpu-baseline --no-extras > $O/plain_$i.json 2> $O/plain_$i.err; echo "plain $i rc=$?"
done
for i in 1 2 3; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$i -- python3 bench.py --no-cpu-baseline --no-extras > $O/stats_$i.log 2>&1; echo "rocprof $i rc=$?"
done

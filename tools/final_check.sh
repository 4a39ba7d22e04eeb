# end-of-round check on the GPU box: whole GPU suite, the driver's bench line, smoke
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final_check
mkdir -p $O
if [ "$1" != "bench-only" ]; then
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_all.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_all.log
[ $rc -eq 0 ] || exit $rc
fi
timeout -k 10 800 python3 bench.py --no-cpu-2p24 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-300 $O/bench_default.json
python3 -c "
import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log

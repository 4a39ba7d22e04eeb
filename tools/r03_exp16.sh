set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_exp16
mkdir -p $O
for wg in 0 4 8 16 32 64 256; do
MS_READBACK_WGS=$wg timeout -k 10 300 python3 tools/io_probe2.py > $O/io_wg$wg.log 2>$O/io.err && echo "wgs=$wg $(cat $O/io_wg$wg.log)"
done

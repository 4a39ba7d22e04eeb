# The later NTT pass taken apart (profiles/r03_ntt_ablations.log).  The variant libraries are built HERE (the GPU box only runs them):
#   for v in "prod" "abl_hot -DMS_NTT_ABL_HOTLOAD" "abl_nostore -DMS_NTT_ABL_NOSTORE" "abl_both -DMS_NTT_ABL_HOTLOAD -DMS_NTT_ABL_NOSTORE"; do set -- $v; n=$1; shift
#     hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-pass-failed "$@" -x hip mini-stark_amd/csrc/ministark.cpp -o tools/_build/lib$n.so; done
#   python3 tools/ntt_phase_ts.py --build-only --out libts_base.so
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_abl
mkdir -p $O
rm -f $O/abl.log
for i in 1 2; do
for v in prod abl_hot abl_nostore abl_both; do
timeout -k 10 120 python3 tools/ntt_bench.py --lib tools/_build/lib$v.so --field 0 --log-rows 20 --reps 40 --tag $v --passes >> $O/abl.log 2>> $O/abl.err
done
done
grep tag $O/abl.log
timeout -k 10 120 python3 tools/ntt_phase_ts.py --out libts_base.so > $O/ts_base.json 2>> $O/abl.err

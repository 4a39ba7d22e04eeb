# r04 GPU call 6: lazy linear provenance (MS_LAZY_LINCOMB) same-box A/B - headline, 2^24 rows, wide AIR - then the parity cases that cover it on the real kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_gpu6
mkdir -p $O
step() { name=$1; shift; echo "== $name"; "$@"; rc=$?; echo "$name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ] || [ $rc -eq 139 ]; then echo "stopping after $name"; exit $rc; fi; }
step pytest timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "test_prove or wide or lincomb or closure or full_size_bit_exact or sharded_proof_on_gpu or ood or cubic or virtual" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
bash tools/ab.sh $O/ab 3 "lazy1|MS_LAZY_LINCOMB=1|--steps 30 --warmup 4" "lazy0|MS_LAZY_LINCOMB=0|--steps 30 --warmup 4"
bash tools/ab.sh $O/ab24 2 "lazy1_24|MS_LAZY_LINCOMB=1|--log-rows 24 --steps 3 --warmup 1" "lazy0_24|MS_LAZY_LINCOMB=0|--log-rows 24 --steps 3 --warmup 1"
for i in 1 2; do for v in 1 0; do MS_LAZY_LINCOMB=$v timeout -k 10 300 python3 tools/wide_bench.py > $O/wide_lazy${v}_$i.log 2>&1; echo "wide lazy=$v pass $i rc=$? $(tail -1 $O/wide_lazy${v}_$i.log | cut -c1-300)"; done; done

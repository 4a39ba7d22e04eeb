# round-3 batch 5: why SHIFT1 did not pay (SQ counters A/B), into-mode with non-coherent pinned memory
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_exp5
mkdir -p $O
for v in shift1 noshift1; do
  L=""; [ $v = noshift1 ] && L="--lib tools/libs/libministark_noshift1.so"
  timeout -k 10 120 python3 tools/ntt_bench.py --field 0 --log-rows 20 --reps 40 --tag $v $L > $O/ntt_$v.log 2>&1
  grep tag $O/ntt_$v.log
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_$v -- python3 tools/ntt_bench.py --field 0 --log-rows 20 --reps 3 --tag $v $L > $O/pmc_$v.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $O/pmc2_$v -- python3 tools/ntt_bench.py --field 0 --log-rows 20 --reps 3 --tag $v $L > $O/pmc2_$v.log 2>&1 || echo "pmc2 $v failed"
done
python3 tools/pmc_summary.py $O/pmc_shift1 $O/pmc_noshift1 $O/pmc2_shift1 $O/pmc2_noshift1 > $O/pmc_summary.txt 2>&1 || true
cat $O/pmc_summary.txt | head -60
timeout -k 10 300 python3 tools/io_probe2.py > $O/io_default.log 2>$O/io_default.err && cat $O/io_default.log
MS_PINNED_FLAGS=0x80000000 timeout -k 10 300 python3 tools/io_probe2.py > $O/io_noncoherent.log 2>$O/io_noncoherent.err && cat $O/io_noncoherent.log

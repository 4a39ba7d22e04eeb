set -e
B="timeout -k 10 200 python tools/ntt_bench.py --log-rows 20 --reps 10"
D=mini-stark_amd
for v in base shift swz tab; do
MS_NTT_MAXPAD=0 MS_NTT_KMAX=8 $B --lib $D/libms_$v.so --tag "$v,novirt,k8" 2>&1 | grep tag
done
for v in base shift tab; do
MS_NTT_TH512=0 $B --lib $D/libms_$v.so --tag "$v,virt32,[9,9],th256" 2>&1 | grep tag
$B --lib $D/libms_$v.so --tag "$v,virt32,[9,9],th512" 2>&1 | grep tag
MS_NTT_MAXRHO=0 MS_NTT_KMAX=8 $B --lib $D/libms_$v.so --tag "$v,virt8,[7,7,6]" 2>&1 | grep tag
done

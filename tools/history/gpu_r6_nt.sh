#!/bin/bash
# round 6: streaming (nt) accesses: nt1 = C / S loads of the band passes, nt2 = their S stores, nt3 = both, cnt = k_cost's C stores
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_band_nt.txt
echo "# second run: repeated A/B, + cnt = k_cost's C stores as nt, cnt_nt1 = cnt + nt1" >> $O
for i in 1 2 3; do bash tools/gpu_exp.sh "" nt1 cnt cnt_nt1 >> $O 2>&1; done
echo "== HH" >> $O
bash tools/gpu_exp.sh "--mode hh" nt1 >> $O 2>&1
tail -40 $O

#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06_gray_waves.txt; : > $O
echo "# k_cost for gray: 16 waves per workgroup (product) vs 8 (g8); 1080p D=128 64 pairs, and C4 (4K D=256, 16 pairs)" >> $O
bash tools/gpu_exp.sh "--channels 1" g8 >> $O 2>&1
bash tools/gpu_exp.sh "--channels 1 --width 3840 --height 2160 --disparities 256 --batch 16" g8 >> $O 2>&1
cat $O

export TMPDIR=/tmp
sed -e 's/for PH in 1 2 4; do/for PH in 2; do/' -e 's#O=gpurun_out/r06_clocks.txt#O=gpurun_out/r06_clocks_merged.txt#' tools/gpu_clock_probe.sh > /tmp/probe2.sh
echo "## unmerged (product)"; bash /tmp/probe2.sh | grep -v "^==\s*idle" | head -8
echo "## merged (8 compute waves)"; CAMD_LIB=$PWD/calibrating_amd/lib/dbg_merged.so bash /tmp/probe2.sh | head -8

#!/bin/bash
# Regenerates everything under profiles/ for one round from the tree as it is (run on the GPU box, from the repo root):
#     tools/refresh_profiles.sh r02
# 1. tools/prof.sh for the three benchmark layers, the 8-rooms-per-GPU size, the hierarchy build and the five
#    BASELINE.json configurations (tools/config_time.py)
#    (kernel-trace stats + separate PMC passes), copied to profiles/<round>_{rocprofv3_summary,kernels,pmc_traffic,command}_<tag>.*
# 2. the default `python bench.py` line (reads the traffic files of step 1) -> profiles/<round>_bench.json
# The copies land in gpurun_out/profiles_<round>/ as well, so that a `gpurun` call brings them home.
set -u
ROUND=${1:-r03}
ROOT=$PWD
DST=$ROOT/profiles
mkdir -p $DST gpurun_out/profiles_$ROUND

keep() {  # keep <tag>
    local src=$ROOT/gpurun_out/prof_${ROUND}_$1
    cp $src/summary.txt $DST/${ROUND}_rocprofv3_summary_$1.txt
    cp $src/kernels.json $DST/${ROUND}_kernels_$1.json
    cp $src/command.txt $DST/${ROUND}_command_$1.txt
    [ -f $src/traffic.json ] && cp $src/traffic.json $DST/${ROUND}_pmc_traffic_$1.json
}

for layer in 1to64 3to8 dw256; do
    tools/prof.sh ${ROUND}_$layer --layer $layer > /dev/null 2>&1
    keep $layer
done
PROF_STEPS=6 PROF_WARM=2 tools/prof.sh ${ROUND}_8rooms --layer 1to64 --rooms-per-gpu 8 > /dev/null 2>&1
keep 8rooms
PROF_NO_PMC=1 PROF_CMD="python $ROOT/tools/hier_time.py" tools/prof.sh ${ROUND}_hier > /dev/null 2>&1
keep hier
# BASELINE.json configurations: one step = PointHierarchy + forward + backward of every convolution of the model's graph
# (PIPE=1: the next batch's hierarchy one step ahead, as bench.py runs the configurations by default; the per-queue view of
# the same trace -- which kernels share the chip, how busy the convolutions' queue is -- goes beside the summary)
for cfg in cfg0 cfg1 cfg2 cfg3 cfg4; do
    PIPE=1 PROF_NO_PMC=1 PROF_WARM_DROP=6 PROF_CMD="python $ROOT/tools/config_time.py $cfg 20" tools/prof.sh ${ROUND}_$cfg > /dev/null 2>&1
    keep $cfg
    python tools/queues.py gpurun_out/prof_${ROUND}_$cfg/trace 0.3 > $DST/${ROUND}_queues_$cfg.txt 2>&1
done

# the full record (bench_details.json) and the compact line the driver parses (the LAST stdout line)
python bench.py --details $DST/${ROUND}_bench.json 2> gpurun_out/profiles_$ROUND/bench.err | tail -n 1 > $DST/${ROUND}_bench_line.json
cp $DST/${ROUND}_* gpurun_out/profiles_$ROUND/
tail -c 600 $DST/${ROUND}_bench_line.json

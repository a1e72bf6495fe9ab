#!/bin/bash
# Kernel trace + the three PMC passes of one bench command (run on the GPU box from the repo root):
#   tools/profile_bench.sh <tag> [bench.py arguments...]
# writes gpurun_out/prof_<tag>/{stats,fetch,write,sq}/ and the per-kernel table gpurun_out/prof_<tag>/kernel_table.csv
set -e
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args="--no-extras --no-cpu-baseline --steps 10 --warmup 3 $*"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out/stats -o t --output-format csv -- python3 $root/bench.py $args > $out/bench_under_rocprof.json 2> $out/stats.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o t --output-format csv -- python3 $root/bench.py $args > /dev/null 2> $out/fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $out/write -o t --output-format csv -- python3 $root/bench.py $args > /dev/null 2> $out/write.err
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $out/sq -o t --output-format csv -- python3 $root/bench.py $args > /dev/null 2> $out/sq.err
cd $root
python3 tools/kernel_table.py $(find $out/stats -name "*kernel_stats.csv") $(find $out/fetch -name "*counter_collection.csv") $(find $out/write -name "*counter_collection.csv") $(find $out/sq -name "*counter_collection.csv") > $out/kernel_table.csv
cp $(find $out/stats -name "*kernel_stats.csv") $out/kernel_stats.csv
head -8 $out/kernel_table.csv

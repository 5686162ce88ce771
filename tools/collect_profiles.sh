#!/bin/bash
# Run ON THE GPU BOX (via gpurun): bench + rocprofv3 kernel stats + PMC passes -> gpurun_out/prof_<tag>/
# usage: tools/collect_profiles.sh <tag>
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
STEPS=5
SSTEPS=40   # the kernel-trace runs: long enough to sit at the chip's steady (power-capped) clocks like bench.py's own timed region
ARGS="$R/bench.py --steps $SSTEPS --warmup 5 --no-cpu-baseline --single-mode --no-pmc"     # (--no-pmc: a profiled bench must not start its own counter passes)
PARGS="$R/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --single-mode --no-configs --no-kernel-events --no-pmc"
# kernel durations: single-stream run (what bench.py's HIP events time; with the two encoders overlapped on two streams a
# trace charges each kernel the time it shared the chip) -> kernel_stats.csv; the default two-stream command -> kernel_stats_two_stream.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $ARGS --no-configs --single-stream > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -- python3 $ARGS --no-configs > $O/stats2.log 2>&1
cp $(find $O/stats2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_two_stream.csv
rm -rf $O/stats2/*/*kernel_trace.csv
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $P | cut -d" " -f1)
  rocprofv3 --pmc $P --output-format csv -d $O/pmc_$N -- python3 $PARGS > $O/pmc_$N.log 2>&1
done
python3 $R/tools/pmc_summary.py $O/pmc_summary.json $((STEPS + 2)) $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES $O/pmc_TCC_HIT_sum > /dev/null
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/stats/*/*kernel_trace.csv   # large; the stats csv is what gets committed
# the judged bench line LAST, with this very collection's counters as `roofline.traffic` (bench.py reads profiles/traffic_latest.json
# and ignores it unless its kernel_sha matches the library's dominant-kernel sources)
cp $O/pmc_summary.json $R/profiles/traffic_latest.json
# (un-profiled: this run makes its own two --pmc passes before it touches the GPU and quotes THEM as roofline.traffic; the passes above
# stay in pmc_summary.json / profiles/traffic_latest.json as the cross-check and for the other counters)
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
tail -c 2500 $O/bench.json
ls $O

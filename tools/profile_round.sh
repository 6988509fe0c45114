#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + stats of the bench command, then two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE: they do not fit one pass on gfx950 and must not be combined with tracing domains other than
# kernel-trace).  Summaries land in gpurun_out/prof_<tag>/; tools/summarize_prof.py condenses them for profiles/.
set -u
TAG=${1:-r01}
CFG=${2:-cfg3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT   # (scratch of an earlier call with the same tag: its CSVs would sit next to this run's)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/bench.py --config $CFG --no-cpu-baseline --no-extras --no-parity > $OUT/bench_under_rocprof.log 2>&1
P="--config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-parity"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $P > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $P > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $P > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/pmc_inst -- python3 $REPO/bench.py $P > $OUT/pmc_inst.log 2>&1
cd $REPO && python3 tools/summarize_prof.py $OUT $TAG $CFG > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# gpurun copies at most 64 MiB back: the summary holds what the raw counter CSVs said, so only the kernel-stats CSV is kept
[ "${KEEP_RAW:-0}" = 1 ] || { rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_inst; find $OUT/kt -name '*kernel_trace.csv' -delete; }

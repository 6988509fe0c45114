#!/bin/bash
# rocprofv3 evidence for the workloads of tools/bench_shapes.py (DESIGN.md 4.1d rows other than the `.in.` configs): per shape a kernel
# trace with stats, then separate PMC passes for FETCH_SIZE and WRITE_SIZE (they do not fit one pass on gfx950; never combined with other
# trace domains).  Usage (GPU box): bash tools/profile_shapes.sh <tag> "<shape> <shape> ..."
set -u
TAG=${1:-r03}
SHAPES=${2:-"match_cfg3 match_utf8 long_1024 nibble_cfg3 chain_cfg3 literal_cfg2 multi6_cfg3 packed_cfg5"}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for sh in $SHAPES; do
  OUT=$REPO/gpurun_out/prof_${TAG}_$sh
  rm -rf $OUT   # (scratch of an earlier call with the same tag: its CSVs would sit next to this run's)
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/tools/bench_shapes.py --shape $sh --steps ${FX_PROF_STEPS:-30} --warmup ${FX_PROF_WARMUP:-10} > $OUT/bench_under_rocprof.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/tools/bench_shapes.py --shape $sh --steps 3 --warmup 1 > $OUT/pmc_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/tools/bench_shapes.py --shape $sh --steps 3 --warmup 1 > $OUT/pmc_write.log 2>&1
  if [ "${3:-}" = "sq" ]; then   # issue / wait / LDS-conflict counters and instruction counts of the dominant kernel (two more passes)
    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $REPO/tools/bench_shapes.py --shape $sh --steps 3 --warmup 1 > $OUT/pmc_sq.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/pmc_inst -- python3 $REPO/tools/bench_shapes.py --shape $sh --steps 3 --warmup 1 > $OUT/pmc_inst.log 2>&1
  fi
  (cd $REPO && python3 tools/summarize_shapes.py $OUT $TAG $sh) > $OUT/summary.txt 2>&1
  cat $OUT/summary.txt
  # gpurun copies at most 64 MiB back: the summary holds what the raw counter CSVs said, so only the kernel-stats CSV is kept
  [ "${KEEP_RAW:-0}" = 1 ] || { rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_inst; find $OUT/kt -name '*kernel_trace.csv' -delete; }
done

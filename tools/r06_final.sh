#!/bin/bash
# Round 6 final measurement pass (GPU box, through gpurun): part 1 = the four configs under rocprofv3 (kernel trace + stats over every launch of a bench run, separate PMC
# passes) + the bench lines; part 2 = the shapes of DESIGN.md 4.1d that changed this round; part 3 = soak of the fuzz tests + the multi-GPU plumbing at world size 1.
PART=${1:-1}
mkdir -p gpurun_out/r06_final
case $PART in
  1) for c in cfg3 cfg4 cfg2 cfg5; do bash tools/profile_round.sh r06_$c $c > gpurun_out/r06_final/prof_$c.log 2>&1; tail -25 gpurun_out/r06_final/prof_$c.log; done
     for c in cfg2 cfg4 cfg5 cfg3; do python bench.py --config $c > gpurun_out/r06_final/bench_$c.json 2> gpurun_out/r06_final/bench_$c.err; tail -c 600 gpurun_out/r06_final/bench_$c.json; echo; done
     python bench.py --steps 20 --warmup 5 > gpurun_out/r06_final/bench_driver_protocol.json 2> gpurun_out/r06_final/bench_driver_protocol.err; tail -c 400 gpurun_out/r06_final/bench_driver_protocol.json ;;
  2) FX_PROF_STEPS=600 FX_PROF_WARMUP=30 bash tools/profile_shapes.sh r06 "packed_cfg5 packed_cfg3" > gpurun_out/r06_final/prof_packed.log 2>&1; tail -30 gpurun_out/r06_final/prof_packed.log
     bash tools/profile_shapes.sh r06 "long_400 long_1024 long_chain_1024 long_4096 ragged_20 ragged_100 ragged_132 utf8_100 rows_16 rows_32 rows_64 nibble_rows_16" sq > gpurun_out/r06_final/prof_shapes.log 2>&1
     grep -E "^shape|time|traffic|frac|VALU" gpurun_out/r06_final/prof_shapes.log | head -120 ;;
  3) bash tools/soak.sh r06_soak "61 62 63" 120
     FXAMD_BENCH_FORCE_DIST=1 python bench.py --config cfg5 --no-cpu-baseline > gpurun_out/r06_final/bench_cfg5_rccl_world1.json 2> gpurun_out/r06_final/bench_cfg5_rccl_world1.err; tail -c 300 gpurun_out/r06_final/bench_cfg5_rccl_world1.json; echo
     python bench.py --gpus 1 --single-process > gpurun_out/r06_final/bench_single_process_1.json 2> gpurun_out/r06_final/bench_single_process_1.err; cat gpurun_out/r06_final/bench_single_process_1.json; tail -3 gpurun_out/r06_final/bench_single_process_1.err
     python bench.py --gpus 1 --single-process --config cfg5 --rows 1562500 > gpurun_out/r06_final/bench_single_process_cfg5_small.json 2>> gpurun_out/r06_final/bench_single_process_1.err; tail -c 300 gpurun_out/r06_final/bench_single_process_cfg5_small.json
     for s in ragged_255 ragged_200 ragged_80 utf8_132 utf8_255 match_cfg3 match_cfg5 nibble_cfg3 chain17_cfg3 literal_cfg2 match_cfg1x in_flags_rows_20; do bash tools/r06_job.sh r06_final shape:$s; done ;;
esac

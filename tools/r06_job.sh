#!/bin/bash
# round 6 GPU jobs, one script with named steps:  tools/r06_job.sh <tag> <step> [<step> ...]
# Steps write under gpurun_out/<tag>/.  Steps: suite, smoke, bench_<cfg>, driver, dist1, k:<pytest -k expr>, f:<test file>, shape:<name>,
#   ab:<cfg>:<ENV>[:reps] (bench.py config, interleaved with / without ENV=1 in one allocation), abs:<shape>:<ENV>[:reps] (tools/bench_shapes.py the same way)
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print(sys.argv[2], "value", round(d["value"]), "step_ms", round(d["ms_per_step"], 4), "kernel_ms", round(r.get("kernel_ms", 0), 4), "frac", round(r.get("frac", 0), 3),
          "parity", (d.get("parity") or {}).get("mismatches"), "settled", (d.get("settled") or {}).get("ms_per_step"))
except Exception as e:
    print(sys.argv[2], "no bench line:", e)
PY
}
for step in "$@"; do
  case $step in
    spec_tests) timeout 1500 python -m pytest tests -m gpu -x -q -k "speculative or config_rows_vs_oracle or exception_queues or few_exception or utf8_rows" > $OUT/spec_tests.log 2>&1; echo "spec_tests rc $?"; tail -5 $OUT/spec_tests.log ;;
    suite) timeout 3000 python -m pytest tests -m gpu -x -q --durations=25 > $OUT/pytest.log 2>&1; echo "suite rc $?"; grep -A27 "slowest" $OUT/pytest.log | head -30; tail -3 $OUT/pytest.log ;;
    stampf:*) IFS=: read -r _ shp obj <<< "$step"   # stampf:<bench_shapes name>:<ch>_1  phases of fx_search_fast (make -C forgex_amd/csrc stamp-fast STAMP_FAST_OBJ=<ch>_1)
      FXAMD_LIB=forgex_amd/libforgex_amd_stamp_fast_$obj.so python tools/stamp_one.py shape:$shp --fast --md > $OUT/stampf_${shp}.md 2> $OUT/stampf_${shp}.err; echo "stampf $shp rc $?"; cat $OUT/stampf_${shp}.md; tail -2 $OUT/stampf_${shp}.err ;;
    stamp:*) IFS=: read -r _ cfg obj extra <<< "$step"   # stamp:<cfg>:<ch>_<part>[:--flags-only] per-phase s_memtime shares of fx_search_one (make -C forgex_amd/csrc stamp-one STAMP_OBJ=<ch>_<part>)
      FXAMD_LIB=forgex_amd/libforgex_amd_stamp_one_$obj.so python tools/stamp_one.py $cfg $extra --md > $OUT/stamp_${cfg}${extra}.md 2> $OUT/stamp_${cfg}${extra}.err; echo "stamp $cfg rc $?"; cat $OUT/stamp_${cfg}${extra}.md; tail -2 $OUT/stamp_${cfg}${extra}.err ;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?"; tail -3 $OUT/smoke.log ;;
    ab_cfg4)   # interleaved A/B in one allocation: the speculative forward pass on / off
      for rep in 1 2 3; do
        python bench.py --config cfg4 --no-cpu-baseline --no-extras > $OUT/cfg4_spec_$rep.json 2> $OUT/cfg4_spec_$rep.err; line $OUT/cfg4_spec_$rep.json "cfg4 spec   $rep"
        FXAMD_NO_SPEC=1 python bench.py --config cfg4 --no-cpu-baseline --no-extras > $OUT/cfg4_nospec_$rep.json 2> $OUT/cfg4_nospec_$rep.err; line $OUT/cfg4_nospec_$rep.json "cfg4 nospec $rep"
      done
      python bench.py --config cfg4 --flags-only --no-cpu-baseline --no-extras > $OUT/cfg4_flags.json 2> $OUT/cfg4_flags.err; line $OUT/cfg4_flags.json "cfg4 flags-only spec"
      FXAMD_NO_SPEC=1 python bench.py --config cfg4 --flags-only --no-cpu-baseline --no-extras > $OUT/cfg4_flags_nospec.json 2> $OUT/cfg4_flags_nospec.err; line $OUT/cfg4_flags_nospec.json "cfg4 flags-only nospec" ;;
    bench_*) cfg=${step#bench_}; python bench.py --config $cfg > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err; line $OUT/bench_$cfg.json "bench $cfg" ;;
    dist1)   # every RCCL call of the multi-rank path at world size 1 (census, per-rank times, packed gather, gathered-shard parity need world > 1)
      FXAMD_BENCH_FORCE_DIST=1 python bench.py --config cfg5 --no-cpu-baseline > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err; line $OUT/bench_dist1.json "cfg5 rccl world 1"
      python3 -c "
import json
d=json.loads(open('$OUT/bench_dist1.json').read().strip().splitlines()[-1])
print('gathered_shards', d['parity'].get('gathered_shards'), 'rccl_ranks', d['rccl_ranks'], 'devices', d['devices'], 'distinct', d['devices_distinct'], 'per_rank', d['per_rank_ms_per_step'], 'gather', d['gather'], 'oracle parity', d['parity'].get('oracle'))" ;;
    driver) python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; line $OUT/bench_driver.json "driver protocol" ;;
    f:*) tf=${step#f:}; timeout 2400 python -m pytest tests/$tf -m gpu -x -q > $OUT/f_$tf.log 2>&1; echo "pytest $tf rc $?"; tail -6 $OUT/f_$tf.log ;;
    ab:*) IFS=: read -r _ cfg envv reps <<< "$step"; reps=${reps:-3}; case $envv in *=*) ;; *) envv="$envv=1" ;; esac
      for rep in $(seq 1 $reps); do
        python bench.py --config $cfg --no-cpu-baseline --no-extras > $OUT/ab_${cfg}_on_$rep.json 2> $OUT/ab_${cfg}_on_$rep.err; line $OUT/ab_${cfg}_on_$rep.json "$cfg default      $rep"
        env $envv python bench.py --config $cfg --no-cpu-baseline --no-extras > $OUT/ab_${cfg}_off_$rep.json 2> $OUT/ab_${cfg}_off_$rep.err; line $OUT/ab_${cfg}_off_$rep.json "$cfg $envv $rep"
      done ;;
    abc:*) IFS=: read -r _ cfg libp reps <<< "$step"; reps=${reps:-3}   # bench.py config with the default library / with another build of it (FXAMD_LIB), interleaved
      for rep in $(seq 1 $reps); do
        python bench.py --config $cfg --no-cpu-baseline --no-extras > $OUT/abc_${cfg}_new_$rep.json 2> $OUT/abc_${cfg}_new_$rep.err; line $OUT/abc_${cfg}_new_$rep.json "$cfg default library   $rep"
        env FXAMD_LIB=$libp python bench.py --config $cfg --no-cpu-baseline --no-extras > $OUT/abc_${cfg}_old_$rep.json 2> $OUT/abc_${cfg}_old_$rep.err; line $OUT/abc_${cfg}_old_$rep.json "$cfg $libp $rep"
      done ;;
    rounds:*) IFS=: read -r _ cfg nrows list <<< "$step"   # rounds:<cfg>:<rows>:<r1,r2,...>  grid rounds (FXAMD_HALF_ROUNDS; 0 = the built-in rule) at a given batch size, twice each, interleaved
      for rep in 1 2; do for r in $(echo $list | tr ',' ' '); do
        if [ $r = 0 ]; then python bench.py --config $cfg --rows $nrows --no-cpu-baseline --no-extras --no-parity > $OUT/rounds_${cfg}_${nrows}_${r}_$rep.json 2> $OUT/rounds_${cfg}_${nrows}_${r}_$rep.err
        else env FXAMD_HALF_ROUNDS=$r python bench.py --config $cfg --rows $nrows --no-cpu-baseline --no-extras --no-parity > $OUT/rounds_${cfg}_${nrows}_${r}_$rep.json 2> $OUT/rounds_${cfg}_${nrows}_${r}_$rep.err; fi
        line $OUT/rounds_${cfg}_${nrows}_${r}_$rep.json "$cfg rows $nrows rounds $r rep $rep"
      done; done ;;
    abe:*) IFS=: read -r _ cfg envv reps <<< "$step"; reps=${reps:-3}; case $envv in *=*) ;; *) envv="$envv=1" ;; esac   # bench.py config with / without an environment setting, interleaved
      for rep in $(seq 1 $reps); do
        python bench.py --config $cfg --no-cpu-baseline --no-extras --no-parity > $OUT/abe_${cfg}_on_$rep.json 2> $OUT/abe_${cfg}_on_$rep.err; line $OUT/abe_${cfg}_on_$rep.json "$cfg default      $rep"
        env $envv python bench.py --config $cfg --no-cpu-baseline --no-extras --no-parity > $OUT/abe_${cfg}_off_$rep.json 2> $OUT/abe_${cfg}_off_$rep.err; line $OUT/abe_${cfg}_off_$rep.json "$cfg $envv $rep"
      done ;;
    abs:*) IFS=: read -r _ sh envv reps <<< "$step"; reps=${reps:-2}; case $envv in *=*) ;; *) envv="$envv=1" ;; esac
      for rep in $(seq 1 $reps); do
        for arm in on off; do
          if [ $arm = on ]; then python tools/bench_shapes.py --shape $sh > $OUT/abs_${sh}_${arm}_$rep.json 2> $OUT/abs_${sh}_${arm}_$rep.err
          else env $envv python tools/bench_shapes.py --shape $sh > $OUT/abs_${sh}_${arm}_$rep.json 2> $OUT/abs_${sh}_${arm}_$rep.err; fi
          python3 - $OUT/abs_${sh}_${arm}_$rep.json "$sh $arm($envv) $rep" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-40s L %4d  %.4f ms  %.0f GB/s  frac %.3f  path %s  matches %s" % (sys.argv[2], d["row_len"], d["ms_per_step"], d["input_gbs"], d["frac_of_hbm_peak"], d["last_path"], d["matches"]))
except Exception as e:
    print(sys.argv[2], "no line:", e, open(sys.argv[1].replace(".json", ".err")).read()[-400:])
PY
        done
      done ;;
    abl:*) IFS=: read -r _ sh libp reps <<< "$step"; reps=${reps:-2}   # bench_shapes with the default library / with another build of it (FXAMD_LIB), interleaved
      for rep in $(seq 1 $reps); do
        for arm in on off; do
          if [ $arm = on ]; then python tools/bench_shapes.py --shape $sh > $OUT/abl_${sh}_${arm}_$rep.json 2> $OUT/abl_${sh}_${arm}_$rep.err
          else env FXAMD_LIB=$libp python tools/bench_shapes.py --shape $sh > $OUT/abl_${sh}_${arm}_$rep.json 2> $OUT/abl_${sh}_${arm}_$rep.err; fi
          python3 - $OUT/abl_${sh}_${arm}_$rep.json "$sh $arm(default / $libp) $rep" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-60s L %4d  %.4f ms  %.0f GB/s  frac %.3f  path %s  matches %s" % (sys.argv[2], d["row_len"], d["ms_per_step"], d["input_gbs"], d["frac_of_hbm_peak"], d["last_path"], d["matches"]))
except Exception as e:
    print(sys.argv[2], "no line:", e, open(sys.argv[1].replace(".json", ".err")).read()[-400:])
PY
        done
      done ;;
    k:*) expr=${step#k:}; timeout 2400 python -m pytest tests -m gpu -x -q -k "$expr" > $OUT/k_$(echo "$expr" | tr -c 'a-zA-Z0-9\n' '_').log 2>&1; echo "pytest -k '$expr' rc $?"; tail -4 $OUT/k_$(echo "$expr" | tr -c 'a-zA-Z0-9\n' '_').log ;;
    shape:*) sh=${step#shape:}; python tools/bench_shapes.py --shape $sh > $OUT/shape_$sh.json 2> $OUT/shape_$sh.err; python3 - $OUT/shape_$sh.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("shape %-16s L %4d  %.4f ms  %.0f GB/s  frac %.3f  path %s  matches %s" % (d["shape"], d["row_len"], d["ms_per_step"], d["input_gbs"], d["frac_of_hbm_peak"], d["last_path"], d["matches"]))
except Exception as e:
    print("shape: no line:", e, open(sys.argv[1].replace(".json", ".err")).read()[-400:])
PY
      ;;
    *) echo "unknown step $step" ;;
  esac
done

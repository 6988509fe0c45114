OUT=gpurun_out/r04_c8; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -k "long_rows or match_operator or match_one_launch" > $OUT/k_long.log 2>&1; echo "long/match tests rc $?"; tail -2 $OUT/k_long.log
for rep in 1 2 3; do
  for v in a b; do
    lib=forgex_amd/libforgex_amd.so; [ $v = b ] && lib=forgex_amd/libforgex_amd_b.so
    FXAMD_LIB=$PWD/$lib python tools/bench_shapes.py --shape match_long_1024 > $OUT/mlong_${v}_$rep.json 2>$OUT/mlong_${v}_$rep.err
    python3 -c "
import json;d=json.loads(open('$OUT/mlong_${v}_$rep.json').read().strip().splitlines()[-1]);print('match_long_1024', '$v', $rep, 'pipe3' if '$v'=='a' else 'two buffers', round(d['ms_per_step'],4),'ms', round(d['input_gbs']),'GB/s frac',round(d['frac_of_hbm_peak'],3),'path',d['last_path'],'matches',d['matches'])"
  done
done

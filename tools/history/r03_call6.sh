#!/bin/bash
# round 3, GPU call 6: many patterns with byte-level tables in the shared pass (tests + timing), rocprofv3 evidence for the shapes of
# DESIGN 4.1d and for configs 2 / 3 / 5
OUT=gpurun_out/r03_c6
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
python tools/exp_multi.py cfg4 > $OUT/multi_cfg4_bytes.txt 2>&1; cat $OUT/multi_cfg4_bytes.txt
FXAMD_MULTI_NO_BYTES=1 python tools/exp_multi.py cfg4 > $OUT/multi_cfg4_nobytes.txt 2>&1; cat $OUT/multi_cfg4_nobytes.txt
FXAMD_NO_MULTI=1 python tools/exp_multi.py cfg4 > $OUT/multi_cfg4_sequential.txt 2>&1; cat $OUT/multi_cfg4_sequential.txt
python tools/exp_multi.py cfg3 > $OUT/multi_cfg3.txt 2>&1; cat $OUT/multi_cfg3.txt
bash tools/profile_shapes.sh r03 "match_cfg3 match_utf8 match_cfg1x long_1024 long_400 nibble_cfg3 chain_cfg3 literal_cfg2 multi6_cfg3 packed_cfg5 ragged_255" > $OUT/profile_shapes.log 2>&1
grep -E "^== (traffic|step)" $OUT/profile_shapes.log
for cfg in cfg2 cfg3 cfg5; do bash tools/profile_round.sh r03_$cfg $cfg > $OUT/prof_$cfg.log 2>&1; tail -3 $OUT/prof_$cfg.log | cut -c1-300; done

#!/bin/bash
OUT=gpurun_out/r03_c11
mkdir -p $OUT
for cfg in cfg3 cfg4 cfg5; do
python tools/exp_multi.py $cfg > $OUT/multi_$cfg.txt 2>&1; grep -v amdgpu.ids $OUT/multi_$cfg.txt
FXAMD_NO_MULTI=1 python tools/exp_multi.py $cfg > $OUT/seq_$cfg.txt 2>&1; grep -v amdgpu.ids $OUT/seq_$cfg.txt
done
python -m pytest tests/test_gpu_parity.py -x -q -k "many_patterns" > $OUT/pytest_multi.log 2>&1; echo "pytest multi rc $?"; tail -2 $OUT/pytest_multi.log

#!/bin/bash
# round 3, GPU call 10: the many-pattern pass finishes its exception rows inside the launch (mixed-pattern queue); tests + timing
OUT=gpurun_out/r03_c10
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -x -q -k "many_patterns" > $OUT/pytest_multi.log 2>&1; echo "pytest multi rc $?"; tail -2 $OUT/pytest_multi.log
for seed in 41 42; do FX_FUZZ_SEED=$seed FX_FUZZ_GROUPS=40 python -m pytest tests/test_gpu_parity.py -x -q -k "many_patterns_fuzz" > $OUT/soak_$seed.log 2>&1; echo "soak $seed rc $?"; tail -1 $OUT/soak_$seed.log; done
python tools/exp_multi.py cfg4 > $OUT/multi.txt 2>&1; grep -v amdgpu.ids $OUT/multi.txt
FXAMD_MULTI_NO_INQ=1 python tools/exp_multi.py cfg4 > $OUT/multi_noinq.txt 2>&1; grep -v amdgpu.ids $OUT/multi_noinq.txt
FXAMD_NO_MULTI=1 python tools/exp_multi.py cfg4 > $OUT/seq.txt 2>&1; grep -v amdgpu.ids $OUT/seq.txt
python tools/exp_multi.py cfg3 > $OUT/multi_cfg3.txt 2>&1; grep -v amdgpu.ids $OUT/multi_cfg3.txt
FXAMD_NO_MULTI=1 python tools/exp_multi.py cfg3 > $OUT/seq_cfg3.txt 2>&1; grep -v amdgpu.ids $OUT/seq_cfg3.txt
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $OUT/pytest.log

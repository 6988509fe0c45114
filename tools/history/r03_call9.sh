#!/bin/bash
OUT=gpurun_out/r03_c9
mkdir -p $OUT
python tools/exp_multi.py cfg4 > $OUT/multi.txt 2>&1; grep -v amdgpu.ids $OUT/multi.txt
FXAMD_MULTI_NO_BYTES=1 python tools/exp_multi.py cfg4 > $OUT/multi_nobytes.txt 2>&1; grep -v amdgpu.ids $OUT/multi_nobytes.txt
FXAMD_NO_MULTI=1 python tools/exp_multi.py cfg4 > $OUT/seq.txt 2>&1; grep -v amdgpu.ids $OUT/seq.txt

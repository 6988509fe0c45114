prune() { find gpurun_out/prof_$1* \( -name "*kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*agent_info.csv" \) -delete 2>/dev/null; }
bash tools/profile_shapes.sh r04c "chain17_128 nibble_cfg3_flags utf8_256" sq > gpurun_out/r04c_shapes2.log 2>&1
prune r04c
grep -h "step under rocprof\|traffic per step\|^  void fx" gpurun_out/r04c_shapes2.log | head -30

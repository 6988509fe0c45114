# final evidence of round 4: rocprofv3 + PMC of the shapes the round's last kernels serve (tag r04c), then the four `.in.` configs again (tag r04f_<cfg>)
# (the raw traces and counter dumps are pruned once summarised: what travels back is bounded)
prune() { find gpurun_out/prof_$1* \( -name "*kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*agent_info.csv" \) -delete 2>/dev/null; }
bash tools/profile_shapes.sh r04c "chain17_cfg3 chain_cfg3 nibble_cfg3 match_chain_cfg3 long_chain_1024 long_400 long_1024 match_cfg1x in_flags_rows_20 match_rows_12" sq > gpurun_out/r04c_shapes.log 2>&1
prune r04c
for c in cfg3 cfg5 cfg4 cfg2; do bash tools/profile_round.sh r04f_$c $c > gpurun_out/r04f_$c.log 2>&1; prune r04f_$c; tail -3 gpurun_out/r04f_$c.log; done
grep -h "step under rocprof\|traffic per step" gpurun_out/r04c_shapes.log | head -60
du -sh gpurun_out

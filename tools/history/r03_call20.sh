#!/bin/bash
# round 3, GPU call 20: config 4's automata at other row lengths / occupancies, what its exception path costs (clean rows), and the instruction
# counts of its kernel (SQ_INSTS_* / SQ_ACTIVE_INST_* in two PMC passes)
OUT=$(pwd)/gpurun_out/r03_c20
mkdir -p $OUT
for rep in 1 2; do
  for s in utf8_192 utf8_192_clean utf8_192_flags utf8_128 utf8_128_flags utf8_96 utf8_64 utf8_64_flags; do
    python tools/bench_shapes.py --shape $s --steps 100 --warmup 30 > $OUT/${s}_$rep.json 2> $OUT/${s}_$rep.err
    python3 -c "
import json
d=json.loads(open('$OUT/${s}_$rep.json').read().strip().splitlines()[-1]); print('$s rep$rep us', round(d['ms_per_step']*1e3,2), 'input_GBs', round(d['input_gbs']), 'frac', round(d['frac_of_hbm_peak'],3), 'path', d['last_path'], 'matches', d['matches'])"
  done
done
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
P="--config cfg4 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-parity"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/pmc_a -- python3 $REPO/bench.py $P > $OUT/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_b -- python3 $REPO/bench.py $P > $OUT/pmc_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_c -- python3 $REPO/bench.py $P > $OUT/pmc_c.log 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc_a", "pmc_b", "pmc_c"):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob("gpurun_out/r03_c20/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "fx_search_one" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    for k in acc: print(d, k, "per launch %.0f" % (acc[k] / max(cnt[k], 1)), "launches", cnt[k])
PY

# experiment: config 5's 128-byte rows on the half-row kernel (64-byte halves, CH = 4) with the default cache policy on its loads
for rep in 1 2; do
  python bench.py --config cfg5 --no-cpu-baseline --no-extras > gpurun_out/r04_c26_a$rep.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('gpurun_out/r04_c26_a$rep.json').read().strip().splitlines()[-1]);print('one-launch', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d['parity'])"
  FXAMD_EXP_HALF128=1 python bench.py --config cfg5 --no-cpu-baseline --no-extras > gpurun_out/r04_c26_b$rep.json 2>gpurun_out/r04_c26_b$rep.err; python3 -c "
import json;d=json.loads(open('gpurun_out/r04_c26_b$rep.json').read().strip().splitlines()[-1]);print('half 128  ', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d['parity'])"
done

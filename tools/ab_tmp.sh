mkdir -p gpurun_out/r03_g5
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r03_g5/pytest.log 2>&1; tail -2 gpurun_out/r03_g5/pytest.log
timeout 300 python bench.py > gpurun_out/r03_g5/bench.json 2> gpurun_out/r03_g5/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_g5/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['path']['frac'], d['single_batch']['batch_latency_us'], d['single_batch']['kernel_us'])
PY

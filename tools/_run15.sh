timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2o_pytest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r2o_pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2o_bench.json 2> gpurun_out/r2o_bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r2o_bench.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"], d["latency"], d["path_l2_err"], d["cpu_baseline"]["value"])
PY
tail -2 gpurun_out/r2o_bench.err

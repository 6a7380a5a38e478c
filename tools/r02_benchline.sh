#!/bin/bash
O=gpurun_out/r02_benchline; mkdir -p $O
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.log 2> $O/bench_20.err; echo exit=$?; tail -1 $O/bench_20.log > $O/bench_20.json
timeout -k 10 500 python bench.py > $O/bench_default.log 2> $O/bench_default.err; echo exit=$?; tail -1 $O/bench_default.log > $O/bench_default.json
python3 - <<'PY'
import json
for f in ("gpurun_out/r02_benchline/bench_20.json", "gpurun_out/r02_benchline/bench_default.json"):
    d = json.load(open(f))
    print(f, "value %.4g ms/step %.4f" % (d["value"], d["ms_per_step"]), d["lm"], d["init"])
    print({k: d["roofline"][k] for k in ("avg_launch_us", "avg_launch_us_back_to_back", "frac", "fp64_frac", "latency_floor_us", "frac_of_floor", "traffic")})
    print("c5", d["c5"]["value"], d["c5"]["ms_per_step"], {k: d["c5"]["roofline"][k] for k in ("avg_launch_us", "fp64_tflops", "fp64_frac")})
    print("maps", {k: v for k, v in d["maps"].items() if k not in ("what",)})
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["dense_solve"]["value"])
PY

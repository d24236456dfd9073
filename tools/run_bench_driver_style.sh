cd $GRAFT_REPO_ROOT
t0=$(date +%s)
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_style.json 2> gpurun_out/bench_driver_style.err
t1=$(date +%s)
echo "bench wall $((t1-t0)) s"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_driver_style.json"))
print(d["value"], d["ms_per_step"], d["value_repeats"]["median"])
for k,v in d["legs"].items():
    if isinstance(v,dict):
        print(k, v.get("ms_per_step") or v.get("ms_per_frame"), v.get("Mvoxels_per_s"), v.get("kernels") if k.startswith("config_1024") else "", v.get("error",""))
    else:
        print(k, v)
PY

# Round 5, first contact: the repaired bench options, the adaptive pre-heat against round 4's fixed one (driver arguments).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_bench_options.py tests/test_gpu_device_ingress.py -m gpu -x -q 2>&1 | tail -5
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/driver_adaptive_$i.json 2> $O/driver_adaptive_$i.err; done
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --preheat-steps 100 --preheat-max-chunks 1 > $O/driver_fixed100_$i.json 2> $O/driver_fixed100_$i.err; done
python bench.py --no-cpu-baseline > $O/default_1.json 2> $O/default_1.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/d*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "%.1f" % (d["value"] / 1e6), [round(v / 1e6, 1) for v in d["runs"]["values"]], d["preheat_steps"],
              [round(v / 1e6, 1) for v in d["config"].get("preheat_chunk_rates", [])])
    except Exception as e:
        print(f, "ERR", e)
PY

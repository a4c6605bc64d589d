# Where does k_agents_mixed_wave's time go?  Variant builds that leave one phase out (results wrong, timing only), C5 as
# written, one part (the kernel alone on the whole batch) - run on the GPU box after building the variants in the build
# container:  for m in 0 1 2 4 8 3 15; do python -c "from bourse_amd import _build; _build.build(out='build_variants/mw_skip$m.so', defines=['BOURSE_AMD_MW_SKIP=$m'])"; done
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/mw_phase
for m in 0 1 2 8 3; do
  for parts in 1 8; do
    BOURSE_AMD_LIBRARY=$R/build_variants/mw_skip$m.so python $R/bench.py --workload C5M --no-cpu-baseline --repeats 0 --wave-parts $parts > $R/gpurun_out/mw_phase/skip${m}_p$parts.json 2> /dev/null
  done
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$R/gpurun_out/mw_phase/skip*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"] / 1e6, 2), {k: round(v["avg_launch_ms"] * 1e3, 1) for k, v in d["roofline"]["kernels"].items()})
    except Exception as e:
        print(f.split("/")[-1], "failed", e)
PY

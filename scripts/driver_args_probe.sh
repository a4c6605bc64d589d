# The driver's command line (bench.py --steps 20 --warmup 5): first (reported) region vs. the following ones, with and without
# the pre-heat.  GPU box.
probe() { echo "$*: $(python bench.py --no-cpu-baseline --repeats 4 --steps 20 --warmup 5 $* 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print([round(v/1e6,1) for v in d['runs']['values']])")"; }
probe --preheat-steps 0
probe --preheat-steps 30
probe --preheat-steps 100
probe --preheat-steps 300
probe --preheat-steps 100 --books 8192
probe --preheat-steps 0 --books 8192

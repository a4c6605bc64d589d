for pe in 0 8 0 8; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-every $pe 2>/tmp/err.txt | tail -1 > /tmp/o.json
  python - $pe <<'PY'
import sys,json
try:
    j=json.loads(open('/tmp/o.json').read()); print("profile_every", sys.argv[1], round(j["value"]/1e6,1), [round(v/1e6,1) for v in j["runs"]["values"]])
except Exception as e:
    print("profile_every", sys.argv[1], "ERR", e, open('/tmp/err.txt').read()[-300:])
PY
done

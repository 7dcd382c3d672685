for st in 1 2 3 4 6 8; do python bench.py --windows 64 --streams $st --steps 100 --warmup 40 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('streams', $st, 'it/s', round(d['value']), 'ms/step', round(d['ms_per_step'],4))"; done
for w in 128 256; do for st in 2 4 8; do python bench.py --windows $w --streams $st --steps 100 --warmup 40 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('windows', $w, 'streams', $st, 'it/s', round(d['value']), 'ms/step', round(d['ms_per_step'],4))"; done; done

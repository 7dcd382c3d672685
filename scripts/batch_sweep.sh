for w in 64 256 1024; do python bench.py --windows $w --steps 100 --warmup 40 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print($w, 'it/s', round(d['value']), 'ms', round(d['ms_per_step'],4), {k:round(v,1) for k,v in r['per_kernel_us'].items()})"; done
for w in 64 256; do python bench.py --windows $w --steps 100 --warmup 40 --no-cpu-baseline --streams 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('1stream', $w, 'it/s', round(d['value']), 'ms', round(d['ms_per_step'],4), {k:round(v,1) for k,v in r['per_kernel_us'].items()})"; done

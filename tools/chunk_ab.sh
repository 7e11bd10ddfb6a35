for ROUND in 1 2; do
for CH in 32 16 8; do
python bench.py --steps 10 --warmup 3 --no-side --no-cpu-baseline --chunk $CH 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('chunk $CH  %6.2f ms/step  mbx %.2f pw %.2f sep %.2f nms %.2f agg %.2f sum %.2f' % (d['ms_per_step'], k['mbx'], k['pw'], k['sep'], k['nms'], k.get('aggregate',0), sum(k.values())))"
done; done

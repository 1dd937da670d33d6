set -e
for rep in 1 2; do for h in 1 0; do
 GPTQ_HESS_HEADS=$h python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also > /tmp/o.json 2>/dev/null
 python -c "
import json; j=json.load(open('/tmp/o.json')); print('llama7b heads=$h', j['value'], j['ms_per_step'], j['phases']['hessian'], j['roofline']['frac'])"
done; done
for h in 1 0; do
 GPTQ_HESS_HEADS=$h python bench.py --workload llama65b --steps 2 --warmup 1 --no-cpu-baseline > /tmp/o.json 2>/dev/null
 python -c "
import json; j=json.load(open('/tmp/o.json')); print('llama65b heads=$h', j['value'], j['ms_per_step'], j['phases']['hessian'], j['roofline']['frac'])"
done

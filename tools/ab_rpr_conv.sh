# A/B of the regression training step: library (MIOpen) vs own implicit-GEMM decoder convolutions x training options
for o in "" siamese siamese,graph; do for v in miopen hip; do
timeout 150 python bench.py --config rpr_train --no-cpu-baseline --hip-opt RPR_CONV=$v --rpr-opts "$o" 2>/tmp/err.log | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '[$o]', d['value'], d['ms_per_step'])
except Exception as e: print('$v [$o] ERR', e, open('/tmp/err.log').read()[-400:])"
done; done

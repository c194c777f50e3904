#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', round(d['value'] / 1e9, 2), 'G', d['unit'], 'ms_per_step', round(d['ms_per_step'], 3), 'frac', round(d['roofline']['frac'], 3), 'traffic', d['roofline'].get('traffic'), 'cpu', round(d['cpu_baseline']['value'] / 1e6, 1), 'M', 'days/hr', d.get('model_days_per_hr'))
print({k: d[k] for k in ('metric', 'n_gpus', 'steps', 'warmup', 'scaling', 'dtype', 'vs_baseline')})"

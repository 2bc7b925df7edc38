#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --pmc off 2>/dev/null | grep '^{' | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('value %.1f' % r['value'], 'two_pipelines', r.get('two_pipelines'), 'alt', r.get('alt_precision',{}).get('value'), 'lat', r.get('latency_b1'))"

#!/bin/bash
export TMPDIR=/tmp
S=$(date +%s.%N)
python bench.py > /tmp/b.json 2> /tmp/b.err
E_=$(date +%s.%N)
echo "wall seconds: $(python -c "print(round($E_ - $S, 1))")"
grep -a "^{" /tmp/b.json | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['config']['network_launch'][:25], {k:(v.get('ms_per_step') if isinstance(v,dict) else None) for k,v in d['companions'].items()})"
tail -3 /tmp/b.err

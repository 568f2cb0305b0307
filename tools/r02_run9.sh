#!/bin/bash
set -u
O=gpurun_out/r02h; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -4 $O/pytest.log
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python -c "
import json; j=json.load(open('$O/bench.json')); print(j['value'], j['detail']['tlas_1000_instances'], j['roofline']['measured_copy_gbps'], j['detail']['config4_strong']['mrays'])"

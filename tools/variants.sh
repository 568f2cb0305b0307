#!/bin/bash
# usage: tools/variants.sh "<variants>" [perf_probe args]   — A/B kernel variants on the GPU box
V=$1; shift
for v in $V; do echo "== variant $v"; timeout 300 python tools/perf_probe.py --variant $v "$@" 2>&1 | grep -E "layout|Error|error"; done

#!/bin/bash
# Round 4, embedded triangles: the GPU suite, the interleaved A/B (hybrid copy derived with / without a triangle in each node's line), the counters of the bench's batches.
set -u
O=$PWD/gpurun_out/r04_run3
mkdir -p $O
export TMPDIR=/tmp
bash tools/r04_suite.sh r04_run3
( timeout 600 python tools/ab_configs.py --side 4096 --rounds 7 noemb=8192:8:0 emb=8192:0:0 noemb2=8192:8:0 emb2=8192:0:0 > $O/ab_embed.txt 2>&1 ); cat $O/ab_embed.txt
( timeout 300 python tools/ab_configs.py --side 2048 --rounds 7 noemb=8192:8:0 emb=8192:0:0 > $O/ab_embed_4m.txt 2>&1 ); cat $O/ab_embed_4m.txt
bash tools/prof_cmd.sh r04_embed python $PWD/tools/ab_probe.py --scene bistro --side 4096 --layout 10 --variants 0 --passes 2 > $O/counters_embed.txt 2>&1
grep -A3 -E "TCP_TCC_READ_REQ_sum|TCC_MISS_sum|TCC_EA0_RDREQ" $O/counters_embed.txt | grep -B1 -A2 "13, 2, 0" | head -60

#!/bin/bash
set -u
O=gpurun_out/r4_23; mkdir -p $O
timeout 200 python tools/r4/fused_warmup_probe.py 20 4 30 > $O/fused_warmup_2p20_4GiB.jsonl 2>/dev/null; cat $O/fused_warmup_2p20_4GiB.jsonl | cut -c1-420
timeout 200 python tools/r4/fused_warmup_probe.py 18 4 30 > $O/fused_warmup_2p18_4GiB.jsonl 2>/dev/null; cat $O/fused_warmup_2p18_4GiB.jsonl | cut -c1-420

#!/bin/bash
# Round 6, side figures on the shipped library: the PCIe-inclusive rate (the batch crosses PCIe inside update_parameters, as in the reference;
# never `value`), the acting latency table, K0 (BASELINE config 1) as a bench line.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6side; mkdir -p $OUT
sha256sum pointcloud_rl_amd/libpcrl_hip.so | tee $OUT/libpcrl_hip.sha256
F="--no-cpu-baseline --no-extra-workloads --no-experimental"
python3 bench.py --replay host $F > $OUT/bench_k1_replay_host.json 2> $OUT/bench_k1_replay_host.err; echo "replay host rc=$?"
python3 bench.py --replay fixed $F > $OUT/bench_k1_replay_fixed.json 2> $OUT/bench_k1_replay_fixed.err; echo "replay fixed rc=$?"
python3 tools/bench_acting.py > $OUT/acting_latency.txt 2> $OUT/acting_latency.err; echo "acting rc=$?"
cat $OUT/acting_latency.txt
for f in $OUT/bench_k1_replay_*.json; do python3 -c "import json,sys;d=json.loads([l for l in open('$f') if l.startswith('{')][-1]);print('$f', round(d['value'],1), round(d['ms_per_step'],4), d['config'].get('replay'))"; done

#!/bin/bash
# A captured step as two graphs (head: sampling launch + first encoder pass | the rest) against one graph: bit-identity test, then same-box
# alternations at K1, its 128 / 64 / 32-cloud shares and K3's 128-cloud share.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6split; mkdir -p $OUT
python -m pytest tests/test_update_step_gpu.py -m gpu -x -q -k "published_before" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
F="--no-cpu-baseline --no-extra-workloads --no-experimental"
line() { python3 -c "import json;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1', round(d['ms_per_step'],4))"; }
for rep in 1 2; do
  for hs in 0 1; do
    python3 bench.py $F --set-fused head_split=$hs > $OUT/k1_hs${hs}_$rep.json 2> $OUT/k1_hs${hs}_$rep.err; line $OUT/k1_hs${hs}_$rep.json
    for b in 128 64 32; do
      python3 bench.py $F --batch $b --set-fused head_split=$hs > $OUT/k1b${b}_hs${hs}_$rep.json 2> $OUT/k1b${b}_hs${hs}_$rep.err; line $OUT/k1b${b}_hs${hs}_$rep.json
    done
    python3 bench.py $F --workload k3 --batch 128 --set-fused head_split=$hs > $OUT/k3b128_hs${hs}_$rep.json 2> $OUT/k3b128_hs${hs}_$rep.err; line $OUT/k3b128_hs${hs}_$rep.json
  done
done

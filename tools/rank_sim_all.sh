#!/bin/bash
# one rank's step of an N-GPU run on ONE GPU with the collectives stubbed (tools/rank_sim.py): the compute side of the
# scaling prediction in DESIGN.md 6.  weak = configs[4]'s form (GCN, 1M rows per rank of an N-times larger graph),
# strong = configs[3]'s (AT, the 1M / 20M graph cut in N).   tools/rank_sim_all.sh > gpurun_out/r05/rank_sim.log
cd "$(dirname "$0")/.."
for n in 1 2 4 8; do
  echo -n "weak GCN   "; timeout -k 10 400 python tools/rank_sim.py $n --gnn_type GCN --no-cpu-baseline --no-secondary 2>&1 | grep "world="
done
for n in 1 2 4 8; do
  echo -n "strong AT  "; timeout -k 10 400 python tools/rank_sim.py $n --scaling strong --no-cpu-baseline --no-secondary 2>&1 | grep "world="
done

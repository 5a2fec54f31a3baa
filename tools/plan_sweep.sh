#!/bin/bash
# Experiment driver (GPU box): bench.py through the stamped twin library, whose regressor plan can be overridden from the
# environment (SPSP_PATCH_TILE / SPSP_PATCH_DB / SPSP_PATCH_RING = four comma-separated values, one per residual stage;
# SPSP_RING_TILE, if exported by the caller, selects the stride-2 kernels' tiles by RingTile enum value).
# Prints throughput and the per-kernel isolated times of every configuration given as "TILE DB RING" triples.
cd "$(dirname "$0")/.."
export SPSP_LIB_PATH=$PWD/soccerplayershapepose_amd/csrc/libspsp_stamps.so
i=0
while [ $# -ge 3 ]; do
  export SPSP_PATCH_TILE=$1 SPSP_PATCH_DB=$2 SPSP_PATCH_RING=$3; shift 3
  i=$((i+1))
  timeout -k 10 200 python bench.py --steps 150 --no-alt-precision --cpu-fit-sample 1 --cpu-threads 16 > gpurun_out/sweep_$i.json 2> gpurun_out/sweep_$i.err || { echo "config $i failed"; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/sweep_$i.json").read().strip().splitlines()[-1])
k=d["kernels_isolated_us_per_step"]
print("tile=$SPSP_PATCH_TILE db=$SPSP_PATCH_DB ring=$SPSP_PATCH_RING :", round(d["value"]), "players/s; patch<false>", k["conv3x3_patch_kernel<false>"], "patch<true>", k["conv3x3_patch_kernel<true>"], "ring", k["conv_ring_kernel"], "parity", d["parity"]["pre_fit_max_m"])
PY
done

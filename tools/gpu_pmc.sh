#!/bin/bash
# rocprofv3 PMC passes over tools/regressor_only.py (one batch in flight): SQ counters, FETCH_SIZE, WRITE_SIZE -- each in its
# own run, no tracing beside the counters.  usage: tools/gpu_pmc.sh <tag> [sq] [fetch] [write] [fetch_tp] [write_tp] [fetch_skin16] [write_skin16]   (writes CSVs under gpurun_out/)
set -o pipefail
tag=${1:-rXX}; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for what in "$@"; do
  case $what in
    sq) ctrs="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; cmd="$root/tools/regressor_only.py 32 4 f16";;
    fetch) ctrs="FETCH_SIZE"; cmd="$root/tools/pipeline_once.py";;
    write) ctrs="WRITE_SIZE"; cmd="$root/tools/pipeline_once.py";;
    fetch_tp) ctrs="FETCH_SIZE"; cmd="$root/tools/pipeline_once.py 32 4 8 throughput";;      # the kernels only plan "throughput" launches (conv_rs)
    write_tp) ctrs="WRITE_SIZE"; cmd="$root/tools/pipeline_once.py 32 4 8 throughput";;
    fetch_skin16) ctrs="FETCH_SIZE"; cmd="$root/tools/pipeline_once.py 32 4 16";;             # experiment: 16 players per skin workgroup
    write_skin16) ctrs="WRITE_SIZE"; cmd="$root/tools/pipeline_once.py 32 4 16";;
    *) echo "unknown pass $what"; exit 2;;
  esac
  rocprofv3 --pmc $ctrs -d "$out/${tag}_pmc_$what" -- python3 $cmd > "$out/${tag}_pmc_$what.log" 2>&1 || exit 1
  db=$(find "$out/${tag}_pmc_$what" -name '*.db' | head -1)
  python3 "$root/profiles/export_rocpd.py" counter "$db" "$out/${tag}_pmc_$what.csv" || exit 1
  rm -rf "$out/${tag}_pmc_$what"
done
echo "pmc passes exported for $tag"

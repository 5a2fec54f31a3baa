#!/bin/bash
# rocprofv3 kernel traces whose summaries are tracked under profiles/ (run on the GPU box through gpurun):
#   (a) one batch in flight:  tools/regressor_only.py 32 4 f16   -> <tag>_kernel_stats_regressor_only.csv
#   (c) (a) in plan "throughput" (narrow persistent grids)      -> <tag>_kernel_stats_regressor_only_throughput_plan.csv
#   (b) four batches in flight, eager: bench.py --no-graphs ...  -> <tag>_kernel_stats_bench_eager.csv
#   (d) bench.py's own per-kernel pass, plan "throughput" / "latency" (bench.py --profile-pass: one batch in flight through the WHOLE
#       path, a synchronize per step) -> <tag>_kernel_stats_isolated_{throughput,latency}_plan.csv: what roofline.rocprof cross-checks
# usage: tools/gpu_profile.sh <tag>      (writes under gpurun_out/)
set -o pipefail
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/${tag}_kt_a" -- python3 "$root/tools/regressor_only.py" 32 4 f16 > "$out/${tag}_kt_a.log" 2>&1 || exit 1
db=$(find "$out/${tag}_kt_a" -name '*.db' | head -1)
python3 "$root/profiles/export_rocpd.py" stats "$db" "$out/${tag}_kernel_stats_regressor_only.csv" || exit 1
rocprofv3 --kernel-trace --stats -d "$out/${tag}_kt_c" -- python3 "$root/tools/regressor_only.py" 32 4 f16 throughput > "$out/${tag}_kt_c.log" 2>&1 || exit 1
db=$(find "$out/${tag}_kt_c" -name '*.db' | head -1)
python3 "$root/profiles/export_rocpd.py" stats "$db" "$out/${tag}_kernel_stats_regressor_only_throughput_plan.csv" || exit 1
rm -rf "$out/${tag}_kt_c"
for plan in throughput latency; do
  rocprofv3 --kernel-trace --stats -d "$out/${tag}_kt_d" -- python3 "$root/bench.py" --profile-pass $plan > "$out/${tag}_kt_d_$plan.log" 2>&1 || exit 1
  db=$(find "$out/${tag}_kt_d" -name '*.db' | head -1)
  python3 "$root/profiles/export_rocpd.py" stats "$db" "$out/${tag}_kernel_stats_isolated_${plan}_plan.csv" || exit 1
  rm -rf "$out/${tag}_kt_d"
done
rocprofv3 --kernel-trace --stats -d "$out/${tag}_kt_b" -- python3 "$root/bench.py" --no-graphs --no-alt-precision --no-cpu-baseline --steps 60 > "$out/${tag}_kt_b.json" 2> "$out/${tag}_kt_b.log" || exit 1
db=$(find "$out/${tag}_kt_b" -name '*.db' | head -1)
python3 "$root/profiles/export_rocpd.py" stats "$db" "$out/${tag}_kernel_stats_bench_eager.csv" || exit 1
rm -rf "$out/${tag}_kt_a" "$out/${tag}_kt_b"
echo "profiles exported for $tag"

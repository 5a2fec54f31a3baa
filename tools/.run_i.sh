timeout -k 10 400 python -m pytest tests -m gpu -q -s -p no:cacheprovider -k "silhouette_in_the_loss" > gpurun_out/g3i_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; grep -n "silhouette fit\|passed\|failed" gpurun_out/g3i_tests.log | head
if [ $rc -lt 124 ]; then
  timeout -k 10 300 python bench.py > gpurun_out/g3i_bench.json 2> gpurun_out/g3i_bench.err; echo "bench rc=$?"
  timeout -k 10 200 python tools/ab_plan.py 32 20 > gpurun_out/g3i_ab.txt 2>&1; echo "ab rc=$?"; tail -8 gpurun_out/g3i_ab.txt
  timeout -k 10 300 python tools/next_rows_bench.py --cpu > gpurun_out/g3i_next_rows.json 2> gpurun_out/g3i_next_rows.err; echo "next rows rc=$?"
  timeout -k 10 400 tools/gpu_profile.sh g3i; echo "profile rc=$?"
fi

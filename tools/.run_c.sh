timeout -k 10 600 python -m pytest tests -m gpu -q -s -p no:cacheprovider -k "backbone_stages or regressor_against or regressor_f16 or batch_72 or end_to_end or multi_view_fit_at" > gpurun_out/g3c_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/g3c_tests.log
if [ $rc -lt 124 ]; then
  timeout -k 10 300 python tools/ab_plan.py 32 20 > gpurun_out/g3c_ab.txt 2>&1; echo "ab rc=$?"; cat gpurun_out/g3c_ab.txt
  timeout -k 10 400 python bench.py --no-alt-precision --no-cpu-baseline > gpurun_out/g3c_bench.json 2> gpurun_out/g3c_bench.err; echo "bench rc=$?"
fi

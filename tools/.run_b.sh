timeout -k 10 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider -k "backbone_stages or regressor or batch_72 or batch_256 or end_to_end or north_star or multi_view_fit_at or pipeline_streams" > gpurun_out/g3b_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/g3b_tests.log
if [ $rc -lt 124 ]; then
  timeout -k 10 400 python bench.py --no-alt-precision > gpurun_out/g3b_bench.json 2> gpurun_out/g3b_bench.err; echo "bench rc=$?"
  timeout -k 10 300 tools/gpu_pmc.sh g3b sq; echo "pmc rc=$?"
  python tools/pmc_summary.py gpurun_out/g3b_pmc_sq.csv > gpurun_out/g3b_pmc_sq_summary.txt 2>&1
fi

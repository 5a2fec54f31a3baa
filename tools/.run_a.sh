timeout -k 10 1000 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r3a_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r3a_tests.log
if [ $rc -lt 124 ]; then
  timeout -k 10 400 python bench.py > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; echo "bench rc=$?"
  timeout -k 10 400 tools/gpu_profile.sh r3a; echo "profile rc=$?"
fi

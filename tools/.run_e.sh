timeout -k 10 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider -k "vjp or silhouette or regressor_against or stall" > gpurun_out/g3e_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 gpurun_out/g3e_tests.log
if [ $rc -lt 124 ]; then
  timeout -k 10 300 tools/gpu_pmc.sh g3e fetch write; echo "pmc rc=$?"
fi

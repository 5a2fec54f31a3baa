timeout -k 10 500 python tools/ab_pipeline.py 200 3 > gpurun_out/g3d_abp.txt 2>&1; echo "abp rc=$?"; cat gpurun_out/g3d_abp.txt | tail -8

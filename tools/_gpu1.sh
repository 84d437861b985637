set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_sapg_long.py -x -q -s > gpurun_out/r4_sapg_long.log 2>&1 || { tail -60 gpurun_out/r4_sapg_long.log; exit 1; }
tail -15 gpurun_out/r4_sapg_long.log
for k in gaussian moffat laplace; do
  timeout -k 10 300 python tools/run_gaussian_demo.py --kind $k > gpurun_out/r4_demo_$k.log 2>&1 || { tail -30 gpurun_out/r4_demo_$k.log; exit 1; }
  cat gpurun_out/r4_demo_$k.log
done

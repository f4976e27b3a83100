# round 6, GPU call l: the new stub test, smoke(), the whole GPU suite once more on the final tree.
mkdir -p gpurun_out/r6l && cd /root/repo
python -m pytest tests/test_integration_stub.py -q > gpurun_out/r6l/stub.log 2>&1; echo "stub rc=$?"; tail -3 gpurun_out/r6l/stub.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6l/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r6l/smoke.log
python -m pytest tests -m gpu -x -q > gpurun_out/r6l/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r6l/gpu_suite.log

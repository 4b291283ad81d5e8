python -m pytest tests -x -q -m gpu > gpurun_out/r4_full5.log 2>&1; tail -4 gpurun_out/r4_full5.log
cp gpurun_out/parity_gpu.json gpurun_out/parity_gpu_full_r4.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2

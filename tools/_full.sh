timeout 1500 python -m pytest tests -q -m gpu --timeout 300 2>&1 | tail -5
echo "== tile path"; EHR_FUSED_PATH=tile timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_fast.py tests/test_gpu_solver.py -q -m gpu --timeout 300 2>&1 | tail -3
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2

cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_parity.py -q --timeout 300 -k "sharded_early_stop or bench_multi_rank" 2>&1 | tail -60

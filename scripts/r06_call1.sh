#!/bin/bash
# round 6, first GPU call: the optional scratch plane, the empty-part merge, the read probe, the bench lines with their own
# parity blocks (N = 1, two gloo ranks, one RCCL rank), the repaired copy baseline of pack_bench
O=gpurun_out/r06; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 600 -k "slat_plane or merge_stop or read_probe or bench_ or overlap_streams or two_streams" > $O/pytest_call1.log 2>&1; echo "pytest rc $?" >> $O/pytest_call1.log
tail -15 $O/pytest_call1.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "rc $?" >> $O/bench_default.err
cat $O/bench_default.json; tail -3 $O/bench_default.err
timeout 300 scripts/probes/_build/pack_bench > $O/pack_bench.log 2>&1; cat $O/pack_bench.log

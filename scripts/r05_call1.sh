#!/bin/bash
# round 5, first GPU call: the scratch-plane test, the same-allocation A/B of the plane, the new bench line (N = 1, 2 gloo ranks)
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "slat_plane or one_launch_per_step or split_api or kernel" > $O/pytest_slat.log 2>&1; echo "pytest rc $?" >> $O/pytest_slat.log
L=geocalib_amd/lib/libgeocalib_hip.so
timeout 900 python scripts/variant_probe.py --models pinhole,simple_radial,radial,simple_divisional --reps 3 r04=geocalib_amd/lib/variants/r04.so off=$L@0@0 auto=$L@0@-1 on=$L@0@1 > $O/variant_slat.log 2>&1; echo "rc $?" >> $O/variant_slat.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "rc $?" >> $O/bench_default.err
timeout 600 python bench.py --gpus 2 --backend gloo --batch 256 --steps 5 --warmup 2 --cpu-sample 16 > $O/bench_2gloo.json 2> $O/bench_2gloo.err; echo "rc $?" >> $O/bench_2gloo.err
timeout 600 python bench.py --gpus 2 --backend gloo --batch 256 --steps 5 --warmup 2 --cpu-sample 16 --shared-group 16 > $O/bench_2gloo_shared.json 2> $O/bench_2gloo_shared.err; echo "rc $?" >> $O/bench_2gloo_shared.err
tail -3 $O/pytest_slat.log; cat $O/variant_slat.log; cat $O/bench_default.json; tail -2 $O/bench_default.err; cat $O/bench_2gloo.json; tail -2 $O/bench_2gloo.err; cat $O/bench_2gloo_shared.json; tail -2 $O/bench_2gloo_shared.err

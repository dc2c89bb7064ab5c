#!/bin/bash
# round 6, second GPU call: simple_divisional with its two remaining per-pixel rare cases
# moved behind the wave-uniform patch (same-allocation A/B against the build without, bits compared)
O=gpurun_out/r06; mkdir -p $O
export TMPDIR=/tmp
L=geocalib_amd/lib/libgeocalib_hip.so; V=geocalib_amd/lib/variants
timeout 900 python scripts/variant_probe.py --models simple_divisional --reps 3 new=$L old=$V/divbr0.so new_noplane=$L@0@0 old_noplane=$V/divbr0.so@0@0 > $O/variant_div.log 2>&1; echo "rc $?" >> $O/variant_div.log
cat $O/variant_div.log
GCLM_LIB_PATH=$PWD/$V/divbr0.so python scripts/dump_results.py $O/bits_old.npz simple_divisional 2>&1 | tail -1
python scripts/dump_results.py $O/bits_new.npz simple_divisional 2>&1 | tail -1
python - <<PY >> $O/variant_div.log
import numpy as np
a, b = np.load("$O/bits_old.npz"), np.load("$O/bits_new.npz")
bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
print(f"bits: {len(a.files)} tensors compared between the two builds, {len(bad)} differ", bad[:6])
PY
tail -1 $O/variant_div.log; rm -f $O/bits_old.npz $O/bits_new.npz
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 900 -k "divisional or randomised or other_models or bad_images" > $O/pytest_div.log 2>&1; echo "pytest rc $?" >> $O/pytest_div.log; tail -5 $O/pytest_div.log

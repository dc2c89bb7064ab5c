# round 6, call 32: the distortion columns of the up field reuse t (n.uv) and r2 (n.p) + 2 t (n.uv) (GCLM_REUSE_TNUV = 1: radial 553 -> 541
# VALU per 8 px, simple_divisional 642 -> 638): same-allocation A/B, agreement of the walks, the -m gpu suite on the default and the forced
# row-pair build, 20 new fuzz seeds on the forced build and 10 on the default one
O=gpurun_out/r06; mkdir -p $O
V=geocalib_amd/lib/variants
timeout 900 python scripts/variant_probe.py --models radial,simple_divisional --reps 4 --allocations 2 reuse=geocalib_amd/lib/libgeocalib_hip.so noreuse=$V/noreuse.so 2>&1 | grep -v amdgpu | cut -c1-200 > $O/variant_reuse_tnuv.log; cat $O/variant_reuse_tnuv.log
timeout 600 python scripts/probes/row_pairs_probe.py 2>&1 | grep -v amdgpu.ids > $O/row_pairs_probe_reuse.log; grep "steps  1\|off-centre" $O/row_pairs_probe_reuse.log | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest_gpu_call32.log 2>&1; grep -E "^FAILED|passed|failed" $O/pytest_gpu_call32.log | tail -5
rm -f gpurun_out/r06q_fuzz_soak.txt; SOAK_TAG=r06q scripts/fuzz_soak.sh 363 372 300 > /dev/null 2>&1; grep -o "^seed [0-9]* cases 300 rc [0-9]*" gpurun_out/r06q_fuzz_soak.txt | awk '{print $2":"$6}' | paste -sd' '
export GCLM_LIB_PATH=$PWD/$V/rp1.so
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_gpu_rp1_reuse.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $O/pytest_gpu_rp1_reuse.log | cut -c1-200 | tail -12
rm -f gpurun_out/r06r_fuzz_soak.txt; SOAK_TAG=r06r scripts/fuzz_soak.sh 373 392 300 > /dev/null 2>&1; grep -o "^seed [0-9]* cases 300 rc [0-9]*" gpurun_out/r06r_fuzz_soak.txt | awk '{print $2":"$6}' | paste -sd' '
grep -h "AssertionError" gpurun_out/r06q_fuzz_soak.txt gpurun_out/r06r_fuzz_soak.txt | cut -c1-500

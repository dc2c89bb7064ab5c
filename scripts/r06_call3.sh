O=gpurun_out/r06; mkdir -p $O
python scripts/probes/div_tiny_k_probe.py 101 270 1 2>&1 | grep "cost_up\|H_kk\|J_up\[k\]\|(37, 3" | cut -c1-260
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -3
L=geocalib_amd/lib/libgeocalib_hip.so
timeout 600 python scripts/variant_probe.py --models simple_divisional --reps 3 new=$L old=geocalib_amd/lib/variants/pre_tie.so 2>&1 | grep -v amdgpu > $O/variant_div_tie.log; cat $O/variant_div_tie.log
rm -f gpurun_out/r06c_fuzz_soak.txt; SOAK_TAG=r06c scripts/fuzz_soak.sh 93 112 300 > /dev/null 2>&1; cut -c1-420 gpurun_out/r06c_fuzz_soak.txt | grep -o "^seed [0-9]* cases 300 rc [0-9]*\|undetermined {[^}]*}\|[0-9]* beyond their gate, [0-9]* beyond it on an unstable yardstick" | paste - - - | head -24

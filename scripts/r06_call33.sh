# round 6, call 33: fuzz 382/203 (simple_radial, 294x325: the scalar path, untouched by this round's kernel changes) -- the seed on the shipped
# build and on the build without the t (n.uv) reuse, then the draw step by step
O=gpurun_out/r06; mkdir -p $O
{
for v in "" geocalib_amd/lib/variants/noreuse.so; do
  [ -n "$v" ] && export GCLM_LIB_PATH=$PWD/$v
  rm -f gpurun_out/r06s_fuzz_soak.txt; SOAK_TAG=r06s scripts/fuzz_soak.sh 382 382 300 > /dev/null 2>&1
  echo "== ${v:-shipped}"; grep -o "^seed [0-9]* cases 300 rc [0-9]*" gpurun_out/r06s_fuzz_soak.txt; grep "AssertionError" gpurun_out/r06s_fuzz_soak.txt | cut -c1-700
done
unset GCLM_LIB_PATH
for img in 0 1 2 3; do echo "== fuzz_trace 382 203 image $img"; timeout 600 python scripts/fuzz_trace.py 382 203 4 $img 2>&1 | grep -v amdgpu | cut -c1-330; done
} > $O/fuzz_382_203.log 2>&1
head -60 $O/fuzz_382_203.log

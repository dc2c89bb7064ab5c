cd ${GRAFT_REPO_ROOT:-.}
export HSA_ENABLE_IPC_MODE_LEGACY=0
for extra in "" "--shared-group 16" "--comm rccl"; do
echo "=== 2 ranks on ONE GPU through RCCL: $extra"
timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --batch 256 --steps 3 --warmup 1 --cpu-sample 0 $extra 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-1500
echo "rc=$?"
done

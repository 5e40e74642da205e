cd $GRAFT_REPO_ROOT
for cfg in 0 14 2 4 13; do echo "== cfg=$cfg"; timeout 120 python scripts/gemm_bench.py --only qkv --opt gemm_cfg=$cfg --reps 60 2>&1 | grep -v "^sum\|amdgpu.ids"; done
python -m pytest tests/test_gpu_kernels.py -q -m gpu -x 2>&1 | tail -2

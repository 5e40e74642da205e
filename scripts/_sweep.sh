cd $GRAFT_REPO_ROOT
for b in 32 64; do for sk in 1 2 4 8; do for cfg in 0 1 4 13; do
echo "== batch=$b splitk=$sk cfg=$cfg"; timeout 120 python scripts/gemm_bench.py --batch $b --only gemm2 --y-bf16 --splitk $sk --opt gemm_cfg=$cfg --reps 60 2>&1 | grep -v "^sum\|amdgpu.ids" ; done; done; done

rm -f gpurun_out/r05_g.txt
for v in "tr_adamw_flags=0 0" "tr_adamw_flags=0 1" "tr_adamw_flags=1 1" "tr_adamw_flags=2 1" "tr_adamw_flags=3 1" "tr_adamw_flags=0 1"; do set -- $v; echo "== $1 overlap=$2" >> gpurun_out/r05_g.txt; MODE_HIP_OPTS=$1 MODE_OPT_OVERLAP=$2 MODE_FUSE_EXPERT_STEP=1 python bench.py --mode train --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['train_ms_per_step_blocks'], d['exposed_exchange_ms'])" >> gpurun_out/r05_g.txt; done
cat gpurun_out/r05_g.txt

cd $GRAFT_REPO_ROOT
python bench.py --mode train --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1   # first-process anomaly absorber
for i in 1 2 3; do
python bench.py --mode train --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('serial ', d['ms_per_step'], d['value'])"
MODE_OPT_OVERLAP=1 python bench.py --mode train --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap', d['ms_per_step'], d['value'])"
MODE_OPT_OVERLAP=1 MODE_ADAMW_BLOCKS=256 python bench.py --mode train --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap256', d['ms_per_step'], d['value'])"
done

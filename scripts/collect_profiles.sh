#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/prof_final (copied into profiles/ by hand): kernel stats of the three bench modes and the
# HBM-traffic / MFMA counters of the forward kernels (separate --pmc passes, never combined with trace domains).  Every profiler run is under
# `timeout`: an unsupported counter set (e.g. the TA_* / TCP_* sums) aborts rocprofv3, which then hangs in its signal handler.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_final; mkdir -p $O
if [ -z "$PMC_ONLY" ]; then
cd $R && python bench.py > $O/bench_sample.json 2> $O/bench_sample.err
python bench.py --mode train > $O/bench_train.json 2>> $O/bench_sample.err
python bench.py --mode rollout > $O/bench_rollout.json 2>> $O/bench_sample.err
python bench.py --mode agent --steps 5 --warmup 2 > $O/bench_agent.json 2>> $O/bench_sample.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sample -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o p -- python $R/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rollout -o p -- python $R/bench.py --mode rollout --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
fi
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $pmc --output-format csv -d $O/pmc$i -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
done
cd $R && python scripts/pmc_summary.py $O $O/gemm_pmc.json > $O/pmc_summary.txt 2>&1
rm -f $O/*/p_kernel_trace.csv $O/*/p_agent_info.csv $O/pmc*/p_counter_collection.csv
ls -R $O | head -40

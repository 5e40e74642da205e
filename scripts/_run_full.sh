python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r05_tests_b.txt
bash scripts/collect_profiles.sh > gpurun_out/r05_collect.log 2>&1
bash scripts/step_kernel_profile.sh train > /dev/null 2>&1
tail -3 gpurun_out/r05_tests_b.txt; tail -5 gpurun_out/r05_collect.log

# GPU box: the round's final check -- full GPU suite, smoke, profile collection (primary + secondary)
cd "${GRAFT_REPO_ROOT:-.}"
python -m pytest tests -m gpu -q > gpurun_out/r03_final_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_final_tests.log; tail -4 gpurun_out/r03_final_tests.log
python3 -c "import __graft_entry__ as g; g.smoke()"
bash tools/collect_profiles.sh r03 > gpurun_out/r03_collect_final.log 2>&1; tail -2 gpurun_out/r03_collect_final.log
bash tools/collect_pmc_secondary.sh r03 > gpurun_out/r03_pmc2_final.log 2>&1; tail -8 gpurun_out/r03_pmc2_final.log

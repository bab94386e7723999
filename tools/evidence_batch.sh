# round-4 evidence batch (GPU box): micro-benchmark, per-call times, whole-map hashes, kernel traces and PMC passes
R=$GRAFT_REPO_ROOT; cd $R
bash tools/ubench/run.sh r4_valu_rate.txt > /dev/null 2>&1
python3 tools/time_calls.py > gpurun_out/r4_time_calls.txt 2>&1
python3 tools/full_compare.py config2 config3 config5 --sha-only > gpurun_out/r4_sha.txt 2>&1
bash tools/kstats.sh r4_ks_c5 --workload config5 --steps 5 --pairs-per-step 4 > gpurun_out/r4_ks_c5.txt 2>&1
bash tools/kstats.sh r4_ks_c4 --workload config4 --steps 3 > gpurun_out/r4_ks_c4.txt 2>&1
bash tools/kstats.sh r4_ks_c3 --workload config3 --steps 10 --pairs-per-step 4 --in-flight 1 > gpurun_out/r4_ks_c3.txt 2>&1
bash tools/kstats.sh r4_ks_c2 --workload config2 --steps 20 --pairs-per-step 16 --in-flight 1 > gpurun_out/r4_ks_c2.txt 2>&1
bash tools/pmc.sh pmc_r4_c2 --workload config2 > gpurun_out/pmc_r4_c2.txt 2>&1
bash tools/pmc.sh pmc_r4_c3 --workload config3 --pairs-per-step 4 > gpurun_out/pmc_r4_c3.txt 2>&1
bash tools/pmc.sh pmc_r4_c5 --workload config5 --pairs-per-step 2 > gpurun_out/pmc_r4_c5.txt 2>&1
bash tools/pmc.sh pmc_r4_c4 --workload config4 > gpurun_out/pmc_r4_c4.txt 2>&1
tail -3 gpurun_out/r4_time_calls.txt; cat gpurun_out/r4_sha.txt; tail -4 gpurun_out/r4_ks_c5.txt | cut -c1-200

python -m pytest tests/test_gpu_mv_chain.py -x -q 2>&1 | tail -3
for k in 4 8 16; do echo "== K $k"; tools/prof_trace.sh r3_k_$k -- python3 /root/repo/tools/bench_configs.py c3scan:$k 2>&1 | grep "k_mvc" | cut -c1-40,100-200; grep -o '"ms_per_sweep": [0-9.]*' gpurun_out/r3_k_$k/run.log; done

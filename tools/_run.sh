mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -q -x 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --full-profile > gpurun_out/bench_c2_$i.json 2> gpurun_out/bench_c2_$i.err; cut -c100-200 gpurun_out/bench_c2_$i.json; echo; done
grep -E "ms/step" gpurun_out/bench_c2_2.err | grep "dec\." | head -12
timeout 300 python bench.py --batch 32 --steps 10 --no-cpu-baseline --full-profile > gpurun_out/bench_b32.json 2> gpurun_out/bench_b32.err; cut -c100-200 gpurun_out/bench_b32.json; echo
grep -E "ms/step" gpurun_out/bench_b32.err | head -8

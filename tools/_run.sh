mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -q -x 2>&1 | tail -2
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_c2b.json 2> gpurun_out/bench_c2b.err; echo c2 rc=$?
timeout 300 python bench.py --config 3 --steps 10 --no-cpu-baseline > gpurun_out/bench_c3b.json 2> gpurun_out/bench_c3b.err; echo c3 rc=$?
timeout 600 python tools/collect_traffic.py gpurun_out/traffic_c2b.json > /dev/null 2>&1; echo traffic rc=$?
for f in gpurun_out/bench_c2b.json gpurun_out/bench_c3b.json; do cut -c100-200 $f; echo; done
python - <<'P'
import json
d=json.load(open('gpurun_out/traffic_c2b.json'))
for k,v in d['kernels'].items(): print(k, v['launches'], 'fetch %.1f MB write %.1f MB'%(v['fetch_bytes']/1e6, v['write_bytes']/1e6))
P

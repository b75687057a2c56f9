for T in 512 1024 1536 2400 4000; do
  BV2_TILE_TARGET=$T timeout 300 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('target $T', d['value'], d['ms_per_step'], [(f['name'],f['launches'],round(f['ms_per_step'],3)) for f in d['roofline']['families']])"
done

for wg in 1 2; do for seg in 8 16 32; do
  python bench.py --no-extras --no-cpu-baseline --raycast-seg $seg --raycast-wg $wg 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wg=$wg seg=$seg', round(d['ms_per_step'],4), round(d['value']/1e6,1), d['kernel_ms'], d['config']['raycast_worklist'])"
done; done

for v in 1 2; do NHIP_TUNABLES=1 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-resid --cpu-seconds 0 --no-drop-in 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['kernels_ms_per_step']; print({x:round(k[x],3) for x in ('grid_build','of_which_grid_clear','csm_match')}, round(d['value']))"; done

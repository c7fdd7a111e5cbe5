for f in rope_s3d_amd/csrc/librope_hip_var_*.so; do for wl in cfg1 cfg5; do echo "== $f $wl"; ROPE_HIP_LIB=$PWD/$f timeout -k 10 120 python bench.py --workload $wl --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    try: d = json.loads(line)
    except Exception: continue
    r = d['roofline']; print('poses/s %.0f  score %.3f layer %.3f ms' % (d['value'], r['score_launch_ms'], r['layer_launch_ms']))
"; done; done

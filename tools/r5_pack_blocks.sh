for rep in 1 2; do for pb in 4096 2048 1024 512 8192; do
  E2S_DEV=1 E2S_PACK_BLOCKS=$pb E2S_BENCH_SECOND_LEG=0 timeout -k 10 300 python3 bench.py --workload c3 --steps 20 --warmup 3 --cpu-sample 0 --second-workload none 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('pack blocks $pb rep $rep: step %.4f (no events %.4f) scan %.4f' % (d['ms_per_step'], d['ms_per_step_without_timing_events'], d['roofline']['launch_ms']))"
done; done

for n in 262144 1048576 4194304 16777216 67108864; do
  timeout -k 10 300 python bench.py --config smc32 --no-cpu-baseline --no-whole-run --no-pattern --no-other-configs --particles-per-gpu $n 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
r=d['roofline']; print('smc32 N',d['config']['particles_total'],'value %.4g'%d['value'],'ms/step %.4f'%d['ms_per_step'],'frac %.3f'%r['frac'],'kernel upd/s %.4g'%r['kernel_updates_per_s'])"
done
for n in 65536 1048576 8388608 33554432; do
  timeout -k 10 300 python bench.py --config mc1d --no-cpu-baseline --no-whole-run --no-other-configs --particles-per-gpu $n 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
r=d['roofline']; print('mc1d N',d['config']['particles_total'],'value %.4g'%d['value'],'ms/step %.4f'%d['ms_per_step'],'kernel upd/s %.4g'%r['kernel_updates_per_s'])"
done
for n in 1048576 8388608 67108864; do
  timeout -k 10 300 python bench.py --config evidence1d --no-cpu-baseline --no-whole-run --no-pattern --no-other-configs --particles-per-gpu $n 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
r=d['roofline']; print('evidence1d N',d['config']['particles_total'],'value %.4g'%d['value'],'ms/step %.4f'%d['ms_per_step'],'kernel upd/s %.4g'%r['kernel_updates_per_s'])"
done

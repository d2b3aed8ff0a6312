for rep in 1 2; do
for K in ${KS:-2 3 4 8}; do
  if [ $K = 2 ]; then L=abcdez.jl_amd/lib/libabcdez_hip.so; else L=tools/build_variants/div$K/libabcdez_hip.so; fi
  ABCDEZ_HIP_LIB=$PWD/$L python bench.py --config mc1d --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
b=d['config'].get('better_particle_draws',{})
print('div',$K,'ms/gen',round(d['ms_per_step'],5),'value',d['value'],'ranked',b.get('generations_with_a_rank_pass'),'err',d['whole_run']['model']['posterior_mean_err'])"
done; done

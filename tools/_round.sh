# round-4 measurement pass 1: seeds + profile rounds of the other configs
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python3 tools/logz_seeds.py 16 > gpurun_out/r04_logz_seeds.log 2>&1; tail -1 gpurun_out/r04_logz_seeds.log > gpurun_out/r04_logz_seeds.json
timeout -k 10 400 python3 tools/bayes_factor_seeds.py 16 > gpurun_out/r04_bf_seeds.log 2>&1; tail -1 gpurun_out/r04_bf_seeds.log > gpurun_out/r04_bayes_factor_seeds.json
for c in mc1d lv evidence1d; do TAG=r04 CFG=$c PMC=1 bash tools/profile_round.sh > gpurun_out/r04_profile_$c.log 2>&1; done
python3 -c "
import json
for f in ('r04_logz_seeds.json','r04_bayes_factor_seeds.json'):
    d=json.load(open('gpurun_out/'+f)); print(f, {k:v for k,v in d.items() if k!='runs'})
"

#!/bin/bash
# round 4: the whole GPU suite on the no-SLP build, then the four bench lines
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04g
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
for WL in rtweekend1 overshadowed mesh1m mesh10m; do
  timeout -k 10 400 python bench.py --workload $WL --no-cpu-baseline > $O/bench_$WL.json 2> $O/bench_$WL.err && python -c "
import json;d=json.load(open('$O/bench_$WL.json'));print('$WL',round(d['value'],1),'Msamples/s',round(d['ms_per_step'],2),'ms split',d['config']['sample_split'],d['roofline']['kernel'])"
done

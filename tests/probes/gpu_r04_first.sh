#!/bin/bash
# round 4, first GPU call: the parity suite on the new build, then the VALU issue microbenchmark with in-kernel cycle counters,
# then the bench lines of config 2 with the general spheres-only kernel and of config 4
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue tests/probes/microbench/valu_issue.hip && timeout -k 10 300 /tmp/valu_issue > $O/valu_issue.txt 2>&1
tail -16 $O/valu_issue.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_rtweekend1.json 2> $O/bench_rtweekend1.err && python -c "
import json;d=json.load(open('$O/bench_rtweekend1.json'));print('rtweekend1',d['value'],d['ms_per_step'],d['config']['sample_split'])"
timeout -k 10 300 python bench.py --workload mesh1m --steps 3 --warmup 1 --no-cpu-baseline --no-walk-stats > $O/bench_mesh1m.json 2> $O/bench_mesh1m.err && python -c "
import json;d=json.load(open('$O/bench_mesh1m.json'));print('mesh1m',d['value'],d['ms_per_step'],d['config']['sample_split'])"

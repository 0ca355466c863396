#!/bin/bash
# round 4 evidence, part B: the bench lines of the four workloads on the final build (default flags = what the driver's tiers do
# not pass), the headline with the driver's flags, the general-kernel probe, and the multi-rank / multi-device rehearsals on one GPU
set -o pipefail
TAG=${TAG:-r04z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}
mkdir -p $O
cd $R
for WL in rtweekend1 overshadowed spheres500 mesh1m mesh10m; do
  timeout -k 10 500 python bench.py --workload $WL > $O/bench_$WL.json 2> $O/bench_$WL.err && python -c "
import json;d=json.load(open('$O/bench_$WL.json'));r=d['roofline'];print('$WL',round(d['value'],1),'Msamples/s',round(d['ms_per_step'],2),'ms split',d['config']['sample_split'],'frac',r.get('frac'),'cpu',round(d['cpu_baseline']['value'],2))"
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_rtweekend1_driver_flags.json 2> $O/bench_rtweekend1_driver_flags.err && python -c "
import json;d=json.load(open('$O/bench_rtweekend1_driver_flags.json'));print('driver flags',round(d['value'],1),d['config'].get('value_general_kernel'))"
timeout -k 10 300 python tests/probes/gpu_r04_general.py > $O/general.txt 2> $O/general.err; cat $O/general.txt
RT_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-walk-stats > $O/bench_rehearsal_2ranks_one_gpu.json 2> $O/rehearsal2.err; python -c "
import json;d=json.load(open('$O/bench_rehearsal_2ranks_one_gpu.json'));print('rehearsal 2 ranks',d['n_gpus'],d['config']['sample_split'],d['config']['rays_shot_per_step'])"
RT_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 4 --steps 3 --warmup 1 --no-cpu-baseline --no-walk-stats > $O/bench_rehearsal_4ranks_one_gpu.json 2> $O/rehearsal4.err; python -c "
import json;d=json.load(open('$O/bench_rehearsal_4ranks_one_gpu.json'));print('rehearsal 4 ranks',d['n_gpus'],d['config']['sample_split'],d['config']['rays_shot_per_step'])"
timeout -k 10 300 python bench.py --abi-devices 0,0 --steps 3 --warmup 1 --no-cpu-baseline --no-walk-stats > $O/bench_abi_devices_0_0.json 2> $O/abi00.err; python -c "
import json;d=json.load(open('$O/bench_abi_devices_0_0.json'));print('abi devices 0,0',round(d['value'],1),d['config']['sample_split'],d['config']['rays_shot_per_step'])"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log

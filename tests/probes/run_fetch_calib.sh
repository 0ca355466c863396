#!/bin/bash
# tests/probes/run_fetch_calib.sh: fabric-counter calibration (see microbench/fetch_calib.hip) -> gpurun_out/fetch_calib/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/fetch_calib
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/fetch_calib $R/tests/probes/microbench/fetch_calib.hip || exit 1
for CASE in "1024 64 0" "1024 128 0" "128 64 0" "2048 64 1"; do
  set -- $CASE
  NAME="set$1_rec$2_stream$3"
  i=0
  for SET in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/${NAME}_$i -- /tmp/fetch_calib $1 $2 $3 > $OUT/${NAME}_$i.out 2> $OUT/${NAME}_$i.err || echo "failed $NAME $SET" >> $OUT/failed.txt
  done
done
python3 - <<'P' > $OUT/summary.txt
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/fetch_calib"
res = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/*/*/*counter_collection.csv")):
    case = f.split("/")[-3].rsplit("_", 1)[0]
    for r in csv.DictReader(open(f)):
        if "chase" in r["Kernel_Name"] or "stream_read" in r["Kernel_Name"]:
            res[case][r["Counter_Name"]] = res[case].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for case, c in res.items():
    txt = open(glob.glob(out + f"/{case}_1.out")[0]).read().strip()
    print(case, "|", txt)
    for k, v in c.items():
        print(f"   {k:28s} {v:.0f}")
    if "FETCH_SIZE" in c:
        print(f"   FETCH_SIZE bytes as reported  {c['FETCH_SIZE']*1024:.0f}")
P
cat $OUT/summary.txt

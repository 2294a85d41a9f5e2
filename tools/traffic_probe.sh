#!/bin/bash
# Developer aid (GPU box): HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) and kernel time of the fused kernel for
# several builds of the library.  usage: bash tools/traffic_probe.sh <name> [<name> ...]   (motionplanning_5d_m_amd/libcfs_<name>.so)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/traffic
mkdir -p $OUT
cp $ROOT/motionplanning_5d_m_amd/libcfs_hip.so $OUT/keep_hip.so
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  cp $ROOT/motionplanning_5d_m_amd/libcfs_$V.so $ROOT/motionplanning_5d_m_amd/libcfs_hip.so 2>/dev/null || cp $OUT/keep_hip.so $ROOT/motionplanning_5d_m_amd/libcfs_hip.so
  for MODE in PSGCFS CFS; do
    for C in FETCH_SIZE WRITE_SIZE; do
      rm -rf $OUT/p_${V}_${MODE}_$C
      rocprofv3 --pmc $C --output-format csv -d $OUT/p_${V}_${MODE}_$C -- python3 $ROOT/bench.py --mode $MODE --steps 5 --warmup 2 --blocks 1 --streams 1 --no-cpu-baseline --no-other-mode > $OUT/p_${V}_${MODE}_$C.log 2>&1
    done
  done
  echo "$V done"
done
cp $OUT/keep_hip.so $ROOT/motionplanning_5d_m_amd/libcfs_hip.so
python3 - "$@" <<'PY'
import csv, glob, os, sys
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "traffic")
for v in sys.argv[1:]:
    for mode in ("PSGCFS", "CFS"):
        tot = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            fs = glob.glob(os.path.join(out, f"p_{v}_{mode}_{c}", "**", "*counter_collection.csv"), recursive=True)
            acc = {}
            for f in fs:
                for r in csv.DictReader(open(f)):
                    if "cfs_solve_fused_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                        acc[r["Dispatch_Id"]] = acc.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
            tot[c] = sum(acc.values()) / max(len(acc), 1)
        print(f"{v:8s} {mode:7s} fetch {tot['FETCH_SIZE']/1024:8.1f} MB (x2 = {2*tot['FETCH_SIZE']/1024:8.1f})  write {tot['WRITE_SIZE']/1024:8.1f} MB  total corrected {(2*tot['FETCH_SIZE']+tot['WRITE_SIZE'])/1024:8.1f} MB")
PY

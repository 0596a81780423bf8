#!/bin/bash
# NOTE: the OI_* A/B and ablation switches only exist in an ablation build of the library:
#   OI_EXTRA_HIPCC_FLAGS=-DOI_ABLATION python -m openintel_amd.build --force
# (the product build ignores them; rebuild without the flag afterwards).
# Effective shader clock of the cosine kernel per ablation mode: GRBM_GUI_ACTIVE / 8 XCDs / wall time
# (MI355X_MICROARCH.md, DVFS give-back).  Usage: tools/clock_probe.sh "0 5 12 4"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in $1; do
  export OI_KS_DEBUG=$m
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/clk_$m -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/clk_$m.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/clk_$m/*/*counter_collection.csv")
rows=list(csv.DictReader(open(f[0])))
ks=[r for r in rows if 'ksplit' in r['Kernel_Name'] and r['Counter_Name']=='GRBM_GUI_ACTIVE']
big=[r for r in ks if int(r['End_Timestamp'])-int(r['Start_Timestamp'])>500000]
if big:
    clk=[float(r['Counter_Value'])/8/((int(r['End_Timestamp'])-int(r['Start_Timestamp']))) for r in big]
    dur=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in big]
    print("dbg=$m launches=%d avg_dur_us=%.1f eff_clock_GHz=%.3f"%(len(big), sum(dur)/len(dur), sum(clk)/len(clk)))
else:
    print("dbg=$m no rows", len(rows), rows[0].keys() if rows else None)
PY
done

#!/bin/bash
# What binds text_parse_kernel (30 us per 16 MiB block): issue / LDS / wave counters of the stage's kernels over two C3 files.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HC_WORKLOAD_CACHE=/tmp/hcw
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/pp -- python3 $R/tools/stage_profile.py --workload c3 --reps 2 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pp/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if any(x in k for x in ("text_parse","text_lines","kept_scatter","flush_text_rows","score_kernel_coop")):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in agg.items():
    m={n: sum(v)/len(v) for n,v in c.items()}
    cyc=m.get("GRBM_GUI_ACTIVE",0)/8
    if not cyc: continue
    print(k[:60], "launches", len(c["GRBM_GUI_ACTIVE"]), "cycles", round(cyc), "valu_busy", round(m["SQ_ACTIVE_INST_VALU"]*4/(1024*cyc),3), "lds_inst_busy", round(m["SQ_ACTIVE_INST_LDS"]*4/(1024*cyc),3),
          "waves/CU", round(m["SQ_WAVE_CYCLES"]*4/(256*cyc),2), "VALU insts", round(m["SQ_INSTS_VALU"]), "LDS insts", round(m["SQ_INSTS_LDS"]), "SALU", round(m["SQ_INSTS_SALU"]), "wait_inst share", round(m["SQ_WAIT_INST_ANY"]/max(m["SQ_WAVE_CYCLES"],1),3))
PY

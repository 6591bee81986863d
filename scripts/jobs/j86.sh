#!/bin/bash
# round 3: LDS / wait-state counters of conv_f16x2_kernel (is the LDS pipe or its bank conflicts what stretches the stagers?);
# final-sample validation at 128x2048 and for the 32-step DDIM
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j86; mkdir -p $O
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_[A-Z0-9_]*LDS[A-Z0-9_]*|SQ_WAIT[A-Z0-9_]*|SQ_INSTS_[A-Z0-9_]*|SQ_ACTIVE_INST[A-Z0-9_]*|SQ_INST_CYCLES[A-Z0-9_]*|SQ_BUSY[A-Z0-9_]*|SQ_WAVE_CYCLES|SQ_LEVEL_WAVES|TCP_[A-Z0-9_]*STALL[A-Z0-9_]*)\b" | sort -u | tr '\n' ' ' > $O/counters.txt; cut -c1-1500 $O/counters.txt; echo
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --steps 2 --warmup 1 --prewarm-s 0.1"
p() { rocprofv3 --pmc $2 --output-format csv -d $O -o $1 -- python $R/bench.py $A > $O/$1.json 2> $O/$1.err; tail -1 $O/$1.err | cut -c1-200; }
p lds1 "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS GRBM_GUI_ACTIVE"
p lds2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
p wait1 "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
p act1 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$O/*_counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_f16x2_kernel" not in k or "pack" in k: continue
        k = "GN-input" if "<2, 2>" in k or "ILi2ELi2" in k else "plain-input"
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f.split("/")[-1][:6], k, {c: "%.4g" % (sum(x) / len(x)) for c, x in v.items()}, "n", len(next(iter(v.values()))))
PY
rm -f $O/*_counter_collection.csv $O/*agent_info.csv
cd $R
{ RES=128x2048 STEPS=32 timeout 900 python scripts/validate_256.py; MODE=ddim STEPS=32 timeout 600 python scripts/validate_256.py; } 2>&1 | grep -v amdgpu.ids | tee $O/validate_more.log

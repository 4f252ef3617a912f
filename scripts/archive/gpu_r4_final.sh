#!/bin/bash
# One gpurun call: everything profiles/r04_* is made of.  usage: gpu_r3_final.sh [what...]   (bench stats pmc; default all)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
WHAT="${@:-bench stats pmc}"
if [[ $WHAT == *bench* ]]; then
  timeout 900 python $R/bench.py --steps 20 --warmup 5 > $O/cfg2_bench.json 2> $O/cfg2_bench.err; echo "bench exit=$?"
  timeout 600 python $R/bench.py --protocol core --cpu-iters 0 --others 0 > $O/cfg2_pcore_bench.json 2>> $O/cfg2_bench.err
  python - <<PY
import json
d = json.load(open("$O/cfg2_bench.json"))
print("cfg2 it/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 4), {k: (round(v["ms_per_launch"], 4) if v["ms_per_launch"] else None) for k, v in d["kernels"].items()})
print("roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), "| secondary", d["roofline_secondary"] and (d["roofline_secondary"]["kernel"], round(d["roofline_secondary"]["frac"], 3)))
for k, v in (d.get("other_configs") or {}).items():
    print(k, v.get("ms_per_step"), v.get("dominant"), v.get("error"))
print("cpu", d["cpu_baseline"] and (d["cpu_baseline"]["value"], d["cpu_baseline"]["threads1"]["value"]), "mse_check", d["mse_check"])
PY
fi
if [[ $WHAT == *stats* ]]; then
  for spec in "cfg2 2 f32 20" "cfg3 3 f32 6" "cfg5 5 f32 8" "f64_cfg2 2 f64 8" "f64_cfg3 3 f64 3" "f64_cfg5 5 f64 4"; do
    set -- $spec
    $R/scripts/gpu_r4_prof.sh $1 $2 $3 $4 | head -9 | cut -c1-150
  done
fi
if [[ $WHAT == *pmc* ]]; then
  run() { # tag name counters cfg prec
    (cd /tmp && timeout 900 rocprofv3 --pmc $3 --output-format csv -d $O/pmc_$2 -o p -- python $R/bench.py --config $4 --precision $5 --steps 4 --warmup 1 --cpu-iters 0 --repeats 1 --others 0 > /dev/null 2> $O/$1_pmc_$2.err; echo "pmc $1 $2 exit=$?")
    f=$(find $O/pmc_$2 -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python $R/scripts/pmc_summary.py "$(dirname "$f")" | grep -E "sweep_scd_q|xprod16|kl_tile|kl_reg64|na_gram|colsolve|errors|xprod_tn|factor16|wh_store" >> $O/$1_pmc_summary.txt
    rm -rf $O/pmc_$2
  }
  for spec in "cfg2 2 f32" "cfg3 3 f32" "cfg5 5 f32" "f64_cfg2 2 f64"; do
    set -- $spec
    : > $O/$1_pmc_summary.txt
    run $1 fetch "FETCH_SIZE" $2 $3
    run $1 write "WRITE_SIZE" $2 $3
    run $1 sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" $2 $3
    run $1 mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE" $2 $3
    python - <<PY
import re
fe = wr = None
for line in open("$O/$1_pmc_summary.txt"):
    pass
PY
    grep -E "FETCH_SIZE|WRITE_SIZE|MFMA_BUSY|GRBM_GUI" $O/$1_pmc_summary.txt | head -24
  done
fi

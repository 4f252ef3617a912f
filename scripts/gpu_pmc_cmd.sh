#!/bin/bash
# PMC passes (separate runs, no tracing) over an arbitrary command.  usage: gpu_pmc_cmd.sh <out-file> <kernel regex> <command...>
OUT=$1; KRE=$2; shift 2
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
: > $OUT
run() { # name, counters
  (cd /tmp && timeout 900 rocprofv3 --pmc $2 --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$1 -o p -- "${CMD[@]}" > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc/$1.err; echo "pmc $1 exit=$?")
  f=$(find gpurun_out/pmc/$1 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scripts/pmc_summary.py "$(dirname "$f")" | grep -E "$KRE" >> $OUT
  rm -rf gpurun_out/pmc/$1
}
CMD=("$@")
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
run sq2 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES"
run mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_MEM_VIOLATIONS"
run grbm "GRBM_GUI_ACTIVE GRBM_COUNT"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
run tcp "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"
cat $OUT

#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel trace + separate PMC passes of one bench.py command.
#   tools/profile_gpu.sh <tag> [bench.py args...]
# Results land in gpurun_out/prof_<tag>/{kt,sq,tcc,fetch,write,tcp}; tools/pmc_summarize.py condenses them.
set -o pipefail
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
args=("$@")
run() {  # name, rocprof options...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 "$@" -d "$out/$name" -o "$name" --output-format csv -- python3 bench.py "${args[@]}" --profile > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$out/$name.log"; return 1; }
  echo "pass $name ok"
}
run kt --kernel-trace --stats &&
run sq --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY &&
run sq2 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT &&
run tcc --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum &&
run tcc2 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum &&
run tcc3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RD_UNCACHED_32B_sum &&
run fetch --pmc FETCH_SIZE &&
run write --pmc WRITE_SIZE &&
run tcp --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE

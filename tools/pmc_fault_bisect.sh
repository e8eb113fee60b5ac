#!/bin/bash
# Run ON THE GPU BOX: ONE side leg of bench.py under the counter pass that aborted in round 1
# (rocprofv3 --pmc SQ_WAVES ...), to tell which leg faults.   tools/pmc_fault_bisect.sh <loop_eager|graph_step|loop_capture>
leg=$1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY \
  -d /tmp/bisect_$leg -o $leg --output-format csv -- python3 bench.py --profile --debug-leg $leg > gpurun_out/bisect_$leg.log 2>&1
rc=$?
echo "leg $leg: exit status $rc" | tee -a gpurun_out/bisect_$leg.log
grep -n "SIGSEGV\|Segmentation\|Aborted\|core dumped\|HSA_STATUS\|Memory access fault" gpurun_out/bisect_$leg.log | head -5
exit 0

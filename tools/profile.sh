#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + PMC passes of bench.py, outputs under
# gpurun_out/prof_<tag>/.  Usage: tools/profile.sh <tag> [bench args...]
set -u
tag=${1:-r1}; shift || true
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 5 --warmup 2 --batches 1 --no-cpu-baseline --no-target --no-single-frame $*"
cd /tmp
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $BENCH > "$out/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$out/trace.log"; exit 1; }
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_INT32" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
           "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pmc --output-format csv -d "$out/pmc$i" -- $BENCH > "$out/pmc$i.log" 2>&1 || { echo "pmc pass $i failed: $pmc"; tail -3 "$out/pmc$i.log"; }
done
find "$out" -name "*.csv" | head -40

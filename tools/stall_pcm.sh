#!/usr/bin/env bash
# The SQ stall split (tools/stall_table.sh's three counter groups) of BOTH forms -- float32 and 16-bit PCM -- of
# both streaming kernels: the bench step with its int16 leg under rocprofv3 --pmc, counters only, one pass per group.
#   bash tools/stall_pcm.sh   -> gpurun_out/round6_stall/stall_pcm.md
export TMPDIR=/tmp
O=gpurun_out/round6_stall; mkdir -p $O
BENCH="python bench.py --steps 3 --warmup 1 --cpu-sample 0 --pmc 0 --full-batch 0 --e2e-utts 0"
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
G2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_LDS"
G3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"
i=0
for G in "$G1" "$G2" "$G3"; do i=$((i+1)); timeout 600 rocprofv3 --pmc $G --output-format csv -d $O/default/g$i -- $BENCH > $O/g$i.log 2>&1 || echo "pass $i failed"; done
SETK_STALL_KERNELS="stft_covar_kernel<8, false, true;beamform_istft_mc_kernel<8, true;stft_covar_kernel<8, false, false;beamform_istft_mc_kernel<8, false" python tools/stall_table.py $O | tee $O/stall_pcm.md

#!/bin/bash
# round 5, third GPU call: elementwise traffic-mix ceiling, gate-gradient statistics of G7 / G8, attention-backward SQ counters
O=gpurun_out
./scratch/ubench/ew_mix > $O/r05_ubench_ew_mix.txt 2>&1; cat $O/r05_ubench_ew_mix.txt
python -m pytest tests/test_model_gpu.py -m gpu -q -x -s -k "g7_blocks or g8_unet_loss" > $O/r05_g78.log 2>&1; echo rc=$? >> $O/r05_g78.log
grep -E "gate gradient norms|scalar gradient norms|passed|failed|rc=" $O/r05_g78.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_attn_$tag -o p -- python3 scratch/attn_bench.py 6 > $O/r05_pmc_attn_$tag.log 2>&1
done
python scratch/pmc_sq.py $O/r05_pmc_attn_bwd_sq.txt $O/pmc_attn_* --only attn_ > /dev/null 2>&1
rm -rf $O/pmc_attn_*
cat $O/r05_pmc_attn_bwd_sq.txt | head -80

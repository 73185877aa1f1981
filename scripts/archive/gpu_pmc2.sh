OUT=gpurun_out/${1:-pmc2}; mkdir -p $OUT; export TMPDIR=/tmp
for w in f w; do
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pmcA_$w -o pmc -- python $GRAFT_REPO_ROOT/scripts/conv_one.py $w 11 128 > $GRAFT_REPO_ROOT/$OUT/pmcA_$w.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmcB_$w -o pmc -- python $GRAFT_REPO_ROOT/scripts/conv_one.py $w 11 128 > $GRAFT_REPO_ROOT/$OUT/pmcB_$w.log 2>&1)
done
python scripts/pmc_summary.py $OUT/pmc_conv.json /tmp/pmcA_f /tmp/pmcB_f /tmp/pmcA_w /tmp/pmcB_w
tail -3 $OUT/pmcA_f.log | cut -c1-200

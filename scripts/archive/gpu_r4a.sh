# round 4, call A: SQ counters on conv fwd / dgrad / wgrad (k=5, 128->128, B=256, T=128) + baseline bench lines
OUT=gpurun_out/r4a; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/$OUT/counters_list.txt 2>&1)
for w in f d w; do
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pmcA_$w -o pmc -- python $GRAFT_REPO_ROOT/scripts/conv_one.py $w 11 128 > $GRAFT_REPO_ROOT/$OUT/pmcA_$w.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmcB_$w -o pmc -- python $GRAFT_REPO_ROOT/scripts/conv_one.py $w 11 128 > $GRAFT_REPO_ROOT/$OUT/pmcB_$w.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pmcC_$w -o pmc -- python $GRAFT_REPO_ROOT/scripts/conv_one.py $w 11 128 > $GRAFT_REPO_ROOT/$OUT/pmcC_$w.log 2>&1)
done
python scripts/pmc_summary.py $OUT/sq_conv.json /tmp/pmcA_f /tmp/pmcB_f /tmp/pmcC_f /tmp/pmcA_d /tmp/pmcB_d /tmp/pmcC_d /tmp/pmcA_w /tmp/pmcB_w /tmp/pmcC_w
tail -3 $OUT/pmcC_w.log | cut -c1-300
python bench.py --steps 20 --warmup 5 > $OUT/bench_f32.json 2> $OUT/bench_f32.err; tail -c 600 $OUT/bench_f32.json
python bench.py --steps 20 --warmup 5 --dtype bf16 > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err; tail -c 400 $OUT/bench_bf16.json

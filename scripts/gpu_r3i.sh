#!/bin/bash
# bf16 pair storage: first GPU contact -- op-level parity, graded-shape parity, bench lines (operand-rounding bf16 vs bf16 storage)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r3i
mkdir -p $O
timeout 900 python -m pytest tests/test_bf16_pairs.py -x -q -m gpu 2>&1 | tail -15 > $O/pairs_tests.log
timeout 900 python -m pytest tests/test_graded_configs.py -x -q -m gpu -k "bf16_storage" -s 2>&1 | tail -15 > $O/graded_bf16s.log
timeout 300 python bench.py --dtype bf16s --steps 30 --warmup 10 > $O/bench_bf16s.json 2> $O/bench_bf16s.err
timeout 300 python bench.py --dtype bf16 --steps 30 --warmup 10 --no-profile > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout 300 python bench.py --dtype bf16s --steps 30 --warmup 10 --batch 64 --frames 1024 --no-profile > $O/bench_bf16s_t1024.json 2> $O/bench_bf16s_t1024.err
tail -3 $O/pairs_tests.log; tail -8 $O/graded_bf16s.log; head -c 600 $O/bench_bf16s.json; echo; head -c 300 $O/bench_bf16.json; echo; tail -3 $O/bench_bf16s.err

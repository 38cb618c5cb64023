#!/bin/bash
# Everything the round's profiles/ directory is built from, in TWO gpurun calls (GPU box, repo root):
#   tools/final_round.sh <tag> counters   hardware-counter passes (solver counters, FETCH / WRITE traffic, per-kernel instruction counts);
#                                         then, in the build container:  python tools/make_profiles.py <tag> counters   (commits the json files)
#   tools/final_round.sh <tag> bench      GPU tests, the bench lines (which read the counter files just committed), micro-benchmarks,
#                                         kernel-trace statistics;   then:  python tools/make_profiles.py <tag>
# The bench line is produced AFTER the counter files it quotes.
set -u
TAG=${1:-r06}
STAGE=${2:-bench}
OUT=gpurun_out
mkdir -p $OUT
if [ "$STAGE" = "counters" ]; then
  bash tools/profile_round.sh $TAG pmc > $OUT/${TAG}_profile_round_pmc.log 2>&1
  bash tools/prof_solver_counters.sh $TAG > $OUT/${TAG}_solver_counters.log 2>&1
  bash tools/prof_step_instructions.sh $TAG > $OUT/${TAG}_step_instructions.log 2>&1
  # the bf16x3 convolution kernels, one instance per ResNet layer shape: matrix-pipe busy cycles, waits, LDS conflicts, instruction mix
  PASSES="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU;SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU;FETCH_SIZE;WRITE_SIZE" \
    timeout 600 bash tools/prof_kernel_counters.sh ${TAG}_convx3 conv3x3_x3 python tools/run_conv_x3_only.py > $OUT/${TAG}_convx3_counters.log 2>&1
  PASSES="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU;SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU;FETCH_SIZE;WRITE_SIZE" \
    timeout 400 bash tools/prof_kernel_counters.sh ${TAG}_headx3 point_head_x3 python tools/run_head_x3_only.py > $OUT/${TAG}_headx3_counters.log 2>&1
  PASSES="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU;SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU;FETCH_SIZE;WRITE_SIZE" \
    timeout 300 bash tools/prof_kernel_counters.sh ${TAG}_stemx3 stem_x3_kernel python tools/run_stem_x3_only.py > $OUT/${TAG}_stemx3_counters.log 2>&1
  ls $OUT | grep "^${TAG}_" | tr '\n' ' '
  exit 0
fi
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -3 > $OUT/${TAG}_gputest_tail.txt
timeout 400 python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err
timeout 200 python bench.py --mode train --steps 6 --warmup 2 > $OUT/${TAG}_train_line.json 2>> $OUT/${TAG}_bench.err
timeout 100 python tools/bench_index_max.py > $OUT/${TAG}_index_max_cold.txt 2>&1
timeout 300 python tools/bench_conv_x3.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_conv_layers.txt
timeout 100 python tools/diag_conv_x3_accuracy.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_conv_x3_accuracy.txt
(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_rounding.hip -o /tmp/probe_mfma_rounding 2>/dev/null && /tmp/probe_mfma_rounding) > $OUT/${TAG}_mfma_rounding.txt 2>&1
ROUNDS=3 REPS=50 timeout 200 python tools/bench_head_x3.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_head_x3.txt
REPS=50 timeout 100 python tools/bench_stem_x3.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_stem_x3.txt
timeout 100 python tools/probe_x3_ranges.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_x3_ranges.txt
timeout 150 python tools/bench_winograd.py > $OUT/${TAG}_winograd_layers.txt 2>&1
PROF=1 PFC=2 timeout 150 python tools/bench_solver.py > $OUT/${TAG}_solver_phases.txt 2>&1
bash tools/profile_round.sh $TAG stats > $OUT/${TAG}_profile_round.log 2>&1
timeout 200 python tools/call_times.py 15 > $OUT/${TAG}_call_times.txt 2>&1
timeout 200 bash tools/sweep_streams2.sh > $OUT/${TAG}_sweep_streams.txt 2>&1
timeout 200 bash tools/probe_additivity.sh > $OUT/${TAG}_additivity.txt 2>&1
export TMPDIR=/tmp; ROOT=$(pwd); cd /tmp; rm -rf /tmp/pt
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python $ROOT/bench.py --mode train --steps 4 --warmup 2 > /tmp/pt.log 2>&1
f=$(find /tmp/pt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $ROOT/$OUT/${TAG}_train_kernel_stats.csv
cd $ROOT; cat $OUT/${TAG}_gputest_tail.txt; head -c 600 $OUT/${TAG}_bench_line.json; echo; ls $OUT | grep "^${TAG}_" | tr '\n' ' '

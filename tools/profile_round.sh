#!/bin/bash
# Profiles of one round, run ON THE GPU BOX from the repo root:  bash tools/profile_round.sh <tag>   (e.g. r01b)
# Kernel-trace statistics of bench.py for C2/C3/C4, then separate PMC passes of the attention kernel alone at C3
# (counters never share a run with the trace; FETCH_SIZE / WRITE_SIZE in their own passes -- MI355X_MICROARCH.md).
set -e -o pipefail
TAG=${1:-r01}
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in c3 c2 c4; do
  # bench.py's OWN windows (c2: 200 timed steps after 20 warm-up steps -- the steady state its JSON line quotes; c3/c4: 100 + 20)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$wl -- python3 $R/bench.py --workload $wl --no-cpu-baseline --no-fa2 > $OUT/kt_$wl.log 2>&1
  cp $(ls $OUT/kt_$wl/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_${wl}_kernel_stats.csv
  grep "^{" $OUT/kt_$wl.log | tail -1 > $OUT/${TAG}_bench_${wl}_under_tracer.json   # (the tracer prints after the bench: not the last line)
done
# the attention kernels alone over the sweep's shapes (all head_dim-64 and -128 rows, FP16 and FP8 PV): steady state
# (30 back-to-back launches per shape) and isolated launches (synchronise + 3 ms pause before each)
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_sweep -- python3 $R/tools/sweep_kernels.py 30 > $OUT/kt_sweep.log 2>&1
python3 $R/tools/trace_summary.py $OUT/kt_sweep 3 > $OUT/${TAG}_sweep_kernel_durations.md
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_sweep_iso -- python3 $R/tools/sweep_kernels.py 12 --isolated > $OUT/kt_sweep_iso.log 2>&1
python3 $R/tools/trace_summary.py $OUT/kt_sweep_iso 3 > $OUT/${TAG}_sweep_kernel_durations_isolated.md
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -- python3 $R/tools/run_attn.py c3 3 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/run_attn.py c3 3 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/pmc_write -- python3 $R/tools/run_attn.py c3 3 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_clk -- python3 $R/tools/run_attn.py c3 3 > $OUT/pmc_clk.log 2>&1 || true
python3 $R/tools/pmc_summary.py attn_i8_kernel $OUT/${TAG}_attn_c3_pmc.json $OUT/pmc1 $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_clk
# the same passes for C4 (INT8 QK^T + FP8 PV, causal, 16K)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/c4_pmc1 -- python3 $R/tools/run_attn.py c4 3 > $OUT/c4_pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c4_pmc_fetch -- python3 $R/tools/run_attn.py c4 3 > $OUT/c4_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/c4_pmc_write -- python3 $R/tools/run_attn.py c4 3 > $OUT/c4_pmc_write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/c4_pmc_clk -- python3 $R/tools/run_attn.py c4 3 > $OUT/c4_pmc_clk.log 2>&1 || true
python3 $R/tools/pmc_summary.py attn_i8_kernel $OUT/${TAG}_attn_c4_pmc.json $OUT/c4_pmc1 $OUT/c4_pmc_fetch $OUT/c4_pmc_write $OUT/c4_pmc_clk
# and for C2 (head_dim 64, 2K keys, FP16 PV): the configuration furthest below the MFMA roofline
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/c2_pmc1 -- python3 $R/tools/run_attn.py c2 5 > $OUT/c2_pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c2_pmc_fetch -- python3 $R/tools/run_attn.py c2 5 > $OUT/c2_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/c2_pmc_write -- python3 $R/tools/run_attn.py c2 5 > $OUT/c2_pmc_write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/c2_pmc_clk -- python3 $R/tools/run_attn.py c2 5 > $OUT/c2_pmc_clk.log 2>&1 || true
python3 $R/tools/pmc_summary.py attn_i8_kernel $OUT/${TAG}_attn_c2_pmc.json $OUT/c2_pmc1 $OUT/c2_pmc_fetch $OUT/c2_pmc_write $OUT/c2_pmc_clk

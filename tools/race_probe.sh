#!/bin/bash
# Evidence for the K-buffer-0 race (round-2 fix).  Needs the probe libraries built by
# tools/build_probe_libs.py under ab_libs/.  Writes gpurun_out/r02_race/*.txt.
#   delay_*  : wave 1 of every workgroup sleeps ~8 us between the prologue barrier and its K(0) fragment reads
#              (-DSAGE_EXP_DELAY_WAVE); without the new barrier its 32 rows must be wrong in EVERY launch, with it never.
#   nobar    : round-1 behaviour (no barrier between the prologue S(0) and the first K(2) copy), fused-Q path, stressed.
#   ctemp*   : the round-1 "C operand is a temporary" form of the first S MFMA, with and without the barrier.
set -e -o pipefail
P=gpurun_out/r02_race
mkdir -p $P
python tools/cmp_libs.py --shape 4,32,2048,64 --causal 1 --runs 20 ab_libs/lib_delay_nobar.so ab_libs/lib_delay_bar.so > $P/delay_d64_causal.txt 2>&1
python tools/cmp_libs.py --shape 2,16,4096,128 --causal 0 --runs 20 ab_libs/lib_delay_nobar.so ab_libs/lib_delay_bar.so > $P/delay_d128.txt 2>&1
SAGE_LIB_OVERRIDE=ab_libs/lib_nobar.so python tools/stress_determinism.py 1200 fuse 0 > $P/stress_nobar_fused.txt 2>&1
python tools/stress_determinism.py 3000 fuse 0 > $P/stress_fixed_fused_cfg0.txt 2>&1
python tools/stress_determinism.py 400 fuse > $P/stress_fixed_fused_all.txt 2>&1
python tools/stress_determinism.py 400 nofuse > $P/stress_fixed_nofuse_all.txt 2>&1
python tools/cmp_libs.py --shape 4,32,2048,64 --causal 1 --runs 100 ab_libs/lib_ctemp_nobar.so ab_libs/lib_ctemp.so > $P/ctemp_d64_causal.txt 2>&1
python tools/cmp_libs.py --shape 4,32,2048,64 --causal 1 --pv fp8 --runs 100 ab_libs/lib_ctemp_nobar.so ab_libs/lib_ctemp.so > $P/ctemp_d64_causal_fp8.txt 2>&1
SAGE_LIB_OVERRIDE=ab_libs/lib_ctemp.so python tools/stress_determinism.py 1000 fuse 0 > $P/stress_ctemp_fused.txt 2>&1
tail -n 3 $P/*.txt

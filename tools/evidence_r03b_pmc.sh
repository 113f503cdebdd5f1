#!/bin/bash
# Hardware counters of the ZI kernels of csrc/dense_zi.hip and of round 2's (dense_f32.hip) at 100k x 20k, K = 100 / 50,
# and of the configs[1] pass kernels (each --pmc pass on its own, counters only with --kernel-trace).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b
mkdir -p $O
F="k_zi_row|k_zi_col|k_dropout_sweep|k_dt_times_factor"
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
C2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"
tools/pmc_cmd.sh "$C1" r03b_zi_sq1 "$F" tools/perf_zi_per_k.py 100000 20000 50 100 > $O/zi_sq1.txt 2>&1
tools/pmc_cmd.sh "$C2" r03b_zi_sq2 "$F" tools/perf_zi_per_k.py 100000 20000 50 100 > $O/zi_sq2.txt 2>&1
cp gpurun_out/pmc/r03b_zi_*.json $O/
cat $O/zi_sq1.txt $O/zi_sq2.txt | cut -c1-600

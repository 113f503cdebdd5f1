#!/bin/bash
# Evidence run of round 4, part 2: hardware counters of the pass kernels of the SECONDARY workloads of the bench line
# (configs[2] ZI-pCMF, configs[4] sparse pCMF): each --pmc pass on its own, counters only with --kernel-trace.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
F="k_row_pass|k_col_pass|k_fixup|k_zi_|k_dropout|k_dt_times|k_zi_images|k_split"
for w in c3_zi c5_sparse; do
  tools/pmc_cmd.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" r04_${w}_sq1 "$F" bench.py --workload $w --steps 3 --warmup 1 --no-cpu > $O/sq1_$w.txt 2>&1
  tools/pmc_cmd.sh "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" r04_${w}_sq2 "$F" bench.py --workload $w --steps 3 --warmup 1 --no-cpu > $O/sq2_$w.txt 2>&1
  tools/pmc_cmd.sh "FETCH_SIZE" r04_${w}_fetch "$F" bench.py --workload $w --steps 3 --warmup 1 --no-cpu > $O/fetch_$w.txt 2>&1
  tools/pmc_cmd.sh "WRITE_SIZE" r04_${w}_write "$F" bench.py --workload $w --steps 3 --warmup 1 --no-cpu > $O/write_$w.txt 2>&1
done
cp gpurun_out/pmc/r04_*.json $O/
cat $O/sq1_c3_zi.txt $O/fetch_c3_zi.txt $O/write_c3_zi.txt $O/sq1_c5_sparse.txt $O/fetch_c5_sparse.txt $O/write_c5_sparse.txt | cut -c1-420

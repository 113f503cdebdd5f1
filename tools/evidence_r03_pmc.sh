#!/bin/bash
# Evidence run of round 3, part 2: hardware counters of the pass kernels at full C4 with the hybrid layout (each --pmc pass
# on its own, counters only with --kernel-trace) and the cycle stamps of the dense row kernel.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
F="k_row_pass|k_col_pass|k_fixup|k_dn_"
tools/pmc_cmd.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" r03_c4_sq1 "$F" bench.py --steps 2 --warmup 1 --no-cpu > $O/sq1.txt 2>&1
tools/pmc_cmd.sh "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" r03_c4_sq2 "$F" bench.py --steps 2 --warmup 1 --no-cpu > $O/sq2.txt 2>&1
tools/pmc_cmd.sh "FETCH_SIZE" r03_c4_fetch "$F" bench.py --steps 2 --warmup 1 --no-cpu > $O/fetch.txt 2>&1
tools/pmc_cmd.sh "WRITE_SIZE" r03_c4_write "$F" bench.py --steps 2 --warmup 1 --no-cpu > $O/write.txt 2>&1
cp gpurun_out/pmc/r03_c4_*.json $O/
ORIANA_HIP_LIB=$GRAFT_REPO_ROOT/scratch/variants/liboriana_STAMP.so timeout 300 python tools/perf_dense_pass.py 125000 30000 100 0.1 0.2 2>&1 | grep -a "stamp blk" | sort | uniq | tail -8 > $O/dense_row_stamps.txt
cat $O/sq1.txt $O/sq2.txt $O/fetch.txt $O/write.txt | cut -c1-400; cat $O/dense_row_stamps.txt

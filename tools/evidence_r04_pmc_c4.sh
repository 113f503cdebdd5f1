#!/bin/bash
# Evidence run of round 4, part 3: hardware counters of the pass kernels at full C4 with the hybrid layout on the final build
# (each --pmc pass on its own, counters only with --kernel-trace).  Filed by tools/assemble_profiles_pmc.py.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
F="k_row_pass|k_col_pass|k_fixup|k_dn_"
tools/pmc_cmd.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" r04_c4_sq1 "$F" bench.py --steps 2 --warmup 1 --no-cpu > $O/sq1_c4.txt 2>&1
tools/pmc_cmd.sh "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" r04_c4_sq2 "$F" bench.py --steps 2 --warmup 1 --no-cpu > $O/sq2_c4.txt 2>&1
tools/pmc_cmd.sh "FETCH_SIZE" r04_c4_fetch "$F" bench.py --steps 2 --warmup 1 --no-cpu > $O/fetch_c4.txt 2>&1
tools/pmc_cmd.sh "WRITE_SIZE" r04_c4_write "$F" bench.py --steps 2 --warmup 1 --no-cpu > $O/write_c4.txt 2>&1
cp gpurun_out/pmc/r04_c4_*.json $O/
cat $O/sq1_c4.txt $O/fetch_c4.txt $O/write_c4.txt | cut -c1-420

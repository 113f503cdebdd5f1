#!/bin/bash
# usage: tools/evidence_pmc.sh <round tag>     (1 GPU; run through gpurun)
# Hardware counters of the pass kernels on the final build of a round -- configs[3] (hybrid layout) and the two secondary
# workloads of the bench line (configs[2] ZI-pCMF, configs[4] sparse pCMF).  Each --pmc pass on its own, counters only with
# --kernel-trace (MI355X_MICROARCH.md, HBM / rocprofv3).  Filed by `python tools/assemble_profiles_pmc.py <round tag>`.
RND=${1:?usage: tools/evidence_pmc.sh <round tag>}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$RND
mkdir -p $O
echo "tools/evidence_pmc.sh $RND" > $O/command_pmc.txt
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"
F="k_row_pass|k_col_pass|k_fixup|k_dn_"
tools/pmc_cmd.sh "$SQ1" ${RND}_c4_sq1 "$F" bench.py --steps 2 --warmup 1 --no-cpu --brief > $O/sq1_c4.txt 2>&1
tools/pmc_cmd.sh "$SQ2" ${RND}_c4_sq2 "$F" bench.py --steps 2 --warmup 1 --no-cpu --brief > $O/sq2_c4.txt 2>&1
tools/pmc_cmd.sh "FETCH_SIZE" ${RND}_c4_fetch "$F" bench.py --steps 2 --warmup 1 --no-cpu --brief > $O/fetch_c4.txt 2>&1
tools/pmc_cmd.sh "WRITE_SIZE" ${RND}_c4_write "$F" bench.py --steps 2 --warmup 1 --no-cpu --brief > $O/write_c4.txt 2>&1
F="k_row_pass|k_col_pass|k_fixup|k_zi_|k_dropout|k_dt_times|k_zi_images|k_split|k_logit"
for w in c3_zi c5_sparse; do
  tools/pmc_cmd.sh "$SQ1" ${RND}_${w}_sq1 "$F" bench.py --workload $w --steps 3 --warmup 1 --no-cpu --brief > $O/sq1_$w.txt 2>&1
  tools/pmc_cmd.sh "$SQ2" ${RND}_${w}_sq2 "$F" bench.py --workload $w --steps 3 --warmup 1 --no-cpu --brief > $O/sq2_$w.txt 2>&1
  tools/pmc_cmd.sh "FETCH_SIZE" ${RND}_${w}_fetch "$F" bench.py --workload $w --steps 3 --warmup 1 --no-cpu --brief > $O/fetch_$w.txt 2>&1
  tools/pmc_cmd.sh "WRITE_SIZE" ${RND}_${w}_write "$F" bench.py --workload $w --steps 3 --warmup 1 --no-cpu --brief > $O/write_$w.txt 2>&1
done
cp gpurun_out/pmc/${RND}_*.json $O/
cat $O/fetch_c4.txt $O/write_c4.txt $O/fetch_c3_zi.txt $O/write_c3_zi.txt $O/fetch_c5_sparse.txt $O/write_c5_sparse.txt | cut -c1-300

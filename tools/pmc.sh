#!/bin/bash
# PMC passes (separate runs, kernel-trace only) for the KG pipeline -> gpurun_out/pmc/<pass>/
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p $R/gpurun_out/pmc
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64" \
            "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAVES" \
            "$PMC_EXTRA"; do
  [ -z "$ctrs" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc/p$i -o p -- python $R/tools/prof_kg.py ${PROF_ARGS:-C3 8 2} > $R/gpurun_out/pmc/p$i.log 2>&1
done
python $R/tools/pmc_summary.py $R/gpurun_out/pmc | tee $R/gpurun_out/pmc/summary.txt

mkdir -p gpurun_out
python tools/attn_pmc.py 20
REPO=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $REPO/gpurun_out/apmc1 -o p1 -- python $REPO/tools/attn_pmc.py 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_WAVES --output-format csv -d $REPO/gpurun_out/apmc2 -o p2 -- python $REPO/tools/attn_pmc.py 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/apmc3 -o p3 -- python $REPO/tools/attn_pmc.py 3 > /dev/null 2>&1
cd $REPO; ls gpurun_out/apmc1 gpurun_out/apmc2 gpurun_out/apmc3 2>/dev/null | head

# dynamic vector-instruction mix per kernel (what the activation / split / addressing work consists of)
export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/pmc_v /tmp/pmc_v2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 -d /tmp/pmc_v -o v --output-format csv -- python3 $root/tools/pmc_run.py parity 2 > /dev/null 2> /tmp/pmc_v.err
python3 $root/tools/pmc_summary.py /tmp/pmc_v | grep -A9 -E "^(sdf_fwd_tph|sdf_fwd_grad_tp|sdf_bwd_tph|color_fwd_tph)"
tail -2 /tmp/pmc_v.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA -d /tmp/pmc_v2 -o v --output-format csv -- python3 $root/tools/pmc_run.py parity 2 > /dev/null 2> /tmp/pmc_v2.err
python3 $root/tools/pmc_summary.py /tmp/pmc_v2 | grep -A9 -E "^(sdf_fwd_tph|sdf_fwd_grad_tp|sdf_bwd_tph|color_fwd_tph)"
tail -2 /tmp/pmc_v2.err

# SQ_VALU_MFMA_COEXEC_CYCLES per kernel (do the vector and the matrix pipe of a SIMD work at the same time?)
export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/pmc_c
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES -d /tmp/pmc_c -o c --output-format csv -- python3 $root/tools/pmc_run.py parity 2 > /dev/null 2> /tmp/pmc_c.err
python3 $root/tools/pmc_summary.py /tmp/pmc_c | grep -A7 -E "^(sdf_fwd_tph|sdf_fwd_grad_tp|sdf_bwd_tph|color_fwd_tph|color_bwd_tph|dw_gemm)"
tail -3 /tmp/pmc_c.err

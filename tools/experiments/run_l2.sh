# L2 (TCC) hits / misses per kernel: is the weight stream of the chain kernels served from L2?
export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/pmc_l
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d /tmp/pmc_l -o l --output-format csv -- python3 $root/tools/pmc_run.py parity 2 > /dev/null 2> /tmp/pmc_l.err
python3 $root/tools/pmc_summary.py /tmp/pmc_l | grep -A5 -E "^(sdf_fwd_tph|sdf_fwd_grad_tp|sdf_bwd_tph|color_fwd_tph|color_bwd_tph|sdf_fwd_tp_kernel)"
tail -3 /tmp/pmc_l.err

for i in 1 2; do python tools/_alt/two_fwd_ddp2.py ref_first 2>&1 | grep mismatches; done
for i in 1 2; do python tools/_alt/two_fwd_ddp2.py net_first 2>&1 | grep mismatches; done
for i in 1 2; do TMF_DDP_DEBUG_SYNC=1 python tools/_alt/two_fwd_ddp2.py ref_first 2>&1 | grep mismatches; done

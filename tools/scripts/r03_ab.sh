cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "wgrad or conv3d" 2>&1 | tail -3
for i in 1 2; do
TMF_LIB=$R/tools/_alt/libtmf_prev.so python3 tools/wgrad_ab.py --fp32 --S 96 2>&1 | grep -v amdgpu.ids
python3 tools/wgrad_ab.py --fp32 --S 96 2>&1 | grep -v amdgpu.ids
done

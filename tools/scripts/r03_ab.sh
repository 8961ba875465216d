cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_kernel_resources.py -q -m gpu -x -k "bf16" 2>&1 | tail -3
for i in 1 2; do
TMF_LIB=$R/tools/_alt/libtmf_prev.so python3 tools/wgrad_ab.py 2>&1 | grep -v amdgpu.ids
python3 tools/wgrad_ab.py 2>&1 | grep -v amdgpu.ids
done

mkdir -p gpurun_out
{
echo "# baseline"; timeout 120 python tools/winox_check.py --time-only --only conv2 --rounds 2 2>&1 | grep -v amdgpu.ids
for v in 1 2 4 8 16 32 47; do echo "# X_ABL=$v"; TMF_LIB=transmf_ad_amd/libtmf_xabl$v.so timeout 120 python tools/winox_check.py --time-only --only conv2 --rounds 2 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r06_winox_ablations.txt 2>&1
cat gpurun_out/r06_winox_ablations.txt

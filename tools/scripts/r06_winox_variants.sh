# structure variants of the split Winograd kernel (compile-time switches X_PF / X_SKEW / X_RB), timing only
mkdir -p gpurun_out
{
echo "# default build (X_PF 2, X_SKEW 0, X_RB 0)"; timeout 200 python tools/winox_check.py --time-only --rounds 2 2>&1 | grep -v amdgpu.ids
for v in 100 210 201; do echo "# X_PF X_SKEW X_RB = $v"; TMF_LIB=transmf_ad_amd/libtmf_xv$v.so timeout 200 python tools/winox_check.py --time-only --rounds 2 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r06_winox_variants.txt 2>&1
grep -E "^#|^sum|fwd" gpurun_out/r06_winox_variants.txt

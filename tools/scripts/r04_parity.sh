# parity margins of the fp32 path against the golden fixtures with the Winograd form (default) and with the direct kernels
mkdir -p gpurun_out/wino
{ echo "# tools/parity_report.py --grads, conv_wino = 3 (default: Winograd forward, data and weight gradients)"; python tools/parity_report.py --grads 2>&1 | grep -v amdgpu.ids; echo; echo "# TMF_CONV_WINO=0 (direct kernels)"; TMF_CONV_WINO=0 python tools/parity_report.py --grads 2>&1 | grep -v amdgpu.ids; } > gpurun_out/wino/r04_parity_report.txt
tail -40 gpurun_out/wino/r04_parity_report.txt

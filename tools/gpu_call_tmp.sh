for layer in l4_conv2_d4 conv_last; do for v in "" _nomma _nodma _dmaonly _mmaonly; do
echo "== $layer variant '$v'"; python tools/probes/gemm_phase_stamps.py --layer $layer --variant "$v" 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once\|blocks with stamp" | tail -8
done; done

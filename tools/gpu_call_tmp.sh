for layer in conv_last l4_conv2_d4; do for t in 25 26; do for v in "" _dmaonly _mmaonly; do
echo "== $layer tile $t variant '$v'"; python tools/probes/gemm_phase_stamps.py --layer $layer --variant "$v" --wino-tile $t 2>&1 | grep "pinned\|   kernel void igemm"
done; done; done

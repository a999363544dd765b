// Cross-kernel host helpers of the C ABI (see include/semseg_hip.h).
#include "common.h"

size_t igemm_workspace_bytes(int M, int Cout, int Cin, int T);   // conv_igemm.hip
size_t wgrad_workspace_bytes(int M, int K, int C, int T);        // conv_wgrad.hip

extern "C" size_t semseg_conv2d_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                                                int dil) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || R <= 0 || S <= 0 || stride <= 0 || dil <= 0) return 0;
    const int OH = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1;
    const int OW = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
    if (OH <= 0 || OW <= 0) return 0;
    const int T = R * S;
    size_t a = igemm_workspace_bytes(N * OH * OW, K, C, T);      // forward
    size_t b = igemm_workspace_bytes(N * H * W, C, K, T);        // dgrad
    size_t c = wgrad_workspace_bytes(N * OH * OW, K, C, T);      // wgrad
    size_t m = a > b ? a : b;
    return m > c ? m : c;
}

// Per-pixel softmax head of the segmentation hot path (gfx950): log_softmax / softmax over the class
// axis (150 classes, NHWC so a pixel's classes are one contiguous 600-byte row), NLL loss with
// ignore_index, pixel accuracy, and their backward.  One 64-lane wavefront owns one pixel: the row is
// read once (3 values per lane for C=150), max and sum(exp) are wavefront shuffle reductions.
// Replaces F.log_softmax / F.softmax (reference models.py:383,482-484,492-493,584), nn.NLLLoss
// (train.py:154) and SegmentationModuleBase.pixel_acc (models.py:12-18).
#include "common.h"

constexpr int HEAD_MAX_PER_LANE = 16;   // C <= 1024

template <bool LOG>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ z, float* __restrict__ out, int P, int C) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const float* row = z + (size_t)p * C;
    float v[HEAD_MAX_PER_LANE];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < HEAD_MAX_PER_LANE; ++k) {
        const int c = lane + 64 * k;
        v[k] = (c < C) ? row[c] : -INFINITY;
        m = fmaxf(m, v[k]);
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < HEAD_MAX_PER_LANE; ++k) {
        const int c = lane + 64 * k;
        if (c < C) s += expf(v[k] - m);
    }
    s = wave_sum(s);
    float* o = out + (size_t)p * C;
    if (LOG) {
        const float lse = m + logf(s);
#pragma unroll
        for (int k = 0; k < HEAD_MAX_PER_LANE; ++k) {
            const int c = lane + 64 * k;
            if (c < C) o[c] = v[k] - lse;
        }
    } else {
        const float inv = 1.0f / s;
#pragma unroll
        for (int k = 0; k < HEAD_MAX_PER_LANE; ++k) {
            const int c = lane + 64 * k;
            if (c < C) o[c] = expf(v[k] - m) * inv;
        }
    }
}

extern "C" int semseg_log_softmax_fwd(const float* z, float* logp, int P, int C, void* stream) {
    if (!z || !logp || P <= 0 || C <= 0 || C > 64 * HEAD_MAX_PER_LANE) return SEMSEG_EINVAL;
    hipLaunchKernelGGL(softmax_fwd_kernel<true>, dim3(ceil_div(P, 4)), dim3(256), 0, (hipStream_t)stream, z, logp, P, C);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}
extern "C" int semseg_softmax_fwd(const float* z, float* prob, int P, int C, void* stream) {
    if (!z || !prob || P <= 0 || C <= 0 || C > 64 * HEAD_MAX_PER_LANE) return SEMSEG_EINVAL;
    hipLaunchKernelGGL(softmax_fwd_kernel<false>, dim3(ceil_div(P, 4)), dim3(256), 0, (hipStream_t)stream, z, prob, P, C);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// dz = dlogp - exp(logp) * sum_c dlogp
__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float* __restrict__ dlogp, const float* __restrict__ logp,
                                                              float* __restrict__ dz, int P, int C) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const float* g = dlogp + (size_t)p * C;
    const float* l = logp + (size_t)p * C;
    float gv[HEAD_MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < HEAD_MAX_PER_LANE; ++k) {
        const int c = lane + 64 * k;
        gv[k] = (c < C) ? g[c] : 0.f;
        s += gv[k];
    }
    s = wave_sum(s);
    float* o = dz + (size_t)p * C;
#pragma unroll
    for (int k = 0; k < HEAD_MAX_PER_LANE; ++k) {
        const int c = lane + 64 * k;
        if (c < C) o[c] = gv[k] - expf(l[c]) * s;
    }
}

extern "C" int semseg_log_softmax_bwd(const float* dlogp, const float* logp, float* dz, int P, int C, void* stream) {
    if (!dlogp || !logp || !dz || P <= 0 || C <= 0 || C > 64 * HEAD_MAX_PER_LANE) return SEMSEG_EINVAL;
    hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3(ceil_div(P, 4)), dim3(256), 0, (hipStream_t)stream, dlogp, logp, dz, P, C);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// pass 1: per-block partial (loss sum fp64, valid count, hit count); one wave per pixel
constexpr int NLL_BLOCKS = 256;
__global__ __launch_bounds__(256) void nll_acc_partial_kernel(const float* __restrict__ logp, const int64_t* __restrict__ label,
                                                              int ignore_index, int P, int C, double* __restrict__ partial) {
    __shared__ double sl[4];
    __shared__ int sv[4], sh[4], sn[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double loss = 0.0;
    int valid = 0, hits = 0, nonneg = 0;
    for (int p = blockIdx.x * 4 + w; p < P; p += gridDim.x * 4) {
        const float* row = logp + (size_t)p * C;
        // first-max argmax over C (torch.max(dim=1) returns the first index of the maximum)
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int c = lane; c < C; c += 64) {
            const float v = row[c];
            if (v > best || (v != v && best == best)) { best = v; bi = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) {
            const long long lab = label[p];
            // pixel_acc (models.py:12-18): valid = label >= 0 ; NLLLoss: valid = label != ignore_index
            if (lab >= 0) {
                hits += ((long long)bi == lab) ? 1 : 0;
                nonneg += 1;
            }
            if (lab != ignore_index) {
                valid += 1;
                if (lab >= 0 && lab < C) loss -= (double)row[lab];
            }
        }
    }
    if (lane == 0) { sl[w] = loss; sv[w] = valid; sh[w] = hits; sn[w] = nonneg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 4 + 0] = sl[0] + sl[1] + sl[2] + sl[3];
        partial[blockIdx.x * 4 + 1] = (double)(sv[0] + sv[1] + sv[2] + sv[3]);
        partial[blockIdx.x * 4 + 2] = (double)(sh[0] + sh[1] + sh[2] + sh[3]);
        partial[blockIdx.x * 4 + 3] = (double)(sn[0] + sn[1] + sn[2] + sn[3]);      // accuracy denominator: label >= 0
    }
}

__global__ void nll_acc_finish_kernel(const double* __restrict__ partial, int nblocks, float* __restrict__ out) {
    // single wave over the <= 256 per-block partials
    const int lane = threadIdx.x;
    double l = 0.0, v = 0.0, h = 0.0, nonneg = 0.0;
    for (int b = lane; b < nblocks; b += 64) {
        l += partial[b * 4]; v += partial[b * 4 + 1]; h += partial[b * 4 + 2]; nonneg += partial[b * 4 + 3];
    }
    l = wave_sum_d(l); v = wave_sum_d(v); h = wave_sum_d(h); nonneg = wave_sum_d(nonneg);
    if (lane == 0) {
        out[0] = (float)(l / v);                                   // mean over non-ignored (0/0 -> NaN as torch)
        out[1] = (float)h / ((float)nonneg + 1e-10f);              // models.py:17
        out[2] = (float)v;
    }
}

extern "C" int semseg_nll_acc_fwd(const float* logp, const int64_t* label, int ignore_index, int P, int C, float* out,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    if (!logp || !label || !out || P <= 0 || C <= 0) return SEMSEG_EINVAL;
    const int blocks = min(NLL_BLOCKS, ceil_div(P, 4));
    if (!workspace || workspace_bytes < (size_t)blocks * 4 * sizeof(double)) return SEMSEG_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(nll_acc_partial_kernel, dim3(blocks), dim3(256), 0, st, logp, label, ignore_index, P, C, (double*)workspace);
    SEMSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(nll_acc_finish_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, blocks, out);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

__global__ void nll_bwd_kernel(const float* __restrict__ gloss, const float* __restrict__ nll_out,
                               const int64_t* __restrict__ label, int ignore_index, float* __restrict__ dlogp, int P, int C) {
    const size_t total = (size_t)P * C;
    const float g = -gloss[0] / nll_out[2];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i / C);
        const int c = (int)(i - (size_t)p * C);
        const long long lab = label[p];
        dlogp[i] = (lab != ignore_index && lab == c) ? g : 0.f;
    }
}

extern "C" int semseg_nll_bwd(const float* gloss, const float* nll_out, const int64_t* label, int ignore_index, float* dlogp,
                              int P, int C, void* stream) {
    if (!gloss || !nll_out || !label || !dlogp || P <= 0 || C <= 0) return SEMSEG_EINVAL;
    size_t b = ceil_div_sz((size_t)P * C, 256);
    if (b > 4096) b = 4096;
    hipLaunchKernelGGL(nll_bwd_kernel, dim3((int)b), dim3(256), 0, (hipStream_t)stream, gloss, nll_out, label, ignore_index, dlogp,
                       P, C);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- SGD (train.py:117-126) ----------
constexpr int SGD_MAX_TENSORS = 48;
struct SgdBatch {
    semseg_sgd_tensor t[SGD_MAX_TENSORS];
    int n;
};

__global__ __launch_bounds__(256) void sgd_kernel(const SgdBatch b, const float* __restrict__ lr_ptr, float momentum,
                                                  float grad_scale) {
    const semseg_sgd_tensor t = b.t[blockIdx.y];
    const float lr = lr_ptr[0];
    const int64_t n = t.numel;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto upd = [&](float w, float gr, float mb, float& m_out, float& w_out) {
        const float g = fmaf(t.weight_decay, w, gr * grad_scale);
        const float m = t.first_step ? g : fmaf(momentum, mb, g);
        m_out = m;
        w_out = w - lr * m;
    };
    // 16-byte lanes when the three arrays allow it (gradients that live inside a bucket may start at any element)
    const bool vec = ((reinterpret_cast<uintptr_t>(t.param) | reinterpret_cast<uintptr_t>(t.grad) |
                       reinterpret_cast<uintptr_t>(t.momentum_buf)) & 15) == 0;
    int64_t done = 0;
    if (vec) {
        const int64_t quads = n >> 2;
        float4* p4 = reinterpret_cast<float4*>(t.param);
        float4* m4 = reinterpret_cast<float4*>(t.momentum_buf);
        const float4* g4 = reinterpret_cast<const float4*>(t.grad);
        for (int64_t i = tid; i < quads; i += stride) {
            const float4 w = p4[i], gr = g4[i];
            const float4 mb = t.first_step ? f4zero() : m4[i];
            float4 m, wn;
            upd(w.x, gr.x, mb.x, m.x, wn.x); upd(w.y, gr.y, mb.y, m.y, wn.y);
            upd(w.z, gr.z, mb.z, m.z, wn.z); upd(w.w, gr.w, mb.w, m.w, wn.w);
            m4[i] = m;
            p4[i] = wn;
        }
        done = quads << 2;
    }
    for (int64_t i = done + tid; i < n; i += stride) {
        float m, wn;
        upd(t.param[i], t.grad[i], t.first_step ? 0.f : t.momentum_buf[i], m, wn);
        t.momentum_buf[i] = m;
        t.param[i] = wn;
    }
}

extern "C" int semseg_sgd_step(const semseg_sgd_tensor* tensors_host, int n, const float* lr, float momentum, float grad_scale,
                               void* stream) {
    if (!tensors_host || n < 0 || !lr) return SEMSEG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += SGD_MAX_TENSORS) {
        SgdBatch b;
        b.n = min(SGD_MAX_TENSORS, n - base);
        int64_t maxn = 0;
        for (int i = 0; i < b.n; ++i) {
            b.t[i] = tensors_host[base + i];
            if (!b.t[i].param || !b.t[i].grad || !b.t[i].momentum_buf) return SEMSEG_EINVAL;
            if (b.t[i].numel > maxn) maxn = b.t[i].numel;
        }
        int gx = (int)((maxn + 256 * 8 - 1) / (256 * 8));
        if (gx < 1) gx = 1;
        if (gx > 512) gx = 512;
        hipLaunchKernelGGL(sgd_kernel, dim3(gx, b.n), dim3(256), 0, st, b, lr, momentum, grad_scale);
        SEMSEG_LAUNCH_CHECK();
    }
    return 0;
}

// ---- round 6: ONE pass over the weights where there were three -------------------------------------------------------------------
// After backward the step read every weight gradient three times: reduce_slabs_multi_kernel summed the deferred split weight
// gradients into .grad, sgd_kernel read .grad back, and wprep_absmax_kernel read every updated conv weight again for the exponent of
// its h2 planes.  sgd_fused_kernel (i) sums the slabs of a deferred gradient itself -- slab order 0, 1, 2, ... as the reduce kernel:
// the same bits -- and leaves the sum in .grad for whoever looks at it afterwards, (ii) leaves, per block, the maximum |w| of the
// UPDATED weights it wrote in the 512 partial slots the weight preparation reduces (a maximum is order-independent: the exponent
// is the one wprep_absmax_kernel finds).  train.py:115-127 semantics unchanged.
constexpr int SGD2_MAX_TENSORS = 48;
constexpr int SGD2_SLOTS = 512;           // partial maxima per tensor (csrc/weights_prep.hip reads as many when told so)
struct Sgd2Batch {
    semseg_sgd_tensor2 t[SGD2_MAX_TENSORS];
    int n;
};
static_assert(sizeof(Sgd2Batch) <= 4000, "kernel arguments");

__device__ __forceinline__ uint32_t sgd_absbits(float v) { return __float_as_uint(v) & 0x7fffffffu; }

__global__ __launch_bounds__(256) void sgd_fused_kernel(const Sgd2Batch b, const float* __restrict__ lr_ptr, float momentum,
                                                        float grad_scale) {
    __shared__ uint32_t red[4];
    const semseg_sgd_tensor2 t = b.t[blockIdx.y];
    const float lr = lr_ptr[0];
    const int64_t n = t.numel;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t mx = 0;
    auto upd = [&](float w, float gr, float mb, float& m_out, float& w_out) {
        const float g = fmaf(t.weight_decay, w, gr * grad_scale);
        const float m = t.first_step ? g : fmaf(momentum, mb, g);
        m_out = m;
        w_out = w - lr * m;
        mx = max(mx, sgd_absbits(w_out));
    };
    const bool vec = ((reinterpret_cast<uintptr_t>(t.param) | reinterpret_cast<uintptr_t>(t.grad) |
                       reinterpret_cast<uintptr_t>(t.momentum_buf) | reinterpret_cast<uintptr_t>(t.slabs)) & 15) == 0 &&
                     (!t.slabs || (n & 3) == 0);
    int64_t done = 0;
    if (vec) {
        const int64_t quads = n >> 2;
        float4* p4 = reinterpret_cast<float4*>(t.param);
        float4* m4 = reinterpret_cast<float4*>(t.momentum_buf);
        float4* g4 = reinterpret_cast<float4*>(t.grad);
        const float4* s4 = reinterpret_cast<const float4*>(t.slabs);
        for (int64_t i = tid; i < quads; i += stride) {
            const float4 w = p4[i];
            const float4 mb = t.first_step ? f4zero() : m4[i];
            float4 gr;
            if (s4) {
                gr = s4[i];
                for (int z = 1; z < t.splits; z += 4) {       // 4 loads in flight, additions in slab order (reduce_slabs_multi_kernel)
                    float4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (z + u < t.splits) v[u] = s4[(int64_t)(z + u) * quads + i];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (z + u < t.splits) { gr.x += v[u].x; gr.y += v[u].y; gr.z += v[u].z; gr.w += v[u].w; }
                }
                g4[i] = gr;
            } else {
                gr = g4[i];
            }
            float4 m, wn;
            upd(w.x, gr.x, mb.x, m.x, wn.x); upd(w.y, gr.y, mb.y, m.y, wn.y);
            upd(w.z, gr.z, mb.z, m.z, wn.z); upd(w.w, gr.w, mb.w, m.w, wn.w);
            m4[i] = m;
            p4[i] = wn;
        }
        done = quads << 2;
    }
    for (int64_t i = done + tid; i < n; i += stride) {
        float gr;
        if (t.slabs) {
            gr = t.slabs[i];
            for (int z = 1; z < t.splits; ++z) gr += t.slabs[(int64_t)z * n + i];
            t.grad[i] = gr;
        } else {
            gr = t.grad[i];
        }
        float m, wn;
        upd(t.param[i], gr, t.first_step ? 0.f : t.momentum_buf[i], m, wn);
        t.momentum_buf[i] = m;
        t.param[i] = wn;
    }
    uint32_t* slots = reinterpret_cast<uint32_t*>(t.absmax_slots);
    if (slots) {                                              // block-uniform
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) slots[blockIdx.x] = max(max(red[0], red[1]), max(red[2], red[3]));
        if (blockIdx.x == 0)
            for (int i = gridDim.x + threadIdx.x; i < SGD2_SLOTS; i += blockDim.x) slots[i] = 0u;
    }
}

extern "C" int semseg_sgd_step_fused(const semseg_sgd_tensor2* tensors_host, int n, const float* lr, float momentum, float grad_scale,
                                     void* stream) {
    if (!tensors_host || n < 0 || !lr) return SEMSEG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += SGD2_MAX_TENSORS) {
        Sgd2Batch b;
        b.n = min(SGD2_MAX_TENSORS, n - base);
        int64_t maxn = 0;
        for (int i = 0; i < b.n; ++i) {
            b.t[i] = tensors_host[base + i];
            if (!b.t[i].param || !b.t[i].grad || !b.t[i].momentum_buf || (b.t[i].slabs && b.t[i].splits < 1)) return SEMSEG_EINVAL;
            if (b.t[i].numel > maxn) maxn = b.t[i].numel;
        }
        int gx = (int)((maxn + 256 * 8 - 1) / (256 * 8));
        if (gx < 1) gx = 1;
        if (gx > SGD2_SLOTS) gx = SGD2_SLOTS;
        hipLaunchKernelGGL(sgd_fused_kernel, dim3(gx, b.n), dim3(256), 0, st, b, lr, momentum, grad_scale);
        SEMSEG_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int semseg_abi_version(void) { return 1; }

// ---------------------------------------------------------------- evaluation metrics --------------
// eval.py:74-84 on the device: pred = argmax_c scores (torch.max: first maximum), then utils.py:128-156
//   accuracy            : acc_sum = #(label >= 0 && pred == label), valid_sum = #(label >= 0)
//   intersectionAndUnion: area_intersection[c] = #(label >= 0 && pred == c && label == c)
//                         area_pred[c]         = #(label >= 0 && pred == c)      (predictions on unlabeled pixels dropped)
//                         area_lab[c]          = #(label == c)                   (0 <= c < C; np.histogram's range drops others)
// counts (int64, ACCUMULATED so that a caller can sum over images): [acc_sum, valid_sum, inter[C], pred[C], lab[C]].
// Integer counting: exact and order independent.  One wave per pixel, per-block LDS histograms flushed by atomics.
__global__ __launch_bounds__(256) void argmax_metrics_kernel(const float* __restrict__ scores, int ld,
                                                             const int64_t* __restrict__ label, int P, int C,
                                                             int64_t* __restrict__ pred_out,
                                                             unsigned long long* __restrict__ counts) {
    extern __shared__ unsigned int hist[];       // [2 + 3C]
    const int nh = 2 + 3 * C;
    for (int i = threadIdx.x; i < nh; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int p = blockIdx.x * 4 + wave; p < P; p += gridDim.x * 4) {
        const float* row = scores + (size_t)p * ld;
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int c = lane; c < C; c += 64) {
            const float v = row[c];
            if (v > best || (v == best && c < bi) || bi == 0x7fffffff) { best = v; bi = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
        }
        if (lane == 0) {
            if (pred_out) pred_out[p] = bi;
            if (label) {
                const int64_t l = label[p];
                if (l >= 0) {
                    atomicAdd(&hist[1], 1u);
                    atomicAdd(&hist[2 + C + bi], 1u);
                    if (l == bi) {
                        atomicAdd(&hist[0], 1u);
                        atomicAdd(&hist[2 + bi], 1u);
                    }
                    if (l < C) atomicAdd(&hist[2 + 2 * C + (int)l], 1u);
                }
            }
        }
    }
    __syncthreads();
    if (label && counts)
        for (int i = threadIdx.x; i < nh; i += blockDim.x)
            if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
}

extern "C" int semseg_argmax_metrics(const float* scores, int ld, const int64_t* label, int P, int C, int64_t* pred_out,
                                     int64_t* counts, void* stream) {
    if (!scores || P <= 0 || C <= 0 || C > 4096 || ld < C || (label && !counts) || (!label && !pred_out)) return SEMSEG_EINVAL;
    int blocks = ceil_div(P, 4 * 8);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(argmax_metrics_kernel, dim3(blocks), dim3(256), (size_t)(2 + 3 * C) * sizeof(unsigned int),
                       (hipStream_t)stream, scores, ld, label, P, C, pred_out, (unsigned long long*)counts);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// the same tallies from an existing prediction map (utils.py:128-156 called on label maps, eval.py:81-82)
__global__ __launch_bounds__(256) void label_metrics_kernel(const int64_t* __restrict__ pred, const int64_t* __restrict__ label,
                                                            int P, int C, unsigned long long* __restrict__ counts) {
    extern __shared__ unsigned int hist[];       // [2 + 3C]
    const int nh = 2 + 3 * C;
    for (int i = threadIdx.x; i < nh; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const int64_t l = label[p], q = pred[p];
        if (l < 0) continue;
        atomicAdd(&hist[1], 1u);
        if (q >= 0 && q < C) atomicAdd(&hist[2 + C + (int)q], 1u);
        if (l == q) {
            atomicAdd(&hist[0], 1u);
            if (q < C) atomicAdd(&hist[2 + (int)q], 1u);
        }
        if (l < C) atomicAdd(&hist[2 + 2 * C + (int)l], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nh; i += blockDim.x)
        if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
}

extern "C" int semseg_label_metrics(const int64_t* pred, const int64_t* label, int P, int C, int64_t* counts, void* stream) {
    if (!pred || !label || !counts || P <= 0 || C <= 0 || C > 4096) return SEMSEG_EINVAL;
    int blocks = ceil_div(P, 256 * 4);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(label_metrics_kernel, dim3(blocks), dim3(256), (size_t)(2 + 3 * C) * sizeof(unsigned int),
                       (hipStream_t)stream, pred, label, P, C, (unsigned long long*)counts);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

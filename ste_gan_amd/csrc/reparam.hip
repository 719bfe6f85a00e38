// Weight re-parametrisation kernels (weight norm; multi-tensor: one launch covers a whole network).
#include "common.h"

namespace {

// one wave per row: scale[row] = g[row] / ||v[row,:]||
__device__ __forceinline__ void wn_scale_row(const float* v, const float* g, float* scale, int row, int C) {
    const int lane = threadIdx.x & 63;
    const float* vr = v + (size_t)row * C;
    // four independent partial sums: a single accumulator makes every load wait for the one before it (a 2304-column row
    // = 36 round trips per wave: 30 us for the generator's 94 MB, 2.5x its HBM time)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = lane;
#pragma unroll 2
    for (; c + 192 < C; c += 256) {
        const float a0 = vr[c], a1 = vr[c + 64], a2 = vr[c + 128], a3 = vr[c + 192];
        s0 += a0 * a0; s1 += a1 * a1; s2 += a2 * a2; s3 += a3 * a3;
    }
    for (; c < C; c += 64) { const float a = vr[c]; s0 += a * a; }
    const float s = wave_sum((s0 + s1) + (s2 + s3));
    if (lane == 0) scale[row] = g[row] / sqrtf(s);
}

// one wave per row: ds = <dw,v>; dg = ds/||v||; dv = (g/||v||) dw - (ds g/||v||^3) v
// zero_dw: the raw gradient row is zeroed once it has been read (the arena then needs no fill before the next backward)
__device__ __forceinline__ void wn_backward_row(const float* v, const float* g, const float* dw, float* dv,
                                                float* dg, int row, int C, int zero_dw = 0) {
    const int lane = threadIdx.x & 63;
    const float* vr = v + (size_t)row * C;
    const float* dwr = dw + (size_t)row * C;
    float* dwz = const_cast<float*>(dwr);
    float* dvr = dv + (size_t)row * C;
    // rows of up to 64 * WN_REG columns stay in registers between the two passes (v and dw are read ONCE: 3 instead of 5
    // arena-sized streams -- the kernel is HBM-bound), and their loads are all issued before the first use
    constexpr int WN_REG = 48;
    if (C <= 64 * WN_REG) {
        float va[WN_REG], da[WN_REG];
#pragma unroll
        for (int q = 0; q < WN_REG; ++q) {
            const int c = lane + 64 * q;
            va[q] = 0.f; da[q] = 0.f;
            if (64 * q < C) {                                  // wave-uniform: short rows skip the rest
                va[q] = c < C ? vr[c] : 0.f;
                da[q] = c < C ? dwr[c] : 0.f;
            }
        }
        float n0 = 0.f, n1 = 0.f, d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int q = 0; q < WN_REG; q += 2) {
            n0 += va[q] * va[q]; d0 += va[q] * da[q];
            n1 += va[q + 1] * va[q + 1]; d1 += va[q + 1] * da[q + 1];
        }
        const float nn = wave_sum(n0 + n1), dot = wave_sum(d0 + d1);
        const float inv = 1.f / sqrtf(nn);
        const float gv = g[row];
        const float s = gv * inv;
        const float k = dot * gv * inv * inv * inv;
#pragma unroll
        for (int q = 0; q < WN_REG; ++q) {
            const int c = lane + 64 * q;
            if (64 * q < C && c < C) {
                dvr[c] = s * da[q] - k * va[q];
                if (zero_dw) dwz[c] = 0.f;
            }
        }
        if (lane == 0) dg[row] = dot * inv;
        return;
    }
    float nn = 0.f, dot = 0.f;
    for (int c = lane; c < C; c += 64) { float a = vr[c]; nn += a * a; dot += a * dwr[c]; }
    nn = wave_sum(nn);
    dot = wave_sum(dot);
    const float inv = 1.f / sqrtf(nn);
    const float gv = g[row];
    const float s = gv * inv;
    const float k = dot * gv * inv * inv * inv;
    for (int c = lane; c < C; c += 64) {
        dvr[c] = s * dwr[c] - k * vr[c];
        if (zero_dw) dwz[c] = 0.f;
    }
    if (lane == 0) dg[row] = dot * inv;
}

__global__ __launch_bounds__(256) void wn_scales_kernel(int n, const float* const* v, const float* const* g,
                                                        float* const* scale, const int* rows, const int* cols) {
    const int ti = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ti >= n || row >= rows[ti]) return;
    wn_scale_row(v[ti], g[ti], scale[ti], row, cols[ti]);
}

__global__ __launch_bounds__(256) void wn_scale1_kernel(const float* v, const float* g, float* scale, int rows, int cols) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    wn_scale_row(v, g, scale, row, cols);
}

__global__ __launch_bounds__(256) void wn_backward_kernel(int n, const float* const* v, const float* const* g,
                                                          const float* const* dw, float* const* dv, float* const* dg,
                                                          const int* rows, const int* cols, int zero_dw) {
    const int ti = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ti >= n || row >= rows[ti]) return;
    wn_backward_row(v[ti], g[ti], dw[ti], dv[ti], dg[ti], row, cols[ti], zero_dw);
}

__global__ __launch_bounds__(256) void wn_backward1_kernel(const float* v, const float* g, const float* dw, float* dv,
                                                           float* dg, int rows, int cols) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    wn_backward_row(v, g, dw, dv, dg, row, cols);
}

// dx_h != nullptr: dx also as a bf16 activation image [B][C / 8][T][8] (C % 8 == 0)
__global__ __launch_bounds__(256) void fold_upsample2_kernel(const float* __restrict__ dxe, float* __restrict__ dx,
                                                             long total, unsigned short* __restrict__ dx_h, int Cc, int T) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (; i < total; i += stride) {
        const float2 v = reinterpret_cast<const float2*>(dxe)[i];
        const float r = v.x + v.y;
        dx[i] = r;
        if (dx_h) {
            const long row = i / T;
            const int t = (int)(i - row * T), b = (int)(row / Cc), c = (int)(row - (long)b * Cc);
            img_store(dx_h, Cc >> 3, T, b, c, t, r);
        }
    }
}

// diagnostics: one lane writes the chip's constant 100 MHz real-time counter -- a time mark inside a captured graph
__global__ void timestamp_kernel(unsigned long long* slot) { *slot = __builtin_amdgcn_s_memrealtime(); }

}  // namespace

/* Diagnostic (scripts/step_marks.py): enqueue a one-lane kernel that stores s_memrealtime (100 MHz, the same clock on every
 * XCD) into *slot: time marks between the kernels of a stream, also inside a captured hipGraph.  Not part of the path. */
extern "C" int stegan_timestamp(uint64_t* slot, void* stream) {
    STEGAN_CHECK_ARG(slot);
    hipLaunchKernelGGL(timestamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)slot);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_wn_scales(int32_t n, const float* const* v, const float* const* g, float* const* scale,
                                const int32_t* rows, const int32_t* cols, int32_t max_rows, void* stream) {
    STEGAN_CHECK_ARG(n > 0 && v && g && scale && rows && cols && max_rows > 0);
    hipLaunchKernelGGL(wn_scales_kernel, dim3(ceil_div(max_rows, 4), n), dim3(256), 0, (hipStream_t)stream, n, v, g,
                       scale, rows, cols);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_wn_backward_z(int32_t n, const float* const* v, const float* const* g, float* const* dw,
                                    float* const* dv, float* const* dg, const int32_t* rows, const int32_t* cols,
                                    int32_t max_rows, int32_t zero_dw, void* stream) {
    STEGAN_CHECK_ARG(n > 0 && v && g && dw && dv && dg && rows && cols && max_rows > 0);
    hipLaunchKernelGGL(wn_backward_kernel, dim3(ceil_div(max_rows, 4), n), dim3(256), 0, (hipStream_t)stream, n, v, g,
                       (const float* const*)dw, dv, dg, rows, cols, zero_dw);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_wn_backward(int32_t n, const float* const* v, const float* const* g, const float* const* dw,
                                  float* const* dv, float* const* dg, const int32_t* rows, const int32_t* cols,
                                  int32_t max_rows, void* stream) {
    return stegan_wn_backward_z(n, v, g, (float* const*)dw, dv, dg, rows, cols, max_rows, 0, stream);
}

extern "C" int stegan_wn_scale1(const float* v, const float* g, float* scale, int32_t rows, int32_t cols, void* stream) {
    STEGAN_CHECK_ARG(v && g && scale && rows > 0 && cols > 0);
    hipLaunchKernelGGL(wn_scale1_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, v, g, scale, rows, cols);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_wn_backward1(const float* v, const float* g, const float* dw, float* dv, float* dg,
                                   int32_t rows, int32_t cols, void* stream) {
    STEGAN_CHECK_ARG(v && g && dw && dv && dg && rows > 0 && cols > 0);
    hipLaunchKernelGGL(wn_backward1_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, v, g, dw, dv, dg,
                       rows, cols);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_fold_upsample2_img(const float* dxe, float* dx, uint16_t* dx_h, int64_t rows, int32_t C, int32_t T,
                                         void* stream) {
    STEGAN_CHECK_ARG(dxe && dx && rows > 0 && T > 0);
    if (dx_h) STEGAN_CHECK_ARG(C > 0 && C % 8 == 0 && rows % C == 0);
    const long total = (long)rows * T;   // T = folded (output) length; dxe rows have 2T floats, 8-byte aligned pairs
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fold_upsample2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dxe, dx, total, dx_h, C, T);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_fold_upsample2(const float* dxe, float* dx, int64_t rows, int32_t T, void* stream) {
    return stegan_fold_upsample2_img(dxe, dx, nullptr, rows, 0, T, stream);
}

extern "C" int stegan_abi_version(void) { return STEGAN_ABI_VERSION; }
extern "C" const char* stegan_operand_type(void) { return STEGAN_OPERAND_NAME; }

extern "C" int stegan_selftest_host(void) {
    // magic division must be exact for every divisor / dividend range the kernels use
    for (uint32_t d = 1; d <= 4200; ++d) {
        FastDiv f = make_fastdiv(d);
        for (uint32_t n = 0; n < 70000; n += (n < 5000 ? 1 : 97)) {
            if (fd_div(n, f) != n / d) return 1;
        }
        const uint32_t big[] = {(1u << 24) - 1, (1u << 24), (1u << 30) + 12345u, 0x7fffffffu};
        for (uint32_t n : big)
            if (fd_div(n, f) != n / d) return 2;
    }
    return 0;
}

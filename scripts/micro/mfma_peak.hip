// Measures the sustained v_mfma_f32_32x32x2_f32 rate of this MI355X (register operands only).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters, const char* tag) {
    float* out; hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f, 0.25f);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = (double)blocks * 4 * iters * NACC * 2.0 * 32 * 32 * 2;
    printf("%s blocks=%d waves/SIMD=%d nacc=%d: %.3f ms  %.1f TFLOP/s\n", tag, blocks, blocks / 256, NACC, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    run<1>(256, 20000, "1wave/SIMD");
    run<4>(256, 5000, "1wave/SIMD");
    run<1>(512, 20000, "2waves/SIMD");
    run<1>(1024, 10000, "4waves/SIMD");
    run<2>(512, 10000, "2waves/SIMD");
    return 0;
}

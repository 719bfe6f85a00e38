// Validates __builtin_amdgcn_global_load_lds semantics on gfx950: per-lane global address, LDS dest = uniform base + lane*size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
__global__ void k(const float* src, const int* idx4, const int* idx1, float* out) {
    __shared__ __align__(16) float lds[2048];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // 16-byte form: thread -> float4 #tid of lds[0..1023]; source float4 index idx4[tid]
    __builtin_amdgcn_global_load_lds((gptr_t)(src + 4 * idx4[tid]), (lds_ptr_t)(lds + wave * 256), 16, 0, 0);
    // 4-byte form: thread -> float #tid of lds[1024..1279]
    __builtin_amdgcn_global_load_lds((gptr_t)(src + idx1[tid]), (lds_ptr_t)(lds + 1024 + wave * 64), 4, 0, 0);
    __syncthreads();
    for (int i = tid; i < 1280; i += 256) out[i] = lds[i];
}
int main() {
    const int N = 4096;
    std::vector<float> h(N); for (int i = 0; i < N; ++i) h[i] = (float)i;
    std::vector<int> i4(256), i1(256);
    for (int t = 0; t < 256; ++t) { i4[t] = (t * 37 + 11) % (N / 4); i1[t] = (t * 53 + 7) % N; }
    float *d, *o; int *d4, *d1;
    (void)hipMalloc(&d, N * 4); (void)hipMalloc(&o, 1280 * 4); (void)hipMalloc(&d4, 1024); (void)hipMalloc(&d1, 1024);
    (void)hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d4, i4.data(), 1024, hipMemcpyHostToDevice); (void)hipMemcpy(d1, i1.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, d4, d1, o);
    std::vector<float> r(1280); (void)hipMemcpy(r.data(), o, 1280 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) for (int c = 0; c < 4; ++c) if (r[t * 4 + c] != (float)(4 * i4[t] + c)) ++bad;
    for (int t = 0; t < 256; ++t) if (r[1024 + t] != (float)i1[t]) ++bad;
    printf("glds test: %d mismatches (of 1280)\n", bad);
    return bad != 0;
}

// Cycles per iteration of an empty producer / consumer loop: 512 threads, one raw s_barrier per iteration (as the ring of
// conv_bf16.hip), 144 blocks.  Variants: 0 barrier only; 1 + the loop bookkeeping of the conv's producers (chunk id with
// wrap-around, stage counter, a switch-dispatched wait).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/barrier_cost.hip -o scripts/micro/barrier_cost && scripts/micro/barrier_cost
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
template <int V>
__global__ __launch_bounds__(512) void k(unsigned long long* cyc, int n, int rot, unsigned* sink) {
    unsigned long long t0 = 0;
    int stage = 0, acc = 0;
    if (threadIdx.x == 0) t0 = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < n; ++j) {
        if (V == 1) {
            const int cid = j + rot - ((j + rot) >= n ? n : 0);
            acc += cid * stage;
            stage = stage + 1 == 2 ? 0 : stage + 1;
            switch ((n - j) & 3) { case 0: asm volatile("s_waitcnt vmcnt(0)"); break; case 1: asm volatile("s_waitcnt vmcnt(1)"); break;
                                   case 2: asm volatile("s_waitcnt vmcnt(2)"); break; default: asm volatile("s_waitcnt vmcnt(3)"); break; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = (__builtin_amdgcn_s_memtime() - t0) / n;
    if (acc == 0x7fffffff) sink[0] = acc;
}
int main() {
    unsigned long long* cyc; unsigned* sink;
    hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 4);
    for (int v = 0; v < 2; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            if (v == 0) hipLaunchKernelGGL(k<0>, dim3(144), dim3(512), 0, 0, cyc, 200, 3, sink);
            else hipLaunchKernelGGL(k<1>, dim3(144), dim3(512), 0, 0, cyc, 200, 3, sink);
            hipDeviceSynchronize();
        }
        std::vector<unsigned long long> h(144);
        hipMemcpy(h.data(), cyc, 144 * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("variant %d: %llu cycles per iteration (min %llu, max %llu)\n", v, h[72], h[0], h[143]);
    }
    return 0;
}

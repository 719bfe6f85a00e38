// How fast can the 4 producer waves of a 512-thread block move a chunk from global memory (L2 / Infinity-Cache resident)
// into LDS?   (a) LDS-DMA: global_load_lds_dwordx4, 1 KiB per wave-instruction, as the image path of conv_bf16.hip does;
//             (b) registers: global_load_dwordx4 -> ds_write_b128, U loads in flight per lane.
// Each block walks its own 1 MiB region chunk by chunk (CHUNK bytes per iteration, barrier per iteration like the conv's
// ring), G blocks at once.  Reported: cycles per chunk and bytes per cycle per CU, median over blocks.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/stage_rate.hip -o scripts/micro/stage_rate && scripts/micro/stage_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
constexpr int ITERS = 24;

template <int MODE, int KB>      // KB: chunk size in KiB (16 / 32 / 64)
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ src, unsigned long long* cyc, unsigned* sink, int shared) {
    extern __shared__ __align__(16) char smem[];       // 2 stages x KB KiB
    const int t = threadIdx.x;
    const bool producer = t >= 256;
    const int tid = t & 255, wave = tid >> 6;
    constexpr int PIECES = KB * 1024 / 16, PER = PIECES / 256;      // 16-byte pieces per chunk, per producer thread
    const uint4* base = src + (shared ? (size_t)(blockIdx.x & 7) * (1 << 16) : (size_t)blockIdx.x * (1 << 16));   // 1 MiB per block, or one per XCD (L2 hits)
    unsigned long long t0 = 0;
    unsigned acc = 0;
    if (t == 256) t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < ITERS; ++c) {
        char* stage = smem + (c & 1) * KB * 1024;
        const uint4* chunk = base + (size_t)(c % (1024 / KB)) * PIECES;
        if (producer) {
            if (MODE == 0) {
#pragma unroll
                for (int it = 0; it < PER; ++it)
                    __builtin_amdgcn_global_load_lds((gptr_t)(chunk + it * 256 + tid), (lds_ptr_t)(stage + (it * 256 + wave * 64) * 16), 16, 0, 0);
                __builtin_amdgcn_s_waitcnt(0);
            } else if (MODE == 3 || MODE == 4) {                 // DMA through a buffer descriptor; 4: runs of 12 pieces (192 B) 18 KiB apart
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 1 << 20, 0x00020000);
#pragma unroll
                for (int it = 0; it < PER; ++it) {
                    const int piece = it * 256 + tid;
                    const int off = MODE == 3 ? piece * 16 : ((piece / 12) * 18432 + (piece % 12) * 16) & ((1 << 20) - 16);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(stage + (it * 256 + wave * 64) * 16), 16, off,
                                                             (c % (1024 / KB)) * (MODE == 3 ? PIECES * 16 : 192), 0, 0);
                }
                __builtin_amdgcn_s_waitcnt(0);
            } else if (MODE == 2) {                              // mixed: a quarter of the chunk through registers, the rest by DMA
                constexpr int PR = PER / 4 > 0 ? PER / 4 : 1;
                uint4 r[PR];
#pragma unroll
                for (int it = 0; it < PR; ++it) r[it] = chunk[it * 256 + tid];
#pragma unroll
                for (int it = PR; it < PER; ++it)
                    __builtin_amdgcn_global_load_lds((gptr_t)(chunk + it * 256 + tid), (lds_ptr_t)(stage + (it * 256 + wave * 64) * 16), 16, 0, 0);
#pragma unroll
                for (int it = 0; it < PR; ++it) *(uint4*)(stage + (it * 256 + tid) * 16) = r[it];
                __builtin_amdgcn_s_waitcnt(0);
            } else {
                uint4 r[PER];
#pragma unroll
                for (int it = 0; it < PER; ++it) r[it] = chunk[it * 256 + tid];
#pragma unroll
                for (int it = 0; it < PER; ++it) *(uint4*)(stage + (it * 256 + tid) * 16) = r[it];
                __builtin_amdgcn_s_waitcnt(0);
            }
        }
        __syncthreads();
        if (!producer) acc += *(const unsigned*)(stage + tid * 16 + (c & 3) * 4);      // the consumers look at the stage
    }
    if (t == 256) cyc[blockIdx.x] = (__builtin_amdgcn_s_memtime() - t0) / ITERS;
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int KB>
void run(const uint4* src, unsigned long long* cyc, unsigned* sink, int G, int shared) {
    hipFuncSetAttribute((const void*)k<MODE, KB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k<MODE, KB>), dim3(G), dim3(512), 2 * KB * 1024, 0, src, cyc, sink, shared);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(G);
    hipMemcpy(h.data(), cyc, G * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-9s %s chunk %2d KiB, %3d blocks: %6llu cycles per chunk = %5.1f B/clk per CU (min %llu, max %llu)\n",
           MODE == 0 ? "LDS-DMA" : MODE == 1 ? "registers" : MODE == 2 ? "mixed 1:3" : MODE == 3 ? "DMA bufdesc" : "DMA 192B runs", shared ? "shared (L2 hits)" : "private (misses) ", KB, G, h[G / 2], KB * 1024.0 / h[G / 2], h[0], h[G - 1]);
}

int main() {
    uint4* src; unsigned long long* cyc; unsigned* sink;
    hipMalloc(&src, (size_t)256 << 20); hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 4);
    hipMemset(src, 1, (size_t)256 << 20);
    for (int shared = 0; shared < 2; ++shared)
    for (int G : {144, 256}) {
        run<0, 16>(src, cyc, sink, G, shared); run<3, 16>(src, cyc, sink, G, shared); run<4, 16>(src, cyc, sink, G, shared);
        run<0, 32>(src, cyc, sink, G, shared); run<3, 32>(src, cyc, sink, G, shared); run<4, 32>(src, cyc, sink, G, shared);
        
    }
    return 0;
}

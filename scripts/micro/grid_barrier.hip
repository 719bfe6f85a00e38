// What does a grid-wide barrier cost on MI355X?  (the estimate for a multi-layer persistent kernel
// needs the price of one layer boundary.)  G blocks of 512 threads, all co-resident (G <= 256: one per CU).  Each round:
// every thread stores a 16-byte piece of a "layer output" (64 KiB per block), then the barrier of cdna_hip_programming.md
// section 6 (G16): waves drain their stores, the workgroup meets, ONE lane releases at agent scope and bumps a counter, polls
// it with agent-scope acquire loads until every block of the round has arrived (bounded: a poll limit ends the kernel with an
// error flag instead of hanging), the workgroup meets again; then every thread reads a piece ANOTHER block wrote (a different
// XCD's, by construction) and checks it.  Reported: cycles (s_memtime) per round, median over blocks, for G = 64 / 128 / 256,
// with and without the payload.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/grid_barrier.hip -o scripts/micro/grid_barrier && scripts/micro/grid_barrier
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int ROUNDS = 20;
constexpr unsigned POLL_LIMIT = 4000000u;

__global__ __launch_bounds__(512) void k(unsigned* counter, uint4* payload, unsigned long long* cycles, unsigned* err, int with_payload, int nap) {
    const unsigned G = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    __shared__ unsigned ok_flag;
    unsigned long long t0 = 0, total = 0;
    for (int r = 0; r < ROUNDS; ++r) {
        if (t == 0) t0 = __builtin_amdgcn_s_memtime();
        if (with_payload) {
#pragma unroll
            for (int q = 0; q < 8; ++q)                        // 8 x 512 x 16 B = 64 KiB per block and round
                payload[((size_t)b * 8 + q) * 512 + t] = make_uint4(r, b, q, t);
        }
        __builtin_amdgcn_s_waitcnt(0);                          // this wave's stores have left
        __syncthreads();
        if (t == 0) {
            __atomic_thread_fence(__ATOMIC_RELEASE);            // hipcc: agent scope on device code -> L2 write-back
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(r + 1) * G;
            unsigned polls = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++polls < POLL_LIMIT) {
                if (nap) __builtin_amdgcn_s_sleep(8);          // back off: fewer polls in flight at the counter's L2 channel
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            ok_flag = polls < POLL_LIMIT;
            if (!ok_flag) atomicOr(err, 1u);
        }
        __syncthreads();
        if (!ok_flag) return;                                   // every wave leaves: the grid drains
        if (with_payload) {
            const unsigned nb = (b + 1) % G;                    // the neighbour's id: another XCD under round-robin placement
            const uint4 v = payload[((size_t)nb * 8 + (r & 7)) * 512 + t];        // plain load behind the acquire
            if (v.x != (unsigned)r || v.y != nb || v.w != t) atomicOr(err, 2u);
        }
        __syncthreads();
        if (t == 0) total += __builtin_amdgcn_s_memtime() - t0;
    }
    if (t == 0) cycles[b] = total / ROUNDS;
}

int main() {
    unsigned *counter, *err;
    uint4* payload;
    unsigned long long* cyc;
    hipMalloc(&counter, 4); hipMalloc(&err, 4); hipMalloc(&payload, (size_t)256 * 8 * 512 * 16); hipMalloc(&cyc, 256 * 8);
    for (int nap = 0; nap < 2; ++nap)
    for (int with_payload = 0; with_payload < 2; ++with_payload)
        for (int G : {64, 128, 256}) {
            for (int rep = 0; rep < 2; ++rep) {                 // second run: warm
                hipMemset(counter, 0, 4); hipMemset(err, 0, 4);
                hipLaunchKernelGGL(k, dim3(G), dim3(512), 0, 0, counter, payload, cyc, err, with_payload, nap);
                hipDeviceSynchronize();
            }
            std::vector<unsigned long long> h(G);
            unsigned e = 0;
            hipMemcpy(h.data(), cyc, G * 8, hipMemcpyDeviceToHost);
            hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            printf("G=%3d payload=%d sleep-in-poll=%d: round = %llu cycles median (min %llu, max %llu) (s_memtime, ~2 GHz); err flags %u\n", G,
                   with_payload, nap, h[G / 2], h[0], h[G - 1], e);
        }
    return 0;
}

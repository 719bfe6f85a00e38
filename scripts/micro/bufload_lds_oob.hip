// Does `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer descriptor) write ZEROS to LDS for lanes whose offset is
// out of the descriptor's range, or does it leave their 16 bytes untouched?  (The bf16 activation images of round 3 are
// staged this way; halo / padding / ragged-channel pieces carry an out-of-range offset and must read as zero.)
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/bufload_lds_oob.hip -o scripts/micro/bufload_lds_oob && scripts/micro/bufload_lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const unsigned* src, unsigned* out, int bytes, int soff) {
    extern __shared__ __align__(16) char smem[];
    unsigned* l = (unsigned*)smem;
    for (int i = threadIdx.x; i < 64 * 4 * 2; i += 64) l[i] = 0xdeadbeefu;
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    int voff = threadIdx.x * 16;
    if (threadIdx.x & 1) voff = 0x7ffff000;                 // odd lanes: out of range
    if (threadIdx.x < 48)                                   // partial exec: lanes 48..63 take no part
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem), 16, voff, soff, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 4 * 2; i += 64) out[i] = l[i];
}
int main() {
    const int n = 64 * 4 + 64;
    std::vector<unsigned> h(n);
    for (int i = 0; i < n; ++i) h[i] = 0x1000 + i;
    unsigned *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 64 * 4 * 2 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, o, 64 * 16, 16);       // soffset 16: lane i reads piece i + 1
    std::vector<unsigned> r(64 * 4 * 2);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const unsigned got = r[lane * 4];
        unsigned want;
        const char* what;
        if (lane >= 48) { want = 0xdeadbeefu; what = "inactive lane: untouched"; }
        else if (lane & 1) { want = 0u; what = "out-of-range offset: zero"; }
        else if (lane == 62) { want = 0u; what = ""; }
        else { want = 0x1000 + (lane + 1) * 4; what = "in range (soffset applied)"; }
        if (got != want) { ++bad; printf("lane %d: got %08x want %08x (%s)\n", lane, got, want, what); }
    }
    printf("second KiB untouched: %s\n", r[64 * 4] == 0xdeadbeefu ? "yes" : "NO");
    printf(bad ? "RESULT: %d lanes differ\n" : "RESULT OK: out-of-range lanes write zeros, inactive lanes write nothing\n", bad);
    return bad != 0;
}

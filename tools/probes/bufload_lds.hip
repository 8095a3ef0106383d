// probe: buffer_load_dwordx4 ... lds with an out-of-range voffset - does the LDS destination receive zeros?  (hipcc --offload-arch=gfx950 -O2)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const unsigned short* in, unsigned short* out, int n, int which) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 256 * 8; i += 256) ((unsigned short*)smem)[i] = 0x7777;
    __syncthreads();
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 2, 0x00020000);
    int voff = (threadIdx.x * 16);
    if (which & (1 << (threadIdx.x & 7))) voff = -1;     // out of range
    const int wrow = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * 64 * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + wrow), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256 * 8; i += 256) out[i] = ((unsigned short*)smem)[i];
}
int main() {
    const int n = 256 * 8;
    std::vector<unsigned short> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = (unsigned short)(i + 1);
    unsigned short *d, *r;
    hipMalloc(&d, n * 2); hipMalloc(&r, n * 2);
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
    k<<<1, 256, 256 * 16>>>(d, r, n, 0xA5);
    hipMemcpy(o.data(), r, n * 2, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        const bool oob = 0xA5 & (1 << (t & 7));
        for (int e = 0; e < 8; ++e) { const unsigned short want = oob ? 0 : h[t * 8 + e]; if (o[t * 8 + e] != want) { if (bad < 5) printf("t%d e%d got %x want %x\n", t, e, o[t * 8 + e], want); ++bad; } }
    }
    printf("bad %d\n", bad);
    return bad != 0;
}

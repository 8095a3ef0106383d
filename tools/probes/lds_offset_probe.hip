// Probe (gfx950): do DS instructions give the same result when part of the address sits in the 16-bit immediate offset field instead of the
// VGPR?  LDS allocation is 128 KB; every 16-bit element holds (element index & 0xffff), so any mis-addressed access shows up.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int OFF>
__device__ void one(unsigned total, unsigned* out, int slot) {
    const unsigned lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
    const unsigned lanepart = (g * 4 + (i >> 2)) * 160 + (i & 3) * 8;      // the tr_frag addressing of pn2_vit.hip
    const unsigned full = total + lanepart, split = total - OFF + lanepart;
    u32x2 a, b; u32x4 c, d; unsigned e, f;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(full));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2\n s_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(split), "n"(OFF));
    asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(c) : "v"(total + lane * 16));
    asm volatile("ds_read_b128 %0, %1 offset:%2\n s_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(total - OFF + lane * 16), "n"(OFF));
    asm volatile("ds_read_u16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(total + lane * 2));
    asm volatile("ds_read_u16 %0, %1 offset:%2\n s_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(total - OFF + lane * 2), "n"(OFF));
    unsigned bad = (a.x != b.x || a.y != b.y) ? 1u : 0u;
    bad |= (c.x != d.x || c.y != d.y || c.z != d.z || c.w != d.w) ? 2u : 0u;
    bad |= (e != f) ? 4u : 0u;
    // 16-bit store through the immediate, read back through the VGPR
    const unsigned wa = total + 4096 + lane * 2;
    asm volatile("ds_write_b16 %0, %1 offset:%2\n s_waitcnt lgkmcnt(0)" :: "v"(wa - OFF), "v"(0xbeefu), "n"(OFF) : "memory");
    unsigned r;
    asm volatile("ds_read_u16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(wa));
    bad |= (r != 0xbeefu) ? 8u : 0u;
    unsigned any = 0;
    for (int o = 0; o < 64; ++o) any |= __shfl(bad, o);
    if (lane == 0) out[slot] = any;
}

__global__ void probe(unsigned* out) {
    extern __shared__ unsigned short lds[];
    for (int i = threadIdx.x; i < 64 * 1024; i += blockDim.x) lds[i] = (unsigned short)i;
    __syncthreads();
    if (threadIdx.x < 64) {
        one<2560>(20480u, out, 0);      // small immediate, low address
        one<20480>(24576u, out, 1);     // larger immediate, all below 64 KB
        one<36864>(40960u, out, 2);     // immediate > 32 KB
        one<40000>(50000u & ~15u, out, 3);
        one<60000>(61440u, out, 4);
        one<8192>(69632u, out, 5);      // everything above 64 KB
        one<8192>(65536u + 2048u, out, 6);   // VGPR part below 64 KB, sum above
        one<58368>(70000u & ~15u, out, 7);   // what the compiler generated for the probabilities of attn_fwd_mfma_k<3>
    }
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 64); (void)hipMemset(d, 0xff, 64);
    (void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 128 * 1024, 0, d);
    unsigned h[16]; (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    const char* names[8] = {"imm 2560 @20K", "imm 20480 @24K", "imm 36864 @40K", "imm 40000 @50K", "imm 60000 @60K", "imm 8192 @68K", "imm 8192 across 64K", "imm 58368 @70K"};
    for (int i = 0; i < 8; ++i) printf("%-22s mismatch mask %u  (1 = tr_b16, 2 = b128, 4 = u16 read, 8 = b16 write)\n", names[i], h[i]);
    return 0;
}

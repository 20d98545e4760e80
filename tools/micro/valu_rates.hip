// Issue cost of the VALU instructions the sampler's Philox is made of, on one SIMD: N independent chains of the same instruction in
// a loop, one wave per SIMD and four.  Prints cycles per wave-instruction (s_memtime ticks are converted with the measured
// wall-clock rate of the counter).  Diagnostic, not part of the library:  hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int OP>
__global__ void k_rate(uint32_t *out, unsigned long long *ticks, int iters, uint32_t seed)
{
    uint32_t a[8];
    unsigned long long acc[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = seed + threadIdx.x * 8 + i;
        acc[i] = a[i];
    }
    const uint32_t m = 0xD2511F53u + seed;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(m), "v"(a[i]) : "vcc");
            if (OP == 1) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 4) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(a[i]) : "v"(m));
            if (OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i] ^ static_cast<uint32_t>(acc[i]) ^ static_cast<uint32_t>(acc[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char *name, int waves, double ticks_per_us)
{
    const int iters = 4096, blocks = 256;
    uint32_t *out;
    unsigned long long *ticks;
    hipMalloc(&out, sizeof(uint32_t) * blocks * waves * 64);
    hipMalloc(&ticks, sizeof(unsigned long long) * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_rate<OP><<<blocks, waves * 64>>>(out, ticks, iters, 1);
    hipEventRecord(e0);
    k_rate<OP><<<blocks, waves * 64>>>(out, ticks, iters, 2);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256];
    hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < blocks; ++i) mean += static_cast<double>(h[i]);
    mean /= blocks;
    const double insts_per_simd = 8.0 * iters * (waves / 4.0 < 1 ? 1 : waves / 4.0);
    printf("%-16s %2d waves/block: %8.0f ticks per block, kernel %7.1f us", name, waves, mean, ms * 1e3);
    if (ticks_per_us > 0) printf("  -> %5.2f us in-kernel", mean / ticks_per_us);
    printf("  | %6.2f ns per wave-instruction per SIMD (kernel time / %g)\n", ms * 1e6 / insts_per_simd, insts_per_simd);
    hipFree(out);
    hipFree(ticks);
}

int main()
{
    for (int waves : {4, 16}) {
        run<0>("v_mad_u64_u32", waves, 0);
        run<1>("v_mul_hi_u32", waves, 0);
        run<2>("v_mul_lo_u32", waves, 0);
        run<3>("v_xor_b32", waves, 0);
        run<4>("v_bitop3_b32", waves, 0);
        run<5>("v_add_u32", waves, 0);
    }
    return 0;
}

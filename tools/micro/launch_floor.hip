// Microbenchmark (development tool): what does a launch of G workgroups x B threads cost on this chip when the
// workgroups do next to nothing?  Separates wave dispatch from the per-workgroup chain of memory round trips.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_empty(int *p) { if (p == nullptr && threadIdx.x == 12345) p[0] = 1; }

__global__ void k_lds(int *p)
{
    extern __shared__ int s[];
    if (threadIdx.x == 0) s[0] = blockIdx.x;
    __syncthreads();
    if (p == nullptr && s[0] == -1) p[0] = 1;
}

// one dependent global load, then a store
__global__ void k_chain1(const int *in, int *out)
{
    extern __shared__ int s[];
    int v = in[blockIdx.x];
    if (threadIdx.x == 0) s[0] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[0] + 1;
}

// streaming: every thread loads NB 16-B pieces (row) + 3 dwords (ids), writes them to LDS, barrier, one store
template <int NB>
__global__ void k_stream(const uint4 *rows, const int *ids, int *out, int row_pieces)
{
    extern __shared__ uint4 s4[];
    const int z = blockIdx.x, tid = threadIdx.x;
    int a = ids[z * 4096 + tid], b = ids[z * 4096 + 512 + tid], c = ids[z * 4096 + 1024 + tid];
    uint4 pc[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m) pc[m] = rows[(size_t)z * row_pieces + tid + m * blockDim.x];
#pragma unroll
    for (int m = 0; m < NB; ++m) s4[tid + m * blockDim.x] = pc[m];
    __syncthreads();
    if (tid == 0) out[z] = a + b + c + s4[7].x;
}

template <typename F>
float time_it(F f, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / reps;
}

int main()
{
    int *d_in, *d_out, *d_ids;
    uint4 *d_rows;
    const int Z = 4096;
    hipMalloc(&d_in, Z * 4);
    hipMalloc(&d_out, Z * 4);
    hipMalloc(&d_ids, (size_t)Z * 4096 * 4);
    hipMalloc(&d_rows, (size_t)Z * 2048 * 16 * 24);  // 24 hours of 32-KiB rows, so that back-to-back launches do not hit in cache
    hipMemset(d_in, 0, Z * 4);
    hipMemset(d_ids, 0, (size_t)Z * 4096 * 4);
    hipMemset(d_rows, 0, (size_t)Z * 2048 * 16 * 24);
    const int reps = 200;
    for (int g : {1024, 4096}) {
        for (int b : {256, 512, 1024}) {
            printf("empty        grid %5d block %4d lds     0: %7.2f us\n", g, b, time_it([&] { hipLaunchKernelGGL(k_empty, dim3(g), dim3(b), 0, 0, d_out); }, reps));
            printf("lds+barrier  grid %5d block %4d lds 16384: %7.2f us\n", g, b, time_it([&] { hipLaunchKernelGGL(k_lds, dim3(g), dim3(b), 16384, 0, d_out); }, reps));
            printf("load+barrier grid %5d block %4d lds 16384: %7.2f us\n", g, b, time_it([&] { hipLaunchKernelGGL(k_chain1, dim3(g), dim3(b), 16384, 0, d_in, d_out); }, reps));
        }
    }
    int hour = 0;
    printf("stream 16 KiB rows + ids, grid 4096 block 512 (NB=2): %7.2f us\n",
           time_it([&] { hipLaunchKernelGGL(k_stream<2>, dim3(Z), dim3(512), 16384, 0, d_rows + (size_t)(hour++ % 24) * Z * 2048, d_ids, d_out, 1024); }, reps));
    printf("stream 32 KiB rows + ids, grid 4096 block 512 (NB=4): %7.2f us\n",
           time_it([&] { hipLaunchKernelGGL(k_stream<4>, dim3(Z), dim3(512), 32768, 0, d_rows + (size_t)(hour++ % 24) * Z * 2048, d_ids, d_out, 2048); }, reps));
    printf("stream 16 KiB rows + ids, grid 4096 block 256 (NB=4): %7.2f us\n",
           time_it([&] { hipLaunchKernelGGL(k_stream<4>, dim3(Z), dim3(256), 16384, 0, d_rows + (size_t)(hour++ % 24) * Z * 2048, d_ids, d_out, 1024); }, reps));
    printf("stream 16 KiB rows + ids, grid 4096 block 1024 (NB=1): %7.2f us\n",
           time_it([&] { hipLaunchKernelGGL(k_stream<1>, dim3(Z), dim3(1024), 16384, 0, d_rows + (size_t)(hour++ % 24) * Z * 2048, d_ids, d_out, 1024); }, reps));
    return 0;
}

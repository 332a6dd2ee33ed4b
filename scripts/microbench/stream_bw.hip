// What does a streaming kernel reach on this part?  Variants of "read two fp32 arrays, write one" (the traffic pattern of
// ds_bn_bwd_apply: 12 B per element) and of a plain copy, over 205 MB arrays (past the 256 MiB Infinity Cache in total).
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/stream_bw.hip -o /tmp/stream_bw && /tmp/stream_bw
// Prints TB/s per variant: grid size (blocks per CU), float4 items per thread per pass (loads of all items before any
// store), nontemporal loads / stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT_LD, bool NT_ST, bool TWO_IN>
__global__ __launch_bounds__(256) void stream_kernel(const f32x4 *__restrict__ a, const f32x4 *__restrict__ b,
                                                     f32x4 *__restrict__ c, long n4) {
    const long stride = (long)gridDim.x * 256;
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += U * stride) {
        f32x4 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + u * stride;
            if (i < n4) {
                va[u] = NT_LD ? __builtin_nontemporal_load(a + i) : a[i];
                if (TWO_IN) vb[u] = NT_LD ? __builtin_nontemporal_load(b + i) : b[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + u * stride;
            if (i < n4) {
                f32x4 r = va[u];
                if (TWO_IN) r = r * vb[u] + vb[u];
                if (NT_ST) __builtin_nontemporal_store(r, c + i);
                else c[i] = r;
            }
        }
    }
}

// contiguous chunk per workgroup instead of a grid-stride walk
template <int U, bool TWO_IN>
__global__ __launch_bounds__(256) void chunk_kernel(const f32x4 *__restrict__ a, const f32x4 *__restrict__ b,
                                                    f32x4 *__restrict__ c, long n4) {
    const long per = (n4 + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
    for (long i0 = lo + threadIdx.x; i0 < hi; i0 += U * 256) {
        f32x4 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + u * 256;
            if (i < hi) {
                va[u] = a[i];
                if (TWO_IN) vb[u] = b[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + u * 256;
            if (i < hi) {
                f32x4 r = va[u];
                if (TWO_IN) r = r * vb[u] + vb[u];
                c[i] = r;
            }
        }
    }
}

template <typename F>
static float time_us(F f, int reps = 10) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    f();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    const long n = 200704L * 256;          // 28 x 28 x 256 channels at B = 256: 205 MB per array
    const long n4 = n / 4;
    f32x4 *a, *b, *c;
    hipMalloc(&a, n * 4);
    hipMalloc(&b, n * 4);
    hipMalloc(&c, n * 4);
    hipMemset(a, 0, n * 4);
    hipMemset(b, 0, n * 4);
    const int bpcs[] = {2, 4, 8, 16, 32};
    printf("%-44s", "variant \\ blocks per CU");
    for (int bp : bpcs) printf(" %7d", bp);
    printf("   (TB/s)\n");
#define ROW(name, bytes, KERNEL)                                                             \
    do {                                                                                     \
        printf("%-44s", name);                                                               \
        for (int bp : bpcs) {                                                                \
            const int grid = 256 * bp;                                                       \
            const float us = time_us([&] { hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(256), 0, 0, a, b, c, n4); }); \
            printf(" %7.2f", (bytes) * (double)n / us / 1e6);                                \
        }                                                                                    \
        printf("\n");                                                                        \
    } while (0)
    ROW("copy U=1", 8, (stream_kernel<1, false, false, false>));
    ROW("copy U=2", 8, (stream_kernel<2, false, false, false>));
    ROW("copy U=4", 8, (stream_kernel<4, false, false, false>));
    ROW("copy U=4 nt loads", 8, (stream_kernel<4, true, false, false>));
    ROW("copy U=4 nt stores", 8, (stream_kernel<4, false, true, false>));
    ROW("copy U=4 nt both", 8, (stream_kernel<4, true, true, false>));
    ROW("copy chunked U=4", 8, (chunk_kernel<4, false>));
    ROW("2 in 1 out U=1", 12, (stream_kernel<1, false, false, true>));
    ROW("2 in 1 out U=2", 12, (stream_kernel<2, false, false, true>));
    ROW("2 in 1 out U=4", 12, (stream_kernel<4, false, false, true>));
    ROW("2 in 1 out U=4 nt loads", 12, (stream_kernel<4, true, false, true>));
    ROW("2 in 1 out U=4 nt stores", 12, (stream_kernel<4, false, true, true>));
    ROW("2 in 1 out U=4 nt both", 12, (stream_kernel<4, true, true, true>));
    ROW("2 in 1 out U=2 nt both", 12, (stream_kernel<2, true, true, true>));
    ROW("2 in 1 out chunked U=4", 12, (chunk_kernel<4, true>));
    {   // in place (c = a): the ds_bn_bwd_apply call of the engine writes dz over z
        printf("%-44s", "2 in, out over the first input, U=2");
        for (int bp : bpcs) {
            const float us = time_us([&] { hipLaunchKernelGGL((stream_kernel<2, false, false, true>), dim3(256 * bp), dim3(256), 0, 0, a, b, a, n4); });
            printf(" %7.2f", 12.0 * n / us / 1e6);
        }
        printf("\n");
        printf("%-44s", "2 in, out over the first input, U=2 nt both");
        for (int bp : bpcs) {
            const float us = time_us([&] { hipLaunchKernelGGL((stream_kernel<2, true, true, true>), dim3(256 * bp), dim3(256), 0, 0, a, b, a, n4); });
            printf(" %7.2f", 12.0 * n / us / 1e6);
        }
        printf("\n");
    }
    const float us = time_us([&] { hipMemcpyAsync(c, a, n * 4, hipMemcpyDeviceToDevice, 0); });
    printf("hipMemcpyAsync d2d: %.2f TB/s\n", 8.0 * n / us / 1e6);
    return 0;
}

// What limits the embedding gather's store stream?  1.26 GB written as (a) a linear fill, 16 B per lane; (b) 1200-byte
// rows, one wave per row (64 + 11 lanes), rows in linear order; each with plain and with non-temporal stores; (c) the
// same rows READ from a 12 MB table (uniform random ids) -- plain / non-temporal.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT> __global__ void fill_lin(f4 *out, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f4 v = {1.f, 2.f, 3.f, 4.f};
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
template <bool NT, bool READ, bool PERM = false> __global__ void rows(const float *table, const long *ids, float *out, long nrows, int D, int rpw, const long *perm = nullptr) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    long r = wave * rpw; const long r1 = r + rpw < nrows ? r + rpw : nrows;
    const int DV = D / 4;
    for (; r < r1; r += 4) {
        const float *src[4]; float *dst[4];
        for (int u = 0; u < 4; ++u) { long id = READ ? ids[r + u] : 0; src[u] = table + id * D; dst[u] = out + (PERM ? perm[r + u] : r + u) * D; }
        for (int v0 = 0; v0 < DV; v0 += 64) {
            const int v = v0 + lane; f4 val[4];
            for (int u = 0; u < 4; ++u) { val[u] = f4{1.f, 2.f, 3.f, 4.f}; if (READ && v < DV) val[u] = *(const f4 *)(src[u] + v * 4); }
            for (int u = 0; u < 4; ++u) if (v < DV) { if (NT) __builtin_nontemporal_store(val[u], (f4 *)(dst[u] + v * 4)); else *(f4 *)(dst[u] + v * 4) = val[u]; }
        }
    }
}
#define TIME(name, launch) { for (int i = 0; i < 3; ++i) { launch; } hipEventRecord(e0); for (int i = 0; i < 20; ++i) { launch; } hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-44s %7.1f us  %.2f TB/s written\n", name, ms * 50.f, bytes / (ms / 20 * 1e-3) / 1e12); }
int main() {
    const long nrows = 1 << 20; const int D = 300, V = 10001; const double bytes = (double)nrows * D * 4;
    float *out, *table; long *ids; hipMalloc(&out, nrows * D * 4); hipMalloc(&table, (long)V * D * 4); hipMalloc(&ids, nrows * 8);
    long *h = (long *)malloc(nrows * 8); srand(1); for (long i = 0; i < nrows; ++i) h[i] = rand() % V; hipMemcpy(ids, h, nrows * 8, hipMemcpyHostToDevice);
    hipMemset(table, 0, (long)V * D * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long n4 = nrows * D / 4; const int rpw = 16; const int grid = (int)((nrows / rpw + 3) / 4);
    TIME("linear fill, plain stores", (fill_lin<false><<<2048, 256>>>((f4 *)out, n4)));
    TIME("linear fill, non-temporal stores", (fill_lin<true><<<2048, 256>>>((f4 *)out, n4)));
    TIME("1200-byte rows, wave per row, plain", (rows<false, false><<<grid, 256>>>(table, ids, out, nrows, D, rpw)));
    TIME("1200-byte rows, wave per row, non-temporal", (rows<true, false><<<grid, 256>>>(table, ids, out, nrows, D, rpw)));
    TIME("gather (12 MB table, uniform ids), plain", (rows<false, true><<<grid, 256>>>(table, ids, out, nrows, D, rpw)));
    TIME("gather (12 MB table, uniform ids), non-temporal", (rows<true, true><<<grid, 256>>>(table, ids, out, nrows, D, rpw)));
    // (d) random OUTPUT order, no reads / L2-resident reads; (e) small (L2-resident) table, linear output; (f) sorted ids
    long *perm; hipMalloc(&perm, nrows * 8); for (long i = 0; i < nrows; ++i) h[i] = i;
    for (long i = nrows - 1; i > 0; --i) { long j = ((long)rand() * 32768 + rand()) % (i + 1); long t = h[i]; h[i] = h[j]; h[j] = t; }
    hipMemcpy(perm, h, nrows * 8, hipMemcpyHostToDevice);
    TIME("rows, no reads, RANDOM output rows, plain", (rows<false, false, true><<<grid, 256>>>(table, ids, out, nrows, D, rpw, perm)));
    TIME("rows, no reads, RANDOM output rows, non-temporal", (rows<true, false, true><<<grid, 256>>>(table, ids, out, nrows, D, rpw, perm)));
    long *ids_small; hipMalloc(&ids_small, nrows * 8); for (long i = 0; i < nrows; ++i) h[i] = rand() % 1024; hipMemcpy(ids_small, h, nrows * 8, hipMemcpyHostToDevice);
    TIME("gather, 1.2 MB table (L2 resident), linear out, plain", (rows<false, true><<<grid, 256>>>(table, ids_small, out, nrows, D, rpw)));
    TIME("gather, 1.2 MB table (L2 resident), linear out, NT", (rows<true, true><<<grid, 256>>>(table, ids_small, out, nrows, D, rpw)));
    TIME("gather, 1.2 MB table, RANDOM out, plain", (rows<false, true, true><<<grid, 256>>>(table, ids_small, out, nrows, D, rpw, perm)));
    for (long i = 0; i < nrows; ++i) h[i] = i * 10001 / nrows; hipMemcpy(ids_small, h, nrows * 8, hipMemcpyHostToDevice);
    TIME("gather, 12 MB table, SORTED ids, linear out, plain", (rows<false, true><<<grid, 256>>>(table, ids_small, out, nrows, D, rpw)));
    TIME("gather, 12 MB table, SORTED ids, RANDOM out, plain", (rows<false, true, true><<<grid, 256>>>(table, ids_small, out, nrows, D, rpw, perm)));
    TIME("gather, 12 MB table, SORTED ids, RANDOM out, NT", (rows<true, true, true><<<grid, 256>>>(table, ids_small, out, nrows, D, rpw, perm)));
    return 0;
}

// Semantics probe for buffer_load_dwordx4 ... lds on gfx950: lane -> LDS placement, OOB handling.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *x, float *y, unsigned bytes, int n) {
    __shared__ __attribute__((aligned(16))) float s[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) s[i] = -7.f;      // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (int)bytes, 0x00020000);
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    unsigned off = (unsigned)(tid ^ 1) * 16u;             // source permutation
    if (tid >= n) off = 0x80000000u;                       // OOB -> zeros or untouched?
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(s + wave * 256), 16, off, 0, 0, 0);
    __syncthreads();
    f32x4 v = *reinterpret_cast<f32x4 *>(s + tid * 4);
    *reinterpret_cast<f32x4 *>(y + tid * 4) = v;
}
int main() {
    float *x, *y; float hx[1024], hy[1024];
    for (int i = 0; i < 1024; ++i) hx[i] = (float)i;
    hipMalloc(&x, 4096); hipMalloc(&y, 4096);
    hipMemcpy(x, hx, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, x, y, 4096u, 200);
    hipMemcpy(hy, y, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) for (int j = 0; j < 4; ++j) {
        float want = t < 200 ? (float)((t ^ 1) * 4 + j) : 0.f;
        if (hy[t * 4 + j] != want) { if (bad < 8) printf("t=%d j=%d got %g want %g\n", t, j, hy[t*4+j], want); ++bad; }
    }
    printf("mismatches: %d (OOB lanes read back as %g)\n", bad, hy[201 * 4]);
    return 0;
}

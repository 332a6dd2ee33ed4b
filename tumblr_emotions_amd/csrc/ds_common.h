// Shared helpers for the gfx950 kernel library (internal; the public ABI is include/ds_kernels.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "ds_kernels.h"

namespace ds {

void set_error(const char *fmt, ...);

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return DS_ERR_LAUNCH;
    }
    return DS_OK;
}

// MI355X: 256 CUs in 8 XCDs.  Memory-bound kernels cap their grid at 8 blocks per CU and
// grid-stride the rest (cdna_hip_programming.md Guideline 11).
constexpr int kCUs = 256;
constexpr int kMaxStreamBlocks = kCUs * 8;

inline int stream_grid(int64_t work_items, int per_block) {
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > kMaxStreamBlocks) b = kMaxStreamBlocks;
    return (int)b;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

}  // namespace ds

#define DS_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            ds::set_error(__VA_ARGS__);       \
            return DS_ERR_ARG;                \
        }                                     \
    } while (0)

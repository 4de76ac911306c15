// common.h -- shared helpers for the gfx950 hot-path library (error plumbing, launch checks).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/iqa_hotpath.h"

namespace iqa {

constexpr int kWave = 64;  // CDNA wavefront width

void set_error(const char *fmt, ...);

inline int fail_inval(const char *what)
{
    set_error("invalid argument: %s", what);
    return IQA_EINVAL;
}

inline int check_launch(const char *kernel)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", kernel, hipGetErrorString(e));
        return IQA_EHIP;
    }
    return IQA_OK;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int frame_bytes(int fmt)
{
    switch (fmt) {
        case IQA_FMT_S16: return 4;
        case IQA_FMT_U8: return 2;
        case IQA_FMT_F32: return 8;
        default: return 0;
    }
}

// 64-lane butterfly sum (every lane ends with the total).
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
    return v;
}

}  // namespace iqa

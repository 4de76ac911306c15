// scan_common.h -- affine-map scan helpers shared by demod.hip and demod_fused.hip.
#pragma once
#include "common.h"

namespace iqa {

constexpr int SC_THREADS = 256;
constexpr int SC_ITEMS = 8;
constexpr int SC_TILE = SC_THREADS * SC_ITEMS;  // 2048 elements per block

struct Aff {  // s -> A*s + B
    double A, B;
};
// apply `l` first, then `r`
__device__ __forceinline__ Aff then(const Aff &l, const Aff &r) { return Aff{r.A * l.A, fma(r.A, l.B, r.B)}; }

// inclusive ordered wave scan; returns the inclusive prefix for this lane
__device__ __forceinline__ Aff wave_inclusive(Aff v, int lane)
{
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const double la = __shfl_up(v.A, o, kWave);
        const double lb = __shfl_up(v.B, o, kWave);
        if (lane >= o) v = then(Aff{la, lb}, v);
    }
    return v;
}

// first index in sorted `arr[0..n)` that is >= v
__device__ __forceinline__ long long lower_bound_ll(const long long *arr, long long n, long long v)
{
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (arr[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

}  // namespace iqa

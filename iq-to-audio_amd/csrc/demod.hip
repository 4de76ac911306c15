// demod.hip -- channel-rate demodulator kernels for gfx950.
//
// Replaces (reference src/iq_to_audio/):
//   decoders/nfm.py:17-24   QuadratureDemod.process      -> k_quadrature
//   decoders/nfm.py:48-62   DeemphasisFilter.process     -> affine scan, OP_DEEMPH
//   decoders/common.py:16-30 DCBlocker.process           -> affine scan, OP_DC
//   decoders/ssb.py:65-80   SSBDecoder._apply_agc        -> affine scan, OP_AGC (segmented)
//   decoders/am.py:28       |z|                          -> k_envelope
//   decoders/ssb.py:42-43   real(z)                      -> k_real
//   processing.py:440-456   AudioWriter.write            -> k_writer_clip
//   processing.py:1105,650-658 mean |z|^2                -> k_mean_power
//
// The three IIR stages are per-sample Python/C loops in the reference.  Each is a
// first-order affine recurrence s[n] = a[n]*s[n-1] + b[n]; affine maps compose
// associatively, so they run here as a 3-launch block scan (reduce -> carry -> apply)
// in float64.  The AGC's "gain restarts at 1.0 on every process() call" becomes a
// segmented scan: at a restart index the element's map is the constant a+b.
#include <cstdlib>
#include "scan_common.h"

namespace iqa {

enum ScanOp { OP_DEEMPH = 0, OP_DC = 1, OP_AGC = 2 };

struct ScanArgs {
    const float *x;
    float *y;
    long long n;
    double p0, p1;            // DEEMPH: alpha, 1-alpha | DC: radius(f32-rounded), - | AGC: target(f32), decay(f32)
    const double *state;      // DEEMPH: {y_last} | DC: {x_last, y_last} | AGC: unused
    const long long *resets;  // AGC: sorted restart indices (may be NULL -> only index 0)
    long long n_resets;
    Aff *agg;       // [nblocks]
    double *carry;  // [nblocks] state entering each block
    double *fin;    // [1] state after the last element
    int nblocks;
};

__device__ __forceinline__ bool is_reset(const ScanArgs &a, long long idx)
{
    if (idx == 0) return true;
    if (a.resets == nullptr || a.n_resets <= 0) return false;
    long long lo = 0, hi = a.n_resets;  // first element >= idx
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (a.resets[mid] < idx) lo = mid + 1; else hi = mid;
    }
    return lo < a.n_resets && a.resets[lo] == idx;
}

// does [lo, hi) contain a restart index?  (block-level early out for the per-element search)
__device__ __forceinline__ bool any_reset_in(const ScanArgs &a, long long lo_i, long long hi_i)
{
    if (lo_i == 0) return true;
    if (a.resets == nullptr || a.n_resets <= 0) return false;
    long long lo = 0, hi = a.n_resets;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (a.resets[mid] < lo_i) lo = mid + 1; else hi = mid;
    }
    return lo < a.n_resets && a.resets[lo] < hi_i;
}

template <int OP>
__device__ __forceinline__ Aff element_map(const ScanArgs &a, long long idx, float x, float x_prev, bool maybe_reset)
{
    if constexpr (OP == OP_DEEMPH) {
        return Aff{a.p0, a.p1 * static_cast<double>(x)};
    } else if constexpr (OP == OP_DC) {
        const float d = x - x_prev;  // float32 difference, as the reference forms it
        return Aff{a.p0, static_cast<double>(d)};
    } else {
        const float mag = fabsf(x);
        Aff m{1.0, 0.0};
        if (mag > 1e-6f) {
            const float desired = static_cast<float>(a.p0) / mag;
            const double decay = a.p1;
            m = Aff{1.0 - decay, decay * static_cast<double>(desired)};
        }
        if (maybe_reset && is_reset(a, idx)) m = Aff{0.0, m.A + m.B};  // gain restarts at 1.0
        return m;
    }
}

template <int OP>
__device__ __forceinline__ void load_items(const ScanArgs &a, long long base, float (&x)[SC_ITEMS], float &x_before)
{
#pragma unroll
    for (int i = 0; i < SC_ITEMS; ++i) x[i] = (base + i < a.n) ? a.x[base + i] : 0.f;
    x_before = 0.f;
    if constexpr (OP == OP_DC) {
        if (base == 0) x_before = static_cast<float>(a.state[0]);
        else if (base - 1 < a.n) x_before = a.x[base - 1];
    }
}

template <int OP>
__global__ __launch_bounds__(SC_THREADS) void k_scan_reduce(ScanArgs a)
{
    __shared__ Aff s_w[SC_THREADS / kWave];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long blk0 = static_cast<long long>(blockIdx.x) * SC_TILE;
    const long long base = blk0 + static_cast<long long>(tid) * SC_ITEMS;
    const bool maybe_reset = (OP == OP_AGC) && any_reset_in(a, blk0, blk0 + SC_TILE);
    float x[SC_ITEMS], xb;
    load_items<OP>(a, base, x, xb);
    Aff t{1.0, 0.0};
#pragma unroll
    for (int i = 0; i < SC_ITEMS; ++i) {
        if (base + i < a.n) t = then(t, element_map<OP>(a, base + i, x[i], i ? x[i - 1] : xb, maybe_reset));
    }
    const Aff inc = wave_inclusive(t, lane);
    if (lane == kWave - 1) s_w[wave] = inc;
    __syncthreads();
    if (tid == 0) {
        Aff tot = s_w[0];
#pragma unroll
        for (int w = 1; w < SC_THREADS / kWave; ++w) tot = then(tot, s_w[w]);
        a.agg[blockIdx.x] = tot;
    }
}

template <int OP>
__global__ __launch_bounds__(kWave) void k_scan_carry(ScanArgs a)
{
    const int lane = threadIdx.x;
    double s;
    if constexpr (OP == OP_DEEMPH) s = a.state[0];
    else if constexpr (OP == OP_DC) s = a.state[1];
    else s = 1.0;
    for (int c = 0; c < a.nblocks; c += kWave) {
        const int b = c + lane;
        Aff v = (b < a.nblocks) ? a.agg[b] : Aff{1.0, 0.0};
        const Aff inc = wave_inclusive(v, lane);
        const double after = fma(inc.A, s, inc.B);
        double before = __shfl_up(after, 1, kWave);
        if (lane == 0) before = s;
        if (b < a.nblocks) a.carry[b] = before;
        s = __shfl(after, kWave - 1, kWave);
    }
    if (lane == 0) a.fin[0] = s;
}

template <int OP>
__global__ __launch_bounds__(SC_THREADS) void k_scan_apply(ScanArgs a)
{
    __shared__ Aff s_w[SC_THREADS / kWave];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long blk0 = static_cast<long long>(blockIdx.x) * SC_TILE;
    const long long base = blk0 + static_cast<long long>(tid) * SC_ITEMS;
    const bool maybe_reset = (OP == OP_AGC) && any_reset_in(a, blk0, blk0 + SC_TILE);
    float x[SC_ITEMS], xb;
    load_items<OP>(a, base, x, xb);
    Aff m[SC_ITEMS];
    Aff t{1.0, 0.0};
#pragma unroll
    for (int i = 0; i < SC_ITEMS; ++i) {
        m[i] = (base + i < a.n) ? element_map<OP>(a, base + i, x[i], i ? x[i - 1] : xb, maybe_reset) : Aff{1.0, 0.0};
        t = then(t, m[i]);
    }
    const Aff inc = wave_inclusive(t, lane);
    if (lane == kWave - 1) s_w[wave] = inc;
    __syncthreads();
    // exclusive prefix of this thread inside the block
    double ea = __shfl_up(inc.A, 1, kWave), eb = __shfl_up(inc.B, 1, kWave);
    Aff ex = (lane == 0) ? Aff{1.0, 0.0} : Aff{ea, eb};
    Aff wpre{1.0, 0.0};
    for (int w = 0; w < wave; ++w) wpre = then(wpre, s_w[w]);
    ex = then(wpre, ex);
    double s = fma(ex.A, a.carry[blockIdx.x], ex.B);
#pragma unroll
    for (int i = 0; i < SC_ITEMS; ++i) {
        s = fma(m[i].A, s, m[i].B);
        if (base + i < a.n) {
            if constexpr (OP == OP_AGC) a.y[base + i] = x[i] * static_cast<float>(s);
            else a.y[base + i] = static_cast<float>(s);
        }
    }
}

// runs after k_scan_apply (which still needs the OLD x_last for element 0)
template <int OP>
__global__ void k_scan_finish(ScanArgs a, double *state_out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if constexpr (OP == OP_DEEMPH) {
        state_out[0] = a.fin[0];
    } else if constexpr (OP == OP_DC) {
        state_out[0] = static_cast<double>(a.x[a.n - 1]);
        state_out[1] = a.fin[0];
    }
}

// ---- elementwise ------------------------------------------------------------------------------

__global__ void k_quadrature(const float2 *z, long long n, const float2 *prev, float *out)
{
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 c = z[i];
    const float2 p = (i == 0) ? prev[0] : z[i - 1];
    // z * conj(p), float32 complex product as numpy forms it
    const float re = c.x * p.x + c.y * p.y;
    const float im = c.y * p.x - c.x * p.y;
    out[i] = atan2f(im, re);
}

__global__ void k_store_last(const float2 *z, long long n, float2 *prev)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) prev[0] = z[n - 1];
}

__global__ void k_envelope(const float2 *z, long long n, float *out)
{
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) out[i] = hypotf(z[i].x, z[i].y);
}

__global__ void k_real(const float2 *z, long long n, float *out)
{
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) out[i] = z[i].x;
}

__global__ void k_decimate(const float2 *in, long long first, int D, float2 *out, long long n_out)
{
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n_out) out[i] = in[first + i * D];
}

__global__ __launch_bounds__(256) void k_mean_power(const float2 *z, long long n, long long skip, double inv_count,
                                                    double *out)
{
    __shared__ double s_w[4];
    double acc = 0.0;
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = skip + static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float2 v = z[i];
        const float m = hypotf(v.x, v.y);  // np.abs(complex64) -> float32, then **2 in float32
        acc += static_cast<double>(m * m);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (s_w[0] + s_w[1] + s_w[2] + s_w[3]) * inv_count);
}

// Short inputs (the mixer-sign probes: a few thousand samples): one block, fixed summation order, the result is
// WRITTEN -- no memset in front, no atomics.
// (a grid of several blocks: block p reduces the p-th stretch of `stride` samples into out[p] -- the two mixer-sign probes
// of a capture in one launch)
__global__ __launch_bounds__(1024) void k_mean_power_small(const float2 *z, long long n, long long skip, double inv_count,
                                                            double *out, long long stride)
{
    __shared__ double s_w[16];
    z += blockIdx.x * stride;
    out += blockIdx.x;
    double acc = 0.0;
    for (long long i = skip + threadIdx.x; i < n; i += 1024) {
        const float2 v = z[i];
        const float m = hypotf(v.x, v.y);
        acc += static_cast<double>(m * m);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += s_w[w];
        out[0] = t * inv_count;
    }
}

// Wideband level of raw capture frames (the precision guard's reference level, reference has no counterpart: its
// channel filter runs in complex128 and has no level dependence, processing.py:300-346): mean square of the VALUES
// (I and Q alike; uint8 minus 128), estimated from eight stretches of 1024 consecutive 16-byte vectors spread evenly over the range.
// One workgroup, fixed order, the result WRITTEN (mapped pinned host memory is fine).
template <int FMT>
__global__ __launch_bounds__(1024) void k_raw_level(const uint4 *raw, long long n_vec, long long step, double *out)
{
    constexpr int ROUNDS = 8;
    __shared__ double s_w[16];
    double acc = 0.0;
    long long cnt = 0;
    // all loads first (independent cache lines, far apart: one round trip instead of ROUNDS), then the arithmetic
    uint4 v[ROUNDS];
    bool ok[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        // round r: 1024 CONSECUTIVE vectors (16 KiB: one coalesced wave-load per wave) from the r-th of ROUNDS stretches
        // spread evenly over the range -- scattered single vectors cost a DRAM page each (measured 34 us for this kernel)
        const long long i = static_cast<long long>(r) * step + threadIdx.x;
        ok[r] = i < n_vec;
        v[r] = ok[r] ? raw[i] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (!ok[r]) continue;
        const unsigned w[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (FMT == IQA_FMT_S16) {
                const float a = static_cast<float>(static_cast<short>(w[k] & 0xFFFF)), b = static_cast<float>(static_cast<short>(w[k] >> 16));
                acc += static_cast<double>(a * a + b * b);
            } else if (FMT == IQA_FMT_U8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = static_cast<float>(static_cast<int>((w[k] >> (8 * j)) & 0xFF) - 128);
                    acc += static_cast<double>(a * a);
                }
            } else {
                const float a = __uint_as_float(w[k]);
                acc += static_cast<double>(a) * static_cast<double>(a);
            }
        }
        cnt += FMT == IQA_FMT_S16 ? 8 : FMT == IQA_FMT_U8 ? 16 : 4;
    }
    acc = wave_sum(acc);
    double c = wave_sum(static_cast<double>(cnt));
    __shared__ double s_c[16];
    if ((threadIdx.x & 63) == 0) {
        s_w[threadIdx.x >> 6] = acc;
        s_c[threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0, n = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            t += s_w[w];
            n += s_c[w];
        }
        out[0] = n > 0.0 ? t / n : 0.0;
    }
}

// AudioWriter.write: running pre-clip peak, clip, per-segment sum of squares (float64).
// One float64 atomic per block when the block lies inside one segment (the common case: a
// reference chunk is >= 40k channel-rate samples); per-thread flushes only for blocks that
// straddle a segment boundary.
__global__ __launch_bounds__(256) void k_writer_clip(const float *a, long long n, unsigned int *peak_bits,
                                                     const long long *seg_starts, long long n_segs, double *sumsq,
                                                     float *out)
{
    __shared__ float s_pk[4];
    __shared__ double s_sq[4];
    __shared__ long long s_seg[2];
    constexpr int ITEMS = 4;
    const long long blk0 = static_cast<long long>(blockIdx.x) * blockDim.x * ITEMS;
    const long long base = blk0 + static_cast<long long>(threadIdx.x) * ITEMS;
    const bool stats = (sumsq != nullptr && n_segs > 0);
    if (stats && threadIdx.x < 2) {
        // segment of the block's first / last element: last start <= idx
        const long long idx = threadIdx.x == 0 ? blk0 : min(blk0 + static_cast<long long>(blockDim.x) * ITEMS, n) - 1;
        long long lo = 0, hi = n_segs;
        while (hi - lo > 1) {
            const long long mid = (lo + hi) >> 1;
            if (seg_starts[mid] <= idx) lo = mid; else hi = mid;
        }
        s_seg[threadIdx.x] = lo;
    }
    __syncthreads();
    const bool uniform = stats && (s_seg[0] == s_seg[1]);
    float pk = 0.f;
    double run = 0.0;
    long long seg = stats ? s_seg[0] : -1;
    if (stats && !uniform && base < n) {
        long long lo = s_seg[0], hi = s_seg[1] + 1;
        while (hi - lo > 1) {
            const long long mid = (lo + hi) >> 1;
            if (seg_starts[mid] <= base) lo = mid; else hi = mid;
        }
        seg = lo;
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const long long idx = base + i;
        if (idx >= n) break;
        const float v = a[idx];
        pk = fmaxf(pk, fabsf(v));
        if (out != nullptr) out[idx] = fminf(fmaxf(v, -0.99f), 0.99f);
        if (stats) {
            if (!uniform) {
                while (seg + 1 < n_segs && seg_starts[seg + 1] <= idx) {
                    if (run != 0.0) atomicAdd(&sumsq[seg * IQA_SUMSQ_SLOTS + (blockIdx.x & (IQA_SUMSQ_SLOTS - 1))], run);
                    run = 0.0;
                    ++seg;
                }
            }
            run += static_cast<double>(v) * static_cast<double>(v);
        }
    }
    if (stats && !uniform && run != 0.0) atomicAdd(&sumsq[seg * IQA_SUMSQ_SLOTS + (blockIdx.x & (IQA_SUMSQ_SLOTS - 1))], run);
    pk = wave_max(pk);
    const double wsq = uniform ? wave_sum(run) : 0.0;
    if ((threadIdx.x & 63) == 0) {
        s_pk[threadIdx.x >> 6] = pk;
        s_sq[threadIdx.x >> 6] = wsq;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (peak_bits != nullptr) {
            // non-negative floats order like their bit patterns; skip atomics that cannot raise the maximum
            const unsigned int m = __float_as_uint(fmaxf(fmaxf(s_pk[0], s_pk[1]), fmaxf(s_pk[2], s_pk[3])));
            if (m > __hip_atomic_load(peak_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(peak_bits, m);
        }
        if (uniform) atomicAdd(&sumsq[s_seg[0] * IQA_SUMSQ_SLOTS + (blockIdx.x & (IQA_SUMSQ_SLOTS - 1))], s_sq[0] + s_sq[1] + s_sq[2] + s_sq[3]);
    }
}

// -acodec pcm_s16le: round-half-even(y * 32768), saturated
__device__ __forceinline__ short pcm16_of(float y)
{
    const double v = rint(static_cast<double>(y) * 32768.0);
    return static_cast<short>(fmin(fmax(v, -32768.0), 32767.0));
}

__global__ void k_float_to_pcm16(const float *y, long long n, short *pcm)
{
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pcm[i] = pcm16_of(y[i]);
}

// 48 kHz polyphase resampler, float64 accumulate.  Row-stationary: outputs J and J+up use the same polyphase
// row (phase p = (J*down) mod up), so the table is read once per `split` part.  A wave owns SIXTEEN CONSECUTIVE
// output residues (rows p, p+inc, p+2inc, ... with inc = down mod up), one per group of four lanes, which keeps
// that row's taps in registers (lane `sub` holds taps sub, sub+4, ...).  Per step g the wave produces the 16
// consecutive outputs J0+16w .. +15 (+ g*up): their input windows overlap almost entirely (16 outputs x 67 taps
// touch ~100 inputs), so the wave stages that stretch of the input ONCE in LDS and every lane picks its taps' samples
// from there.
//
// What the kernel waited for in its first form (46.5 us at config 2, two thirds of it idle) was the memory system's
// LATENCY, twice per step: consecutive steps of a wave lie `down` inputs apart (192 KB at config 2), so every step's
// window is a fresh L2/MALL access, and it was fetched only one step ahead; and the step's (branched) store made hipcc
// wait for vmcnt(0) in front of the next step's window -- i.e. for the store's acknowledgement.  Now the windows of the
// next RS_AHEAD steps are in flight at any time: LDS-DMA (buffer_load_dwordx4 ... lds, ONE instruction per window,
// range-checked: positions outside [0, n_in) arrive as zeros) into a ring of RS_RING windows per wave, no registers, and
// ONE counted wait per step that leaves the younger windows and the stores of the groups of steps behind them outstanding.
// For that count to be exact every group of steps issues the same number of stores: unconditional buffer stores whose
// offset is out of range in lanes that have nothing to write, and the prologue pads with dropped stores; the number of
// younger windows is wave-uniform (it shrinks over a wave's last RS_AHEAD steps) and selects the wait.
//
// Steps go in groups of RS_GROUP = 4: the four sums of a group stay in registers (per lane: its taps' share of each) and
// are reduced ACROSS the quad in one transposing pass -- lane k of the quad ends up with the whole sum of step k -- so
// the rounding to float32, the PCM16 conversion and the store run once per group in all lanes instead of once per step
// in one lane of four, and the four steps' chains of float64 FMAs are independent of each other.  The order of the
// additions per output is the one of the first form: ((lane 0 + lane 2) + (lane 1 + lane 3)) of ((a0+a1)+(a2+a3)).

// value of lane ^ 2 (CTRL 0x4E) / lane ^ 1 (0xB1) within each quad: DPP quad permutes, register to register
// (__shfl_xor goes through ds_bpermute and its LDS latency)
template <int CTRL>
__device__ __forceinline__ double quad_swap(double v)
{
    return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true),
                            __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true));
}

// Sum over the four lanes of a quad, in every lane.
__device__ __forceinline__ double quad_sum(double v)
{
    v += quad_swap<0x4E>(v);
    v += quad_swap<0xB1>(v);
    return v;
}

// pcm16_of on float32 alone: y * 32768 is exact in float32 (a power of two), so rounding it there is rounding the
// float64 product; +-inf and NaN end where pcm16_of sends them.
__device__ __forceinline__ short pcm16_of_f32(float y)
{
    const float v = rintf(y * 32768.0f);
    return static_cast<short>(static_cast<int>(fminf(fmaxf(v, -32768.0f), 32767.0f)));
}

#ifndef IQA_RS_TARGET_WAVES
#define IQA_RS_TARGET_WAVES 8192
#endif
#ifndef IQA_RS_ABLATE
#define IQA_RS_ABLATE 0  // timing experiments only: 1 no stores, 2 no window DMAs, 4 no window reads / FMAs, 8 no table loads
#endif
constexpr int RS_WAVES = 4;
constexpr int RS_LANES = 4;    // lanes per output
constexpr int RS_GROUP = 4;    // steps whose sums are reduced and stored together (= RS_LANES: one per lane of the quad)
#ifndef IQA_RS_AHEAD
#define IQA_RS_AHEAD 8
#endif
constexpr int RS_AHEAD = IQA_RS_AHEAD;  // windows in flight behind the one being read, whole groups of steps (measured at
                                         // config 2: 4 -> 28.7 us, 8 -> 29.7, 12 -> 30.9, 16 -> 33.1: not the latency any more)
constexpr int RS_RING = RS_AHEAD + 2;  // + the one being read + the one read a step ago (its reads may still be in the LDS queue)
constexpr unsigned int RS_NOWHERE = 0x80000000u;  // buffer offset beyond every descriptor used here: the access is dropped
static_assert(RS_AHEAD % RS_GROUP == 0 && RS_GROUP == RS_LANES, "the counted wait assumes whole groups of stores per RS_AHEAD steps");
constexpr int RS_GROUPS_AHEAD = RS_AHEAD / RS_GROUP;

typedef __attribute__((address_space(3))) void rs_lds_t;

template <int NI>
struct RsGeo {
    // largest spread of input positions inside a wave that the staged window covers: the 16 outputs of a step lie
    // 15 down/up inputs apart and a row has ~32 down/up taps, i.e. ~1.9 NI -- waves with more (a ratio that outgrows the
    // one-instruction window, or residues that wrap around `up`) read their samples straight from memory
    static constexpr int SPREAD = (2 * NI + 2 < 253 - 4 * NI) ? 2 * NI + 2 : 253 - 4 * NI;
    // a window starts at a multiple of 4 samples (16-byte DMA lanes that never straddle x[0]): up to 3 samples of slack
    static constexpr int CHUNKS = (4 * NI + SPREAD + 3 + 63) / 64;  // 64-float quarters of the one DMA
    static constexpr int WINDOW = 64 * CHUNKS;                      // floats
    static_assert(SPREAD >= 16 && CHUNKS <= 4, "a window is at most 64 lanes x 16 bytes");
};

struct RsArgs {
    const float *x;
    long long n_in;
    const double *table;
    int up, down, T, split;
    long long n_out;
    float *y;
    short *pcm;
    // host-side quotients (the kernel would spend a third of its instructions on 64-bit divisions otherwise)
    long long g_all;   // ceil(n_out / up): steps any residue can need
    long long g_per;   // ceil(g_all / split): steps per wave
    int j0_mod_up;     // j0 mod up
    long long j0_q;    // floor(j0 * down / up)
    long long j0_r;    // (j0 * down) mod up
};

// floor(num / den) and the remainder for 0 <= num < 2^52, 0 < den < 2^31: one float64 division and a correction
__device__ __forceinline__ long long rs_divmod(long long num, int den, int &rem)
{
    long long q = static_cast<long long>(static_cast<double>(num) / static_cast<double>(den));
    long long r = num - q * den;
    if (r < 0) { --q; r += den; }
    else if (r >= den) { ++q; r -= den; }
    rem = static_cast<int>(r);
    return q;
}

// The few waves whose 16 residues straddle the wrap of (j0 + jj) mod up (their input positions lie far apart) and the
// one with residues >= up: per-lane loads, no staging.  Same sums in the same order as the staged loop.
template <int NI>
__device__ __forceinline__ void resample_unstaged(const float *x, long long n_in, const double (&h)[NI], long long q0, int down, int T,
                                                   int sub, long long jj0, int up, long long g_lo, long long g_hi, long long n_out,
                                                   float *y, short *pcm)
{
    for (long long g = g_lo; g < g_hi; ++g) {
        const long long jj = jj0 + g * up;
        const long long top = q0 + g * down + T - sub;  // input index of this lane's first tap; tap i reads top - 4 i
        float xv[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const long long nidx = top - RS_LANES * i;
            const float v = x[min(max(nidx, 0LL), max(n_in - 1, 0LL))];
            xv[i] = (nidx >= 0 && nidx < n_in) ? v : 0.f;
        }
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const double xd = static_cast<double>(xv[i]);
            if ((i & 3) == 0) a0 = fma(h[i], xd, a0);
            else if ((i & 3) == 1) a1 = fma(h[i], xd, a1);
            else if ((i & 3) == 2) a2 = fma(h[i], xd, a2);
            else a3 = fma(h[i], xd, a3);
        }
        const double acc = quad_sum((a0 + a1) + (a2 + a3));
        if (jj < n_out && sub == 0) {
            const float v = static_cast<float>(acc);
            if (y != nullptr) y[jj] = v;
            if (pcm != nullptr) pcm[jj] = pcm16_of_f32(v);
        }
    }
}

// One window = 64 CH consecutive floats in ONE 16-byte-per-lane DMA of the first 16 CH lanes (lane l: bytes
// [voff, voff + 16) -> dst + 16 l); the texture addresser's cost is per lane and instruction, not per byte.
template <int CH>
__device__ __forceinline__ void rs_dma(__amdgpu_buffer_rsrc_t src, rs_lds_t *dst, int voff, int lane)
{
    static_assert(CH >= 1 && CH <= 4, "a window is at most 64 lanes x 16 bytes");
    if (CH == 4 || lane < 16 * CH) __builtin_amdgcn_raw_ptr_buffer_load_lds(src, dst, 16, voff, 0, 0, 0);
}

__device__ __forceinline__ long long rs_uniform64(long long v)  // a value every lane holds, moved to scalar registers
{
    const int lo = __builtin_amdgcn_readfirstlane(static_cast<int>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<int>(v >> 32));
    return (static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo);
}

template <int NI, bool WANT_Y, bool WANT_PCM>  // NI: taps per lane = ceil(row_len / 4)
__global__ __launch_bounds__(RS_WAVES *kWave) void k_resample(RsArgs a)
{
    using G = RsGeo<NI>;
    constexpr int CH = G::CHUNKS, WIN = G::WINDOW, NS = (WANT_Y ? 1 : 0) + (WANT_PCM ? 1 : 0);
    // memory operations a wave issues behind a window's DMAs before it reads that window: the DMAs of the RS_AHEAD
    // younger windows and the stores of the RS_AHEAD / RS_GROUP groups that ended since
    static_assert(RS_AHEAD + RS_GROUPS_AHEAD * NS <= 63, "vmcnt holds six bits");
    static_assert(4 * NI + G::SPREAD + 3 <= WIN, "window too small for this row length");
    // (ONE LDS object on purpose: with two, hipcc tags their accesses with alias scopes and then drains vmcnt to 0 in
    // front of every read of the ring -- it knows the DMAs write there, not which of them)
    __shared__ float s_x[RS_WAVES][RS_RING][WIN];
    const int up = a.up, down = a.down, T = a.T;
    const int lane = threadIdx.x & 63, sub = lane & (RS_LANES - 1), slot = lane >> 2;
    const int wv = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const unsigned int wid = blockIdx.x * RS_WAVES + wv;
    const unsigned int w = wid / static_cast<unsigned int>(a.split);  // group of 16 residues of (j0 + jj) mod up handled by this wave
    const int part = static_cast<int>(wid - w * a.split);
    if (static_cast<long long>(w) * 16 >= up) return;
    const int row_len = 2 * T + 1;
    const long long g_lo = part * a.g_per, g_hi = min(a.g_all, g_lo + a.g_per);
    if (g_lo >= g_hi) return;
    // this quad's residue: first output jj0, its input position q0 and its row
    const int res = static_cast<int>(w) * 16 + slot;
    const bool ok = res < up;
    int jj0 = res - a.j0_mod_up;  // (res - j0 mod up) mod up
    if (jj0 < 0) jj0 += up;
    if (!ok) jj0 = 0;
    // (j0 + jj0) * down = (j0_q * up + j0_r) + jj0 * down
    int p;
    const long long q0 = a.j0_q + rs_divmod(a.j0_r + static_cast<long long>(jj0) * down, up, p);
    const double *row = a.table + static_cast<long long>(p) * row_len;
    // all loads unconditional (clamped index, masked afterwards): a branch around a load makes hipcc wait for
    // every load separately
    double h[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) h[i] = (IQA_RS_ABLATE & 8) ? 1e-3 * (p + i) : row[min(sub + RS_LANES * i, row_len - 1)];
#pragma unroll
    for (int i = 0; i < NI; ++i)
        if (sub + RS_LANES * i >= row_len) h[i] = 0.0;
    // the wave's window: positions [qmin + g*down + T - (4 NI - 1), qmax + g*down + T]; tap t = sub + 4 i of a lane
    // reads position q0 + g*down + T - t = window[(q0 - qmin) + 4 NI - 1 - t]
    long long qmin = ok ? q0 : (1LL << 62), qmax = ok ? q0 : -(1LL << 62);
#pragma unroll
    for (int m = 4; m < 64; m <<= 1) {
        qmin = min(qmin, __shfl_xor(qmin, m, kWave));
        qmax = max(qmax, __shfl_xor(qmax, m, kWave));
    }
    qmin = rs_uniform64(qmin);
    qmax = rs_uniform64(qmax);
    if (qmax - qmin > G::SPREAD) {  // (wave-uniform) only where the residues wrap around `up`
        resample_unstaged<NI>(a.x, a.n_in, h, q0, down, T, sub, ok ? jj0 : a.n_out, up, g_lo, g_hi, a.n_out, WANT_Y ? a.y : nullptr,
                              WANT_PCM ? a.pcm : nullptr);
        return;
    }
    float *ring = &s_x[wv][0][0];
    // this lane's LAST tap's sample in a window; tap i's lies 4 (NI - 1 - i) above it
    const int my_low = (ok ? static_cast<int>(q0 - qmin) : 0) + 3 - sub;
    const long long w_first = qmin + T - (4 * NI - 1);  // + g*down: first position of step g's window

    // input windows: one descriptor over x[0, n_in) -- positions outside it (the stream's edges) arrive as zeros --
    // and the same with no records for the DMAs past this wave's last step
    const int x_bytes = static_cast<int>(a.n_in * 4);
    long long dma_pos = w_first + g_lo * down, rd_pos = dma_pos;  // first position of the next window to request / to read
    const int steps = static_cast<int>(g_hi - g_lo);
    unsigned int lane16 = 16u * lane;
    int issued = 0;  // windows requested so far (wave-uniform)
    auto issue = [&](int into) {
        // past the wave's last step: no records -- dropped by the range check, still counted
        const __amdgpu_buffer_rsrc_t src =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, issued < steps ? x_bytes : 0, 0x00020000);
        // from the multiple of 4 samples at or below dma_pos (mod 2^32: a negative offset is out of range)
        const unsigned int from = static_cast<unsigned int>((dma_pos & ~3LL) * 4);
        if (!(IQA_RS_ABLATE & 2)) rs_dma<CH>(src, (rs_lds_t *)(ring + into * WIN), static_cast<int>(from + lane16), lane);
        dma_pos += down;
        ++issued;
    };
    // wait until the window of the current step has landed: behind its DMA the wave has issued the DMAs of the RS_AHEAD
    // younger windows and the stores of RS_AHEAD / RS_GROUP groups of steps
    auto wait_window = [&](int) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((IQA_RS_ABLATE & 2) ? 0 : RS_AHEAD + RS_GROUPS_AHEAD * NS) : "memory"); };
    // outputs: descriptors over this wave's stretch [j_base, j_base + (g_hi - g_lo) up) of y / pcm, cut at n_out -- the
    // range check is the `jj < n_out` test and the one for the steps past g_hi of the last group; lane k of a quad
    // writes the group's step k; quads without a residue aim at RS_NOWHERE
    const long long j_base = g_lo * up;
    const long long j_cnt = min(a.n_out - j_base, (g_hi - g_lo) * up);
    __amdgpu_buffer_rsrc_t y_dst, p_dst;
    if constexpr (WANT_Y) y_dst = __builtin_amdgcn_make_buffer_rsrc(a.y + j_base, 0, static_cast<int>(j_cnt * 4), 0x00020000);
    if constexpr (WANT_PCM) p_dst = __builtin_amdgcn_make_buffer_rsrc(a.pcm + j_base, 0, static_cast<int>(j_cnt * 2), 0x00020000);
    const unsigned int first = static_cast<unsigned int>(jj0) + static_cast<unsigned int>(sub) * static_cast<unsigned int>(up);
    unsigned int y_off = ok ? 4u * first : RS_NOWHERE, p_off = ok ? 2u * first : RS_NOWHERE;
    const unsigned int y_inc = ok ? 4u * RS_GROUP * static_cast<unsigned int>(up) : 0u, p_inc = ok ? 2u * RS_GROUP * static_cast<unsigned int>(up) : 0u;
    auto dropped_stores = [&]() {
        if constexpr (WANT_Y) __builtin_amdgcn_raw_buffer_store_b32(0u, y_dst, static_cast<int>(RS_NOWHERE), 0, 0);
        if constexpr (WANT_PCM) __builtin_amdgcn_raw_buffer_store_b16(static_cast<unsigned short>(0), p_dst, static_cast<int>(RS_NOWHERE), 0, 0);
    };

    // windows of the first RS_AHEAD steps, with (dropped) stores where two earlier groups would have ended
#pragma unroll
    for (int r = 0; r < RS_AHEAD; ++r) {
        issue(r);
        if ((r + 1) % RS_GROUP == 0) dropped_stores();
    }
    int rd = 0, fill = RS_AHEAD;
    for (int cur = 0; cur < steps; cur += RS_GROUP) {
        double part_sum[RS_GROUP];  // this lane's taps' share of the group's four sums
#pragma unroll
        for (int k = 0; k < RS_GROUP; ++k) {
            // refill the slot read TWO steps ago -- and only once that step's sum exists, i.e. its LDS reads have
            // returned (steps 0 and 1 of a group: the previous group's stores, in front of this point, took its sums)
            if (k >= 2) asm volatile("" : "+v"(lane16) : "v"(part_sum[k >= 2 ? k - 2 : 0]));
            issue(fill);
            wait_window(cur + k);
            const float *low = ring + rd * WIN + static_cast<int>(rd_pos & 3) + my_low;
            rd_pos += down;
            float xv[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) xv[i] = low[RS_LANES * (NI - 1 - i)];
            // four independent chains per step
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const double xd = static_cast<double>(xv[i]);
                if ((i & 3) == 0) a0 = fma(h[i], xd, a0);
                else if ((i & 3) == 1) a1 = fma(h[i], xd, a1);
                else if ((i & 3) == 2) a2 = fma(h[i], xd, a2);
                else a3 = fma(h[i], xd, a3);
            }
            part_sum[k] = (IQA_RS_ABLATE & 4) ? h[k] : (a0 + a1) + (a2 + a3);
            asm volatile("" ::: "memory");  // this window's reads stay in front of the later DMAs
            rd = rd + 1 == RS_RING ? 0 : rd + 1;
            fill = fill + 1 == RS_RING ? 0 : fill + 1;
        }
        // transposing reduction over the quad: lanes 0, 1 collect steps 0, 1 and lanes 2, 3 steps 2, 3 (one exchange
        // with lane ^ 2), then lane k keeps step k (one exchange with lane ^ 1)
        const bool upper = (sub & 2) != 0, odd = (sub & 1) != 0;
        const double r0 = (upper ? part_sum[2] : part_sum[0]) + quad_swap<0x4E>(upper ? part_sum[0] : part_sum[2]);
        const double r1 = (upper ? part_sum[3] : part_sum[1]) + quad_swap<0x4E>(upper ? part_sum[1] : part_sum[3]);
        const float v = static_cast<float>((odd ? r1 : r0) + quad_swap<0xB1>(odd ? r0 : r1));
        if constexpr (WANT_Y) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), y_dst,
                                                  static_cast<int>((IQA_RS_ABLATE & 1) && v != 1e30f ? RS_NOWHERE : y_off), 0, 0);
            y_off += y_inc;
        }
        if constexpr (WANT_PCM) {  // the writer's PCM16 leg in the same pass
            __builtin_amdgcn_raw_buffer_store_b16(static_cast<unsigned short>(pcm16_of_f32(v)), p_dst,
                                                  static_cast<int>((IQA_RS_ABLATE & 1) && v != 1e30f ? RS_NOWHERE : p_off), 0, 0);
            p_off += p_inc;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may land in LDS the workgroup has given back
}

template <int OP>
int run_scan(ScanArgs a, double *state_out, void *work, hipStream_t s)
{
    a.nblocks = static_cast<int>((a.n + SC_TILE - 1) / SC_TILE);
    char *w = static_cast<char *>(work);
    a.agg = reinterpret_cast<Aff *>(w);
    a.carry = reinterpret_cast<double *>(w + sizeof(Aff) * a.nblocks);
    a.fin = a.carry + a.nblocks;
    hipLaunchKernelGGL(k_scan_reduce<OP>, dim3(a.nblocks), dim3(SC_THREADS), 0, s, a);
    hipLaunchKernelGGL(k_scan_carry<OP>, dim3(1), dim3(kWave), 0, s, a);
    hipLaunchKernelGGL(k_scan_apply<OP>, dim3(a.nblocks), dim3(SC_THREADS), 0, s, a);
    if (OP != OP_AGC) hipLaunchKernelGGL(k_scan_finish<OP>, dim3(1), dim3(1), 0, s, a, state_out);
    return check_launch("affine scan");
}

}  // namespace iqa

using namespace iqa;

static inline dim3 grid1d(int64_t n, int block) { return dim3(static_cast<unsigned>((n + block - 1) / block)); }

extern "C" int64_t iqa_scan_workspace_bytes(int64_t n)
{
    const int64_t nb = (n + SC_TILE - 1) / SC_TILE;
    return nb * (sizeof(Aff) + sizeof(double)) + 64;
}

extern "C" int iqa_deemphasis(const void *x_dev, int64_t n, double alpha, void *state_dev, void *y_dev,
                              void *work_dev, void *stream)
{
    if (n < 0) return fail_inval("negative length");
    if (n == 0) return IQA_OK;
    if (!x_dev || !state_dev || !y_dev || !work_dev) return fail_inval("NULL device pointer");
    ScanArgs a{};
    a.x = static_cast<const float *>(x_dev);
    a.y = static_cast<float *>(y_dev);
    a.n = n;
    a.p0 = alpha;
    a.p1 = 1.0 - alpha;
    a.state = static_cast<const double *>(state_dev);
    return run_scan<OP_DEEMPH>(a, static_cast<double *>(state_dev), work_dev, as_stream(stream));
}

extern "C" int iqa_dc_block(const void *x_dev, int64_t n, double radius, void *state_dev, void *y_dev,
                            void *work_dev, void *stream)
{
    if (n < 0) return fail_inval("negative length");
    if (!(radius > 0.0 && radius < 1.0)) return fail_inval("radius must be between 0 and 1");
    if (n == 0) return IQA_OK;
    if (!x_dev || !state_dev || !y_dev || !work_dev) return fail_inval("NULL device pointer");
    ScanArgs a{};
    a.x = static_cast<const float *>(x_dev);
    a.y = static_cast<float *>(y_dev);
    a.n = n;
    a.p0 = static_cast<double>(static_cast<float>(radius));  // the reference's in-loop r is float32
    a.state = static_cast<const double *>(state_dev);
    return run_scan<OP_DC>(a, static_cast<double *>(state_dev), work_dev, as_stream(stream));
}

extern "C" int iqa_agc(const void *x_dev, int64_t n, double target, double decay, const void *reset_starts_dev,
                       int64_t n_resets, void *y_dev, void *work_dev, void *stream)
{
    if (n < 0 || n_resets < 0) return fail_inval("negative length");
    if (n == 0) return IQA_OK;
    if (!x_dev || !y_dev || !work_dev) return fail_inval("NULL device pointer");
    ScanArgs a{};
    a.x = static_cast<const float *>(x_dev);
    a.y = static_cast<float *>(y_dev);
    a.n = n;
    a.p0 = static_cast<double>(static_cast<float>(target));
    a.p1 = static_cast<double>(static_cast<float>(decay));
    a.resets = static_cast<const long long *>(reset_starts_dev);
    a.n_resets = n_resets;
    return run_scan<OP_AGC>(a, nullptr, work_dev, as_stream(stream));
}

extern "C" int iqa_quadrature(const void *z_dev, int64_t n, void *prev_dev, void *out_dev, void *stream)
{
    if (n < 0) return fail_inval("negative length");
    if (n == 0) return IQA_OK;
    if (!z_dev || !prev_dev || !out_dev) return fail_inval("NULL device pointer");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_quadrature, grid1d(n, 256), dim3(256), 0, s, static_cast<const float2 *>(z_dev), (long long)n,
                       static_cast<const float2 *>(prev_dev), static_cast<float *>(out_dev));
    hipLaunchKernelGGL(k_store_last, dim3(1), dim3(1), 0, s, static_cast<const float2 *>(z_dev), (long long)n,
                       static_cast<float2 *>(prev_dev));
    return check_launch("k_quadrature");
}

extern "C" int iqa_envelope(const void *z_dev, int64_t n, void *out_dev, void *stream)
{
    if (n < 0) return fail_inval("negative length");
    if (n == 0) return IQA_OK;
    if (!z_dev || !out_dev) return fail_inval("NULL device pointer");
    hipLaunchKernelGGL(k_envelope, grid1d(n, 256), dim3(256), 0, as_stream(stream), static_cast<const float2 *>(z_dev),
                       (long long)n, static_cast<float *>(out_dev));
    return check_launch("k_envelope");
}

extern "C" int iqa_real_part(const void *z_dev, int64_t n, void *out_dev, void *stream)
{
    if (n < 0) return fail_inval("negative length");
    if (n == 0) return IQA_OK;
    if (!z_dev || !out_dev) return fail_inval("NULL device pointer");
    hipLaunchKernelGGL(k_real, grid1d(n, 256), dim3(256), 0, as_stream(stream), static_cast<const float2 *>(z_dev),
                       (long long)n, static_cast<float *>(out_dev));
    return check_launch("k_real");
}

extern "C" int iqa_decimate(const void *in_dev, int64_t n, int64_t first, int32_t D, void *out_dev, int64_t n_out,
                            void *stream)
{
    if (n < 0 || n_out < 0 || first < 0 || D < 1) return fail_inval("bad decimate sizes");
    if (n_out == 0) return IQA_OK;
    if (first + (n_out - 1) * static_cast<int64_t>(D) >= n) return fail_inval("decimate reads past the input");
    if (!in_dev || !out_dev) return fail_inval("NULL device pointer");
    hipLaunchKernelGGL(k_decimate, grid1d(n_out, 256), dim3(256), 0, as_stream(stream),
                       static_cast<const float2 *>(in_dev), (long long)first, (int)D, static_cast<float2 *>(out_dev),
                       (long long)n_out);
    return check_launch("k_decimate");
}

extern "C" int iqa_mean_power(const void *z_dev, int64_t n, int64_t skip, void *power_dev, void *stream)
{
    if (n < 0 || skip < 0 || skip > n) return fail_inval("bad range");
    if (!power_dev) return fail_inval("NULL device pointer");
    hipStream_t s = as_stream(stream);
    const int64_t count = n - skip;
    if (count > 0 && count <= 65536) {
        if (!z_dev) return fail_inval("NULL device pointer");
        hipLaunchKernelGGL(k_mean_power_small, dim3(1), dim3(1024), 0, s, static_cast<const float2 *>(z_dev), (long long)n,
                           (long long)skip, 1.0 / static_cast<double>(count), static_cast<double *>(power_dev), 0LL);
        return check_launch("k_mean_power_small");
    }
    if (hipMemsetAsync(power_dev, 0, sizeof(double), s) != hipSuccess) {
        set_error("hipMemsetAsync failed");
        return IQA_EHIP;
    }
    if (count == 0) return IQA_OK;
    if (!z_dev) return fail_inval("NULL device pointer");
    const int64_t blocks = std::min<int64_t>((count + 255) / 256, 2048);
    hipLaunchKernelGGL(k_mean_power, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const float2 *>(z_dev),
                       (long long)n, (long long)skip, 1.0 / static_cast<double>(count), static_cast<double *>(power_dev));
    return check_launch("k_mean_power");
}

extern "C" int iqa_mean_power_batch(const void *z_dev, int64_t n_each, int32_t parts, int64_t skip, void *power_dev, void *stream)
{
    if (n_each < 0 || skip < 0 || skip > n_each || parts < 0) return fail_inval("bad range");
    if (parts == 0) return IQA_OK;
    if (!power_dev) return fail_inval("NULL device pointer");
    const int64_t count = n_each - skip;
    if (count > 0 && count <= 65536) {
        if (!z_dev) return fail_inval("NULL device pointer");
        hipLaunchKernelGGL(k_mean_power_small, dim3(static_cast<unsigned>(parts)), dim3(1024), 0, as_stream(stream),
                           static_cast<const float2 *>(z_dev), (long long)n_each, (long long)skip, 1.0 / static_cast<double>(count),
                           static_cast<double *>(power_dev), (long long)n_each);
        return check_launch("k_mean_power_small");
    }
    for (int32_t p = 0; p < parts; ++p) {  // long stretches: one reduction each
        const int rc = iqa_mean_power(z_dev ? static_cast<const float2 *>(z_dev) + static_cast<int64_t>(p) * n_each : nullptr, n_each, skip,
                                      static_cast<double *>(power_dev) + p, stream);
        if (rc != IQA_OK) return rc;
    }
    return IQA_OK;
}

extern "C" int iqa_raw_level(int32_t fmt, const void *raw_dev, int64_t n_values, void *mean_square_out, void *stream)
{
    if (fmt != IQA_FMT_S16 && fmt != IQA_FMT_U8 && fmt != IQA_FMT_F32) return fail_inval("bad sample format");
    if (n_values < 0) return fail_inval("negative length");
    if (!mean_square_out) return fail_inval("NULL output pointer");
    const int per_vec = fmt == IQA_FMT_S16 ? 8 : fmt == IQA_FMT_U8 ? 16 : 4;
    const long long n_vec = n_values / per_vec;
    if (n_vec > 0 && !raw_dev) return fail_inval("NULL device pointer");
    if (n_vec > 0 && (reinterpret_cast<uintptr_t>(raw_dev) & 15)) return fail_inval("raw frames must be 16-byte aligned");
    constexpr int ROUNDS = 8;  // up to 8192 vectors = 65536 int16 values: 0.4 % relative standard error on noise
    const long long step = std::max<long long>(1024, n_vec / ROUNDS);  // distance between the sampled stretches, in vectors
    hipStream_t s = as_stream(stream);
    double *out = static_cast<double *>(mean_square_out);
    const uint4 *raw = static_cast<const uint4 *>(raw_dev);
    if (fmt == IQA_FMT_S16) hipLaunchKernelGGL(k_raw_level<IQA_FMT_S16>, dim3(1), dim3(1024), 0, s, raw, n_vec, step, out);
    else if (fmt == IQA_FMT_U8) hipLaunchKernelGGL(k_raw_level<IQA_FMT_U8>, dim3(1), dim3(1024), 0, s, raw, n_vec, step, out);
    else hipLaunchKernelGGL(k_raw_level<IQA_FMT_F32>, dim3(1), dim3(1024), 0, s, raw, n_vec, step, out);
    return check_launch("k_raw_level");
}

extern "C" int iqa_writer_clip(const void *a_dev, int64_t n, void *peak_dev, const void *seg_starts_dev,
                               int64_t n_segs, void *sumsq_dev, void *out_dev, void *stream)
{
    if (n < 0 || n_segs < 0) return fail_inval("negative length");
    if (n == 0) return IQA_OK;
    if (!a_dev) return fail_inval("NULL device pointer");
    if (sumsq_dev && n_segs > 0 && !seg_starts_dev) return fail_inval("seg_starts is NULL");
    hipLaunchKernelGGL(k_writer_clip, grid1d(n, 256 * 4), dim3(256), 0, as_stream(stream),
                       static_cast<const float *>(a_dev), (long long)n, static_cast<unsigned int *>(peak_dev),
                       static_cast<const long long *>(seg_starts_dev), (long long)n_segs,
                       static_cast<double *>(sumsq_dev), static_cast<float *>(out_dev));
    return check_launch("k_writer_clip");
}

extern "C" int iqa_resample(const void *x_dev, int64_t n_in, const void *table_dev, int32_t up, int32_t down,
                            int32_t T, int64_t j0, int64_t n_out, void *y_dev, void *pcm16_dev, void *stream)
{
    if (n_in < 0 || n_out < 0 || j0 < 0 || up < 1 || down < 1 || T < 0) return fail_inval("bad resampler sizes");
    if (n_out == 0) return IQA_OK;
    if (!table_dev || (!y_dev && !pcm16_dev) || (n_in > 0 && !x_dev)) return fail_inval("NULL device pointer");
    if (2 * T + 1 > 192) return fail_inval("resampler rows longer than 192 taps are not supported");
    if (n_in > (1LL << 30) - 4096) return fail_inval("resampler input longer than 2^30 samples: process it in blocks");
    if (j0 > (1LL << 40) || static_cast<int64_t>(up) * down >= (1LL << 50)) return fail_inval("resampler position out of range");
    RsArgs a;
    a.x = static_cast<const float *>(x_dev);
    a.n_in = n_in;
    a.table = static_cast<const double *>(table_dev);
    a.up = up, a.down = down, a.T = T;
    a.n_out = n_out;
    a.y = static_cast<float *>(y_dev);
    a.pcm = static_cast<short *>(pcm16_dev);
    a.g_all = (n_out + up - 1) / up;  // outputs per polyphase row
    // enough waves to fill the chip (>= ~8 per SIMD) while a wave still amortises its 4*NI tap loads over several steps
    const int64_t groups = (static_cast<int64_t>(up) + 15) / 16;
    // ... and never fewer than RS_GROUP steps per wave (short streams: a wave's prologue -- its 17 tap loads, the first
    // windows -- costs as much as half a dozen steps)
    a.split = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(a.g_all / RS_GROUP, (IQA_RS_TARGET_WAVES + groups - 1) / groups)));
    a.g_per = (a.g_all + a.split - 1) / a.split;
    a.j0_mod_up = static_cast<int>(j0 % up);
    a.j0_q = j0 * down / up;  // (j0 * down < 2^40 * 2^31: the sizes above keep it inside int64)
    a.j0_r = j0 * down % up;
    // a wave addresses its stretch of the output (its steps x up samples) with 32-bit buffer offsets
    if ((a.g_per + RS_GROUP + 1) * static_cast<int64_t>(up) >= (1LL << 29))
        return fail_inval("resampler output too long for one launch at this ratio: process it in blocks");
    const dim3 grid = grid1d(groups * a.split, RS_WAVES), block(RS_WAVES * kWave);
    const int ni = (2 * T + 1 + 3) / 4;
#define IQA_RS_LAUNCH_AS(NI, WY, WP) hipLaunchKernelGGL((k_resample<NI, WY, WP>), grid, block, 0, as_stream(stream), a)
#define IQA_RS_LAUNCH(NI)                                         \
    do {                                                          \
        if (y_dev && pcm16_dev) IQA_RS_LAUNCH_AS(NI, true, true); \
        else if (y_dev) IQA_RS_LAUNCH_AS(NI, true, false);        \
        else IQA_RS_LAUNCH_AS(NI, false, true);                   \
    } while (0)
    if (ni <= 17) IQA_RS_LAUNCH(17);
    else if (ni <= 24) IQA_RS_LAUNCH(24);
    else if (ni <= 32) IQA_RS_LAUNCH(32);
    else IQA_RS_LAUNCH(48);
#undef IQA_RS_LAUNCH
#undef IQA_RS_LAUNCH_AS
    return check_launch("k_resample");
}

extern "C" int iqa_float_to_pcm16(const void *y_dev, int64_t n, void *pcm_dev, void *stream)
{
    if (n < 0) return fail_inval("negative length");
    if (n == 0) return IQA_OK;
    if (!y_dev || !pcm_dev) return fail_inval("NULL device pointer");
    hipLaunchKernelGGL(k_float_to_pcm16, grid1d(n, 256), dim3(256), 0, as_stream(stream),
                       static_cast<const float *>(y_dev), (long long)n, static_cast<short *>(pcm_dev));
    return check_launch("k_float_to_pcm16");
}

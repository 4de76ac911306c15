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
__global__ __launch_bounds__(1024) void k_mean_power_small(const float2 *z, long long n, long long skip, double inv_count,
                                                            double *out)
{
    __shared__ double s_w[16];
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
// touch ~100 inputs), so the wave stages that stretch of the input ONCE in its own 1 KB of LDS (two coalesced loads,
// prefetched a step ahead, zeros outside [0, n_in)) and every lane picks its taps' samples from there -- per-lane
// global loads (17 per lane and step) moved 268 B per output through the L1 and were what the kernel waited for.
// The cross-lane reduction is two quad steps per output.
// Sum over the four lanes of a quad, in every lane: two DPP quad permutes (register-to-register; __shfl_xor goes
// through ds_bpermute and its LDS latency, twice in a row, once per output).
__device__ __forceinline__ double quad_sum(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    v += __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true));  // lanes ^ 2
    lo = __double2loint(v);
    hi = __double2hiint(v);
    v += __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true));  // lanes ^ 1
    return v;
}

constexpr int RS_WAVES = 4;
constexpr int RS_LANES = 4;     // lanes per output
constexpr int RS_WINDOW = 256;  // floats of LDS per wave: 4*NI taps + the spread of the 16 outputs' positions
constexpr int RS_SPREAD = 60;   // largest spread of input positions inside a wave that the staged window covers

template <int NI>  // taps per lane = ceil(row_len / 4)
__global__ __launch_bounds__(RS_WAVES *kWave) void k_resample(const float *x, long long n_in, const double *table, int up,
                                                               int down, int T, long long j0, long long n_out, float *y,
                                                               short *pcm, int split)
{
    static_assert(4 * NI + RS_SPREAD + 4 <= RS_WINDOW, "window too small for this row length");
    constexpr int CHUNKS = (4 * NI + RS_SPREAD + 63) / 64;  // staging loads per lane and step
    __shared__ float s_x[RS_WAVES][RS_WINDOW];
    const int lane = threadIdx.x & 63, sub = lane & (RS_LANES - 1), slot = lane >> 2, wv = threadIdx.x >> 6;
    const long long wid = static_cast<long long>(blockIdx.x) * RS_WAVES + wv;
    const long long w = wid / split;  // group of 16 residues of (j0 + jj) mod up handled by this wave
    const int part = static_cast<int>(wid - w * split);
    if (w * 16 >= up) return;
    const int row_len = 2 * T + 1;
    const long long g_all = (n_out + up - 1) / up;  // steps any residue can need
    const long long g_per = (g_all + split - 1) / split;
    const long long g_lo = part * g_per, g_hi = min(g_all, g_lo + g_per);
    if (g_lo >= g_hi) return;
    // this quad's residue: first output jj0, its input position q0 and its row
    const long long res = w * 16 + slot;
    const bool ok = res < up;
    const long long jj0 = ok ? ((res - (j0 % up) + up) % up) : n_out;  // >= n_out: never live
    const long long c0 = (j0 + (ok ? jj0 : 0)) * down;
    const long long q0 = c0 / up;
    const int p = static_cast<int>(c0 - q0 * up);
    const double *row = table + static_cast<long long>(p) * row_len;
    // all loads unconditional (clamped index, masked afterwards): a branch around a load makes hipcc wait for
    // every load separately
    double h[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) h[i] = row[min(sub + RS_LANES * i, row_len - 1)];
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
    const bool staged = (qmax - qmin) <= RS_SPREAD;  // wave-uniform; false only where the residues wrap around `up`
    float *win = s_x[wv];
    const int my = ok ? static_cast<int>(q0 - qmin) + 4 * NI - 1 - sub : 4 * NI - 1 - sub;
    const long long w_first = qmin + T - (4 * NI - 1);  // + g*down
    // staging loads go through a buffer descriptor over x[0, n_in): positions outside it (the stream's edges) come
    // back as zeros from the range check -- no clamps, no branches around the loads (hipcc waits for a branched load
    // on the spot, which would undo the prefetch below)
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, static_cast<int>(n_in * 4), 0x00020000);
    float nxt[CHUNKS];
    if (staged) {
        const int k0 = static_cast<int>(w_first + g_lo * down) + lane;
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c)
            nxt[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, (k0 + 64 * c) * 4, 0, 0));
    }
    for (long long g = g_lo; g < g_hi; ++g) {
        const long long jj = jj0 + g * up;
        float xv[NI];
        if (staged) {
#pragma unroll
            for (int c = 0; c < CHUNKS; ++c)
                if (lane + 64 * c < RS_WINDOW) win[lane + 64 * c] = nxt[c];
            {  // next step's stretch: in flight while this one is multiplied (past the last step: harmless, in range or zero)
                const int k0 = static_cast<int>(w_first + (g + 1) * down) + lane;
#pragma unroll
                for (int c = 0; c < CHUNKS; ++c)
                    nxt[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, (k0 + 64 * c) * 4, 0, 0));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // one wave: its LDS operations complete in order
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int i = 0; i < NI; ++i) xv[i] = win[my - RS_LANES * i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // the next step's stores stay behind these reads
        } else {
            const long long top = q0 + g * down + T - sub;  // input index of this lane's first tap; tap i reads top - 4 i
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const long long nidx = top - RS_LANES * i;
                const float v = x[min(max(nidx, 0LL), max(n_in - 1, 0LL))];
                xv[i] = (nidx >= 0 && nidx < n_in) ? v : 0.f;
            }
        }
        // four independent chains: a wave has few neighbours to hide a 17-deep float64 FMA dependency behind
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const double xd = static_cast<double>(xv[i]);
            if ((i & 3) == 0) a0 = fma(h[i], xd, a0);
            else if ((i & 3) == 1) a1 = fma(h[i], xd, a1);
            else if ((i & 3) == 2) a2 = fma(h[i], xd, a2);
            else a3 = fma(h[i], xd, a3);
        }
        double acc = (a0 + a1) + (a2 + a3);
        acc = quad_sum(acc);
        if (jj < n_out && sub == 0) {
            const float v = static_cast<float>(acc);
            if (y != nullptr) y[jj] = v;
            if (pcm != nullptr) pcm[jj] = pcm16_of(v);  // the writer's PCM16 leg in the same pass
        }
    }
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
                           (long long)skip, 1.0 / static_cast<double>(count), static_cast<double *>(power_dev));
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
    const int64_t g_total = (n_out + up - 1) / up;  // outputs per polyphase row
    // enough waves to fill the chip (>= ~8 per SIMD) while a wave still amortises its 4*NI tap loads over several steps
    const int64_t groups = (static_cast<int64_t>(up) + 15) / 16;
    const int split = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(g_total, (8192 + groups - 1) / groups)));
    const dim3 grid = grid1d(groups * split, RS_WAVES), block(RS_WAVES * kWave);
    const int ni = (2 * T + 1 + 3) / 4;
#define IQA_RS_LAUNCH(NI)                                                                                          \
    hipLaunchKernelGGL((k_resample<NI>), grid, block, 0, as_stream(stream), static_cast<const float *>(x_dev),     \
                       (long long)n_in, static_cast<const double *>(table_dev), (int)up, (int)down, (int)T,        \
                       (long long)j0, (long long)n_out, static_cast<float *>(y_dev), static_cast<short *>(pcm16_dev), split)
    if (ni <= 17) IQA_RS_LAUNCH(17);
    else if (ni <= 24) IQA_RS_LAUNCH(24);
    else if (ni <= 32) IQA_RS_LAUNCH(32);
    else IQA_RS_LAUNCH(48);
#undef IQA_RS_LAUNCH
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

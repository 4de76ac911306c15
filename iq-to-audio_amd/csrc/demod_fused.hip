// demod_fused.hip -- whole demodulator + AudioWriter.write for a block of channel samples in three
// launches (reduce -> carry -> apply), instead of one launch per reference stage.
//
// Replaces, for one block of decimated samples z (reference src/iq_to_audio/):
//   decoder.process(z)            processing.py:1128
//     nfm: QuadratureDemod + DeemphasisFilter      decoders/nfm.py:17-24, 48-62
//     am : abs + DCBlocker                         decoders/am.py:28, decoders/common.py:16-30
//     ssb: real + DCBlocker [+ _apply_agc]         decoders/ssb.py:42-45, 65-80
//   audio_writer.write(audio)     processing.py:1147 -> :440-456 (pre-clip peak, clip +-0.99)
//   stats rms_dbfs per chunk      decoders/nfm.py:88-89 (sum of squares per reference chunk)
//
// The source stage (discriminator / envelope / real part) is evaluated on the fly inside the
// scan passes, and the sink (peak, clip, per-chunk sum of squares) inside the apply pass, so z is
// read twice and the audio written once: 20 B per channel sample instead of ~52.
// SSB with AGC needs two dependent recurrences and therefore two scans (DC blocker to a float
// scratch, then the segmented AGC scan with the sink).
#include "scan_common.h"

namespace iqa {

enum FOp { F_DEEMPH = 0, F_DC = 1, F_AGC = 2 };
enum FSrc { S_F32 = 0, S_QUAD = 1, S_ENV = 2, S_REAL = 3 };
enum FSink { K_PLAIN = 0, K_CLIP = 1 };

struct FusedArgs {
    const float2 *z;     // S_QUAD / S_ENV / S_REAL
    const float *x;      // S_F32
    float *y;
    long long n;
    double p0, p1;       // DEEMPH: alpha, 1-alpha | DC: radius (f32-rounded) | AGC: target (f32), decay (f32)
    const float2 *prev;  // S_QUAD: previous complex sample
    const double *st;    // DEEMPH: {y_last} | DC: {x_last, y_last}
    const long long *segs;  // chunk starts (AGC restarts and statistics segments), segs[0] == 0
    long long n_segs;
    unsigned int *peak_bits;
    double *sumsq;
    Aff *agg;
    double *carry;
    double *fin;
    int nblocks;
    int z_aligned, y_aligned;  // z / y are 16-byte aligned: the vector load / store paths may be used
    int fresh;  // the decoder has seen nothing: prev = 1 + 0j, filter states 0 (the caller's state block is not read), and the
                // carry pass clears the peak and the per-chunk sums -- iqa_demodulate_from_reset, no reset copy in front
};

template <int SRC>
__device__ __forceinline__ float src_value(const FusedArgs &a, long long i, float2 zc, float2 zp)
{
    if constexpr (SRC == S_QUAD) {
        const float re = zc.x * zp.x + zc.y * zp.y;  // z * conj(z_prev), float32 as numpy forms it
        const float im = zc.y * zp.x - zc.x * zp.y;
        return atan2f(im, re);
    } else if constexpr (SRC == S_ENV) {
        return hypotf(zc.x, zc.y);
    } else {
        return zc.x;
    }
}

// u[base-1 .. base+7] -> x[0..7] plus x_before (only DC needs u[base-1])
template <int OP, int SRC>
__device__ __forceinline__ void load_u(const FusedArgs &a, long long base, float (&u)[SC_ITEMS], float &u_before)
{
    u_before = 0.f;
    if constexpr (SRC == S_F32) {
#pragma unroll
        for (int i = 0; i < SC_ITEMS; ++i) u[i] = (base + i < a.n) ? a.x[base + i] : 0.f;
        if constexpr (OP == F_DC) {
            if (base == 0) u_before = a.fresh ? 0.f : static_cast<float>(a.st[0]);
            else if (base - 1 < a.n) u_before = a.x[base - 1];
        }
    } else {
        float2 zz[SC_ITEMS + 2];  // z[base-2 .. base+7]
        static_assert(SC_ITEMS == 8, "the vector path below moves 8 float2 = 4 x 16 bytes per thread");
        // Whole-wave fast path (every tile but the last): a thread's eight samples are 64 contiguous, 16-byte
        // aligned bytes -> four 16-byte loads; the two samples in front of them are the previous lane's last two
        // (one shuffle each) except for lane 0.  Scalar float2 loads at a 64-byte lane stride cost 2.5x the
        // load instructions and touch every line five times.
        const int lane = threadIdx.x & 63;
        const long long wave_base = base - static_cast<long long>(lane) * SC_ITEMS;
        if (a.z_aligned && wave_base + 64 * SC_ITEMS <= a.n) {
            const float4 *p = reinterpret_cast<const float4 *>(a.z + base);
            const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
            zz[2] = make_float2(q0.x, q0.y);
            zz[3] = make_float2(q0.z, q0.w);
            zz[4] = make_float2(q1.x, q1.y);
            zz[5] = make_float2(q1.z, q1.w);
            zz[6] = make_float2(q2.x, q2.y);
            zz[7] = make_float2(q2.z, q2.w);
            zz[8] = make_float2(q3.x, q3.y);
            zz[9] = make_float2(q3.z, q3.w);
            float2 e0 = make_float2(0.f, 0.f), e1 = make_float2(0.f, 0.f);
            if (lane == 0 && base >= 2) {
                e0 = a.z[base - 2];
                e1 = a.z[base - 1];
            }
            zz[0] = make_float2(__shfl_up(q3.x, 1, kWave), __shfl_up(q3.y, 1, kWave));
            zz[1] = make_float2(__shfl_up(q3.z, 1, kWave), __shfl_up(q3.w, 1, kWave));
            if (lane == 0) {
                zz[0] = e0;
                zz[1] = e1;
            }
        } else {
#pragma unroll
            for (int i = 0; i < SC_ITEMS + 2; ++i) {
                const long long k = base - 2 + i;
                zz[i] = (k >= 0 && k < a.n) ? a.z[k] : make_float2(0.f, 0.f);
            }
        }
        if constexpr (SRC == S_QUAD) {
            if (base == 0) zz[1] = a.fresh ? make_float2(1.f, 0.f) : a.prev[0];  // z[-1]
        }
#pragma unroll
        for (int i = 0; i < SC_ITEMS; ++i) u[i] = (base + i < a.n) ? src_value<SRC>(a, base + i, zz[i + 2], zz[i + 1]) : 0.f;
        if constexpr (OP == F_DC) {
            // DC blocker's x[n-1]: u[base-1] from z (ENV/REAL never need z[base-2]); carried state at 0
            if (base == 0) u_before = a.fresh ? 0.f : static_cast<float>(a.st[0]);
            else u_before = src_value<SRC>(a, base - 1, zz[1], zz[0]);
        }
    }
}

template <int OP>
__device__ __forceinline__ Aff fmap(const FusedArgs &a, long long idx, float x, float x_prev, bool maybe_reset)
{
    if constexpr (OP == F_DEEMPH) {
        return Aff{a.p0, a.p1 * static_cast<double>(x)};
    } else if constexpr (OP == F_DC) {
        return Aff{a.p0, static_cast<double>(x - x_prev)};
    } else {
        const float mag = fabsf(x);
        Aff m{1.0, 0.0};
        if (mag > 1e-6f) {
            const float desired = static_cast<float>(a.p0) / mag;
            m = Aff{1.0 - a.p1, a.p1 * static_cast<double>(desired)};
        }
        if (maybe_reset) {
            bool rst = (idx == 0);
            if (!rst && a.segs != nullptr && a.n_segs > 0) {
                const long long lo = lower_bound_ll(a.segs, a.n_segs, idx);
                rst = lo < a.n_segs && a.segs[lo] == idx;
            }
            if (rst) m = Aff{0.0, m.A + m.B};
        }
        return m;
    }
}

__device__ __forceinline__ bool block_has_restart(const FusedArgs &a, long long lo_i, long long hi_i)
{
    if (lo_i == 0) return true;
    if (a.segs == nullptr || a.n_segs <= 0) return false;
    const long long lo = lower_bound_ll(a.segs, a.n_segs, lo_i);
    return lo < a.n_segs && a.segs[lo] < hi_i;
}

template <int OP, int SRC>
__global__ __launch_bounds__(SC_THREADS) void k_fused_reduce(FusedArgs a)
{
    __shared__ Aff s_w[SC_THREADS / kWave];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long blk0 = static_cast<long long>(blockIdx.x) * SC_TILE;
    const long long base = blk0 + static_cast<long long>(tid) * SC_ITEMS;
    const bool mr = (OP == F_AGC) && block_has_restart(a, blk0, blk0 + SC_TILE);
    float u[SC_ITEMS], ub;
    load_u<OP, SRC>(a, base, u, ub);
    Aff t{1.0, 0.0};
#pragma unroll
    for (int i = 0; i < SC_ITEMS; ++i)
        if (base + i < a.n) t = then(t, fmap<OP>(a, base + i, u[i], i ? u[i - 1] : ub, mr));
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

// Carry pass: exclusive scan of the per-tile maps.  One block of 1024 threads: a thread composes its run of
// consecutive tiles, a wave scan and a 16-entry table in LDS give it the state in front of its run, and it walks
// the run once more to store the carries (a single wave walking thousands of tiles serially cost 30 us).
constexpr int FC_THREADS = 1024;
template <int OP>
__global__ __launch_bounds__(FC_THREADS) void k_fused_carry(FusedArgs a)
{
    __shared__ Aff s_w[FC_THREADS / kWave];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double s0;
    if constexpr (OP == F_DEEMPH) s0 = a.fresh ? 0.0 : a.st[0];
    else if constexpr (OP == F_DC) s0 = a.fresh ? 0.0 : a.st[1];
    else s0 = 1.0;
    const int per = (a.nblocks + FC_THREADS - 1) / FC_THREADS;
    const int b0 = tid * per;
    Aff t{1.0, 0.0};
    for (int i = 0; i < per; ++i)
        if (b0 + i < a.nblocks) t = then(t, a.agg[b0 + i]);
    const Aff inc = wave_inclusive(t, lane);
    if (lane == kWave - 1) s_w[wave] = inc;
    __syncthreads();
    Aff pre{1.0, 0.0};
    for (int w = 0; w < wave; ++w) pre = then(pre, s_w[w]);
    Aff exl{__shfl_up(inc.A, 1, kWave), __shfl_up(inc.B, 1, kWave)};
    if (lane == 0) exl = Aff{1.0, 0.0};
    const Aff ex = then(pre, exl);
    double s = fma(ex.A, s0, ex.B);  // state in front of this thread's run
    for (int i = 0; i < per; ++i) {
        const int b = b0 + i;
        if (b < a.nblocks) {
            a.carry[b] = s;
            const Aff v = a.agg[b];
            s = fma(v.A, s, v.B);
        }
    }
    if (tid == FC_THREADS - 1) a.fin[0] = s;  // runs past the end are identity maps: the last thread holds the total
    // Snapshot of the incoming state for the apply pass (fin[1] = prev, fin[2] = st[0]): its last block hands the
    // outgoing state to the next call while its first block may not have read the incoming one yet.
    if (a.fresh) {  // (the apply pass -- the only writer of these -- runs behind this kernel)
        if (tid == 0 && a.peak_bits != nullptr) a.peak_bits[0] = 0u;
        if (a.sumsq != nullptr)
            for (long long i = tid; i < a.n_segs * IQA_SUMSQ_SLOTS; i += FC_THREADS) a.sumsq[i] = 0.0;
    }
    if (tid == 0) {
        if (a.prev != nullptr) reinterpret_cast<float2 *>(a.fin + 1)[0] = a.fresh ? make_float2(1.f, 0.f) : a.prev[0];
        if (a.st != nullptr) a.fin[2] = a.fresh ? 0.0 : a.st[0];
    }
}

// `a.prev` / `a.st` point at the carry pass's snapshot; prev_out / st_out at the caller's state block (written by
// the last block: the streaming state for the next call, OP != F_AGC)
template <int OP, int SRC, int SINK>
__global__ __launch_bounds__(SC_THREADS) void k_fused_apply(FusedArgs a, float2 *prev_out, double *st_out)
{
    __shared__ Aff s_w[SC_THREADS / kWave];
    __shared__ float s_pk[SC_THREADS / kWave];
    __shared__ double s_sq[2 * (SC_THREADS / kWave)];  // low | high part of a tile that straddles a chunk boundary
    __shared__ long long s_seg[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long blk0 = static_cast<long long>(blockIdx.x) * SC_TILE;
    const long long base = blk0 + static_cast<long long>(tid) * SC_ITEMS;
    const bool mr = (OP == F_AGC) && block_has_restart(a, blk0, blk0 + SC_TILE);
    const bool stats = (SINK == K_CLIP) && a.sumsq != nullptr && a.n_segs > 0;
    // Issued first, consumed late: the state in front of this tile, and the segment (reference chunk) of the tile's
    // first and last element, counted by the whole block in ONE round of loads (a two-thread binary search in front
    // of the first barrier cost eight dependent global loads per block: two thirds of this kernel's time was waiting).
    const double carry_in = a.carry[blockIdx.x];
    // this thread's share of the chunk starts (one round of loads, issued in front of the tile's own loads)
    long long seg_first = 0;
    if (stats && tid < a.n_segs) seg_first = a.segs[tid];
    if (stats && tid < 2) s_seg[tid] = -1;
    float u[SC_ITEMS], ub;
    load_u<OP, SRC>(a, base, u, ub);
    // The maps of the AGC (a division and a restart look-up each) are kept for the second walk; those of the two
    // linear filters are one multiply / one subtract and are formed again there: 16 doubles fewer to hold, which is what
    // decides between five and eight resident waves per SIMD for a kernel that lives two or three rounds of blocks.
    constexpr bool KEEP = (OP == F_AGC);
    Aff m[KEEP ? SC_ITEMS : 1];
    Aff t{1.0, 0.0};
#pragma unroll
    for (int i = 0; i < SC_ITEMS; ++i) {
        const Aff mi = (base + i < a.n) ? fmap<OP>(a, base + i, u[i], i ? u[i - 1] : ub, mr) : Aff{1.0, 0.0};
        if constexpr (KEEP) m[i] = mi;
        t = then(t, mi);
    }
    if (stats) {
        __syncthreads();  // s_seg initialised
        const long long first = blk0, last = min(blk0 + SC_TILE, a.n) - 1;
        int c0 = (tid < a.n_segs) && seg_first <= first, c1 = (tid < a.n_segs) && seg_first <= last;
        for (long long kk = tid + SC_THREADS; kk < a.n_segs; kk += SC_THREADS) {
            const long long st = a.segs[kk];
            c0 += st <= first;
            c1 += st <= last;
        }
        c0 = static_cast<int>(wave_sum(static_cast<float>(c0)));
        c1 = static_cast<int>(wave_sum(static_cast<float>(c1)));
        if (lane == 0 && (c0 | c1)) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&s_seg[0]), static_cast<unsigned long long>(c0));
            atomicAdd(reinterpret_cast<unsigned long long *>(&s_seg[1]), static_cast<unsigned long long>(c1));
        }
    }
    const Aff inc = wave_inclusive(t, lane);
    if (lane == kWave - 1) s_w[wave] = inc;
    __syncthreads();
    const double ea = __shfl_up(inc.A, 1, kWave), eb = __shfl_up(inc.B, 1, kWave);
    Aff ex = (lane == 0) ? Aff{1.0, 0.0} : Aff{ea, eb};
    Aff wpre{1.0, 0.0};
    for (int w = 0; w < wave; ++w) wpre = then(wpre, s_w[w]);
    ex = then(wpre, ex);
    double s = fma(ex.A, carry_in, ex.B);

    // A tile (2048 samples) lies inside one reference chunk (`uniform`) or, a chunk being >= 40 k samples, straddles
    // exactly one boundary (`simple`): the squares go to a low and a high running sum split at that boundary, reduced
    // over the block, two atomics per block at most.  Only tiles with several boundaries (chunks shorter than a tile)
    // take the general per-thread path with its own look-ups.
    const long long seg0 = stats ? s_seg[0] : -1, seg1 = stats ? s_seg[1] : -1;
    const bool uniform = stats && (seg0 == seg1);
    const bool simple = stats && (seg1 == seg0 + 1);
    const bool general = stats && !uniform && !simple;
    long long bnd = a.n;  // first index of the high part
    if (simple) bnd = a.segs[seg1];
    long long seg = seg0;
    if (general && base < a.n) seg = lower_bound_ll(a.segs, a.n_segs, base + 1) - 1;
    float pk = 0.f;
    double run = 0.0, run_hi = 0.0;
    float vout[SC_ITEMS];
#pragma unroll
    for (int i = 0; i < SC_ITEMS; ++i) {
        const long long idx = base + i;
        Aff mi;
        if constexpr (KEEP) mi = m[i];
        else mi = (idx < a.n) ? fmap<OP>(a, idx, u[i], i ? u[i - 1] : ub, mr) : Aff{1.0, 0.0};
        s = fma(mi.A, s, mi.B);
        vout[i] = 0.f;
        if (idx < a.n) {
            const float v = (OP == F_AGC) ? u[i] * static_cast<float>(s) : static_cast<float>(s);
            if constexpr (SINK == K_CLIP) {
                pk = fmaxf(pk, fabsf(v));
                vout[i] = fminf(fmaxf(v, -0.99f), 0.99f);
                if (stats) {
                    const double vv = static_cast<double>(v) * static_cast<double>(v);
                    if (general) {
                        while (seg + 1 < a.n_segs && a.segs[seg + 1] <= idx) {
                            if (run != 0.0) atomicAdd(&a.sumsq[seg * IQA_SUMSQ_SLOTS + (blockIdx.x & (IQA_SUMSQ_SLOTS - 1))], run);
                            run = 0.0;
                            ++seg;
                        }
                        run += vv;
                    } else {
                        run += idx < bnd ? vv : 0.0;
                        run_hi += idx < bnd ? 0.0 : vv;
                    }
                }
            } else {
                vout[i] = v;
            }
        }
    }
    if constexpr (SINK == K_CLIP) {
        if (general && run != 0.0) atomicAdd(&a.sumsq[seg * IQA_SUMSQ_SLOTS + (blockIdx.x & (IQA_SUMSQ_SLOTS - 1))], run);
        pk = wave_max(pk);
        const double wlo = (stats && !general) ? wave_sum(run) : 0.0;
        const double whi = simple ? wave_sum(run_hi) : 0.0;
        if (lane == 0) {
            s_pk[wave] = pk;
            s_sq[wave] = wlo;
            s_sq[4 + wave] = whi;
        }
        __syncthreads();
        if (tid == 0) {
            if (a.peak_bits != nullptr) {
                // thousands of atomics on one word serialise in L2 (~11 ns each): skip the ones that cannot
                // raise the running maximum (a stale read only costs a redundant atomic, never a wrong result)
                const unsigned int m = __float_as_uint(fmaxf(fmaxf(s_pk[0], s_pk[1]), fmaxf(s_pk[2], s_pk[3])));
                if (m > __hip_atomic_load(a.peak_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.peak_bits, m);
            }
            const int sub = blockIdx.x & (IQA_SUMSQ_SLOTS - 1);
            if (stats && !general) atomicAdd(&a.sumsq[seg0 * IQA_SUMSQ_SLOTS + sub], s_sq[0] + s_sq[1] + s_sq[2] + s_sq[3]);
            if (simple) atomicAdd(&a.sumsq[seg1 * IQA_SUMSQ_SLOTS + sub], s_sq[4] + s_sq[5] + s_sq[6] + s_sq[7]);
        }
    }
    if (OP != F_AGC && blockIdx.x == a.nblocks - 1 && tid == 0) {  // hand the streaming state to the next call
        if constexpr (SRC == S_QUAD) prev_out[0] = a.z[a.n - 1];
        if constexpr (OP == F_DEEMPH) {
            st_out[0] = a.fin[0];
        } else if constexpr (OP == F_DC) {
            float last;
            if constexpr (SRC == S_F32) last = a.x[a.n - 1];
            else last = src_value<SRC>(a, a.n - 1, a.z[a.n - 1], make_float2(0.f, 0.f));
            st_out[0] = static_cast<double>(last);
            st_out[1] = a.fin[0];
        }
    }
    // the audio goes out last: nothing waits for these stores (a barrier behind them would)
    if (a.y_aligned && base + SC_ITEMS <= a.n) {  // 32 contiguous, aligned bytes per thread: two 16-byte stores
        float4 *yp = reinterpret_cast<float4 *>(a.y + base);
        yp[0] = make_float4(vout[0], vout[1], vout[2], vout[3]);
        yp[1] = make_float4(vout[4], vout[5], vout[6], vout[7]);
    } else {
#pragma unroll
        for (int i = 0; i < SC_ITEMS; ++i)
            if (base + i < a.n) a.y[base + i] = vout[i];
    }
}

template <int OP, int SRC, int SINK>
static int launch_fused(FusedArgs a, float2 *prev_out, double *st_out, void *work, hipStream_t s)
{
    a.nblocks = static_cast<int>((a.n + SC_TILE - 1) / SC_TILE);
    char *w = static_cast<char *>(work);
    a.agg = reinterpret_cast<Aff *>(w);
    a.carry = reinterpret_cast<double *>(w + sizeof(Aff) * a.nblocks);
    a.fin = a.carry + a.nblocks;
    hipLaunchKernelGGL((k_fused_reduce<OP, SRC>), dim3(a.nblocks), dim3(SC_THREADS), 0, s, a);
    hipLaunchKernelGGL((k_fused_carry<OP>), dim3(1), dim3(FC_THREADS), 0, s, a);
    FusedArgs b = a;  // the apply pass reads the incoming state from the carry pass's snapshot
    b.prev = reinterpret_cast<const float2 *>(a.fin + 1);
    b.st = a.fin + 2;
    b.fresh = 0;
    hipLaunchKernelGGL((k_fused_apply<OP, SRC, SINK>), dim3(a.nblocks), dim3(SC_THREADS), 0, s, b, prev_out, st_out);
    return check_launch("fused demodulator");
}

}  // namespace iqa

using namespace iqa;

static int demodulate(const iqa_demod_params *p, const void *z_dev, int64_t n, void *state_dev, const void *seg_starts_dev,
                      int64_t n_segs, void *peak_dev, void *sumsq_dev, void *audio_out_dev, void *scratch_dev, void *work_dev,
                      void *stream, int fresh)
{
    if (p == nullptr) return fail_inval("params is NULL");
    if (n < 0 || n_segs < 0) return fail_inval("negative length");
    if (p->mode < IQA_DEMOD_NFM || p->mode > IQA_DEMOD_LSB) return fail_inval("Unsupported demod mode");
    if (n == 0) return IQA_OK;
    if (!z_dev || !state_dev || !audio_out_dev || !work_dev) return fail_inval("NULL device pointer");
    if (n_segs > 0 && !seg_starts_dev) return fail_inval("seg_starts is NULL");
    hipStream_t s = as_stream(stream);
    // state block: [0] float2 prev (8 B) | [8] double y_last | [16] double x_last, y_last
    char *st = static_cast<char *>(state_dev);
    FusedArgs a{};
    a.fresh = fresh;
    a.z = static_cast<const float2 *>(z_dev);
    a.n = n;
    a.segs = static_cast<const long long *>(seg_starts_dev);
    a.n_segs = n_segs;
    a.peak_bits = static_cast<unsigned int *>(peak_dev);
    a.sumsq = static_cast<double *>(sumsq_dev);
    a.y = static_cast<float *>(audio_out_dev);
    a.z_aligned = (reinterpret_cast<uintptr_t>(z_dev) & 15) == 0;
    a.y_aligned = (reinterpret_cast<uintptr_t>(audio_out_dev) & 15) == 0;
    a.prev = reinterpret_cast<const float2 *>(st);
    if (p->mode == IQA_DEMOD_NFM) {
        a.p0 = p->deemph_alpha;
        a.p1 = 1.0 - p->deemph_alpha;
        a.st = reinterpret_cast<const double *>(st + 8);
        return launch_fused<F_DEEMPH, S_QUAD, K_CLIP>(a, reinterpret_cast<float2 *>(st), reinterpret_cast<double *>(st + 8),
                                                      work_dev, s);
    }
    if (!(p->dc_radius > 0.0 && p->dc_radius < 1.0)) return fail_inval("radius must be between 0 and 1");
    a.p0 = static_cast<double>(static_cast<float>(p->dc_radius));
    a.st = reinterpret_cast<const double *>(st + 16);
    double *dc_out = reinterpret_cast<double *>(st + 16);
    if (p->mode == IQA_DEMOD_AM) return launch_fused<F_DC, S_ENV, K_CLIP>(a, nullptr, dc_out, work_dev, s);
    if (!p->agc_enabled) return launch_fused<F_DC, S_REAL, K_CLIP>(a, nullptr, dc_out, work_dev, s);
    // SSB with AGC: DC blocker into scratch, then the segmented AGC scan with the writer sink
    if (!scratch_dev) return fail_inval("SSB with AGC needs a float scratch buffer of n elements");
    FusedArgs d = a;
    d.y = static_cast<float *>(scratch_dev);
    d.y_aligned = (reinterpret_cast<uintptr_t>(scratch_dev) & 15) == 0;
    d.peak_bits = nullptr;
    d.sumsq = nullptr;
    int rc = launch_fused<F_DC, S_REAL, K_PLAIN>(d, nullptr, dc_out, work_dev, s);
    if (rc != IQA_OK) return rc;
    FusedArgs g = a;
    g.z = nullptr;
    g.x = static_cast<const float *>(scratch_dev);
    g.p0 = static_cast<double>(static_cast<float>(p->agc_target));
    g.p1 = static_cast<double>(static_cast<float>(p->agc_decay));
    g.st = nullptr;
    return launch_fused<F_AGC, S_F32, K_CLIP>(g, nullptr, nullptr, work_dev, s);
}

extern "C" int iqa_demodulate(const iqa_demod_params *p, const void *z_dev, int64_t n, void *state_dev,
                              const void *seg_starts_dev, int64_t n_segs, void *peak_dev, void *sumsq_dev,
                              void *audio_out_dev, void *scratch_dev, void *work_dev, void *stream)
{
    return demodulate(p, z_dev, n, state_dev, seg_starts_dev, n_segs, peak_dev, sumsq_dev, audio_out_dev, scratch_dev, work_dev, stream, 0);
}

extern "C" int iqa_demodulate_from_reset(const iqa_demod_params *p, const void *z_dev, int64_t n, void *state_dev,
                                         const void *seg_starts_dev, int64_t n_segs, void *peak_dev, void *sumsq_dev,
                                         void *audio_out_dev, void *scratch_dev, void *work_dev, void *stream)
{
    if (n == 0) return fail_inval("iqa_demodulate_from_reset needs samples (an empty block resets nothing)");
    return demodulate(p, z_dev, n, state_dev, seg_starts_dev, n_segs, peak_dev, sumsq_dev, audio_out_dev, scratch_dev, work_dev, stream, 1);
}

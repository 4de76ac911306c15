// channelize.hip -- fused ingest + NCO mix + channel FIR + decimate for gfx950.
//
// Replaces (reference src/iq_to_audio/processing.py):
//   IQReader._extract_iq :268-279, ComplexOscillator.mix :289-297,
//   OverlapSaveFIR.process :325-346, Decimator.process :354-360
// as ONE pass over the raw capture: only the kept outputs (global index = 0 mod D)
// are computed, each as a direct dot product of the raw int16/u8/f32 frames with
// host-pre-rotated complex taps g[k] = h[k] e^{+j s w k}; the NCO then reduces to one
// rotation per OUTPUT sample (64-bit fixed-point phase, float64 sincospi).
//
// v1 mapping ("wave per output, lanes over taps, shuffle reduction"):
//   block = 8 waves, wave = 4 consecutive outputs, lane = 4 consecutive taps per step.
//   Taps are staged through LDS in 2048-tap slices (shared by the block's 32 outputs);
//   frames are read straight from global memory with one 16-byte load per 4 frames
//   (coalesced 1 KiB per wave-instruction; neighbouring outputs re-read the same
//   lines out of L1/L2).  Cross-lane reduction by DPP/shuffle butterfly at the end.
#include "common.h"

namespace iqa {

constexpr int CH_WAVES = 8;
constexpr int CH_R = 4;
constexpr int CH_TCH = 2048;  // taps per LDS slice (16 KiB of float2)
constexpr int CH_OUT_PER_BLOCK = CH_WAVES * CH_R;
constexpr int CH_THREADS = CH_WAVES * kWave;
constexpr int CH_TAP_ALIGN = 256;  // taps padded to a multiple of 64 lanes x 4

struct ChanArgs {
    const float2 *taps;
    const void *raw;
    const void *hist;
    float2 *out;
    long long n_frames, consumed, m_first, n_out;
    int L, Lpad, D;
    int conj_sum, rotate;
    unsigned long long rot_step, rot_base;
    float sc_re, sc_im;
};

// ---- frame loaders -------------------------------------------------------------------------
// Unaligned 16-byte global loads are legal on gfx950 under HSA (unaligned access mode);
// vector types declared with the element's own alignment make hipcc emit one
// global_load_dwordx4 / dwordx2 instead of splitting the access.
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef v4i_t v4i_a4 __attribute__((aligned(4)));
typedef unsigned short v4us_t __attribute__((ext_vector_type(4)));
typedef v4us_t v4us_a2 __attribute__((aligned(2)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef v4f_t v4f_a8 __attribute__((aligned(8)));

template <int FMT>
__device__ __forceinline__ void load4(const void *raw, long long f, float (&re)[4], float (&im)[4]);

template <>
__device__ __forceinline__ void load4<IQA_FMT_S16>(const void *raw, long long f, float (&re)[4], float (&im)[4])
{
    const v4i_t v = *reinterpret_cast<const v4i_a4 *>(reinterpret_cast<const int *>(raw) + f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        re[j] = static_cast<float>(static_cast<short>(v[j] & 0xffff));
        im[j] = static_cast<float>(v[j] >> 16);
    }
}

template <>
__device__ __forceinline__ void load4<IQA_FMT_U8>(const void *raw, long long f, float (&re)[4], float (&im)[4])
{
    const v4us_t v = *reinterpret_cast<const v4us_a2 *>(reinterpret_cast<const unsigned short *>(raw) + f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        re[j] = static_cast<float>(v[j] & 0xff) - 128.0f;
        im[j] = static_cast<float>(v[j] >> 8) - 128.0f;
    }
}

template <>
__device__ __forceinline__ void load4<IQA_FMT_F32>(const void *raw, long long f, float (&re)[4], float (&im)[4])
{
    const float *p = reinterpret_cast<const float *>(raw) + 2 * f;
    const v4f_t lo = *reinterpret_cast<const v4f_a8 *>(p);
    const v4f_t hi = *reinterpret_cast<const v4f_a8 *>(p + 4);
    re[0] = lo.x; im[0] = lo.y; re[1] = lo.z; im[1] = lo.w;
    re[2] = hi.x; im[2] = hi.y; re[3] = hi.z; im[3] = hi.w;
}

template <int FMT>
__device__ __forceinline__ float2 load1(const void *buf, long long f)
{
    if constexpr (FMT == IQA_FMT_S16) {
        int v = reinterpret_cast<const int *>(buf)[f];
        return make_float2(static_cast<float>(static_cast<short>(v & 0xffff)), static_cast<float>(v >> 16));
    } else if constexpr (FMT == IQA_FMT_U8) {
        unsigned short v = reinterpret_cast<const unsigned short *>(buf)[f];
        return make_float2(static_cast<float>(v & 0xff) - 128.0f, static_cast<float>(v >> 8) - 128.0f);
    } else {
        return reinterpret_cast<const float2 *>(buf)[f];
    }
}

// frame `f` of the virtual stream (hist | raw); zero outside.
template <int FMT>
__device__ __forceinline__ float2 load_guarded(const ChanArgs &a, long long f)
{
    if (f >= 0) {
        if (f < a.n_frames) return load1<FMT>(a.raw, f);
        return make_float2(0.f, 0.f);
    }
    long long h = f + (a.L - 1);
    if (a.hist != nullptr && h >= 0) return load1<FMT>(a.hist, h);
    return make_float2(0.f, 0.f);
}

// SPLITK = false: throughput form (wave = 4 outputs x all taps; 32 outputs per block).
// SPLITK = true : latency form for short launches (mixer-sign probe, head/tail of a block): the 8 waves
//                 of a block share 4 outputs and split the taps 8 ways, so a 4800-output probe becomes
//                 1200 blocks of 3-4 iterations instead of 151 blocks of 26.
template <int FMT, bool SPLITK>
__global__ __launch_bounds__(CH_THREADS) void k_channelize_v1(ChanArgs a)
{
    __shared__ __attribute__((aligned(16))) float2 s_taps[CH_TCH];
    __shared__ float s_part[CH_WAVES][CH_R][2];
    constexpr int OUT_PER_BLOCK = SPLITK ? CH_R : CH_OUT_PER_BLOCK;
    constexpr int LANE_STRIDE = SPLITK ? CH_WAVES * kWave * 4 : kWave * 4;

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    // XCD-aware tile order: blocks are dealt round-robin to the 8 XCDs (b and b+8 share an L2), and
    // neighbouring output tiles re-read ~L/(32 D) of each other's window.  Give every XCD one
    // contiguous eighth of the tiles so those re-reads hit its own L2 instead of HBM.
    // (speed only: any placement computes the same outputs)
    const unsigned nblk = gridDim.x, per = nblk >> 3;
    unsigned tile = blockIdx.x;
    if (tile < per * 8u) tile = (tile & 7u) * per + (tile >> 3);
    const long long o_blk = static_cast<long long>(tile) * OUT_PER_BLOCK;
    const long long o0 = o_blk + (SPLITK ? 0 : wave * CH_R);
    const int lane0 = (SPLITK ? (wave * kWave + lane) : lane) * 4;

    // window start (local frame index) of each of this wave's outputs
    long long start[CH_R];
#pragma unroll
    for (int r = 0; r < CH_R; ++r) start[r] = (a.m_first + o0 + r) * a.D - a.consumed - (a.L - 1);

    // block-uniform: can every lane of every wave read 4 frames unguarded?
    const long long blk_first = (a.m_first + o_blk) * a.D - a.consumed - (a.L - 1);
    const long long blk_last = blk_first + static_cast<long long>(OUT_PER_BLOCK - 1) * a.D;
    const bool interior = (blk_first >= 0) && (blk_last + a.Lpad <= a.n_frames) && (o_blk + OUT_PER_BLOCK <= a.n_out);

    float acc_re[CH_R], acc_im[CH_R];
#pragma unroll
    for (int r = 0; r < CH_R; ++r) acc_re[r] = acc_im[r] = 0.f;

    for (int tc = 0; tc < a.Lpad; tc += CH_TCH) {
        const int cnt = min(CH_TCH, a.Lpad - tc);
        // (block-uniform) a slice of taps whose frames lie in front of the stream for every output of the block -- the
        // start-up outputs of a capture, no history: zeros -- or behind the block's last frame contributes nothing: the
        // head edge of a long filter skips most of its slices this way
        if (a.hist == nullptr && blk_last + tc + cnt <= 0) continue;
        if (blk_first + tc >= a.n_frames) break;
        __syncthreads();
        for (int t = tid * 2; t < cnt; t += CH_THREADS * 2) {
            const float4 g = *reinterpret_cast<const float4 *>(&a.taps[tc + t]);
            *reinterpret_cast<float4 *>(&s_taps[t]) = g;
        }
        __syncthreads();

        if (interior) {
            for (int it = lane0; it < cnt; it += LANE_STRIDE) {
                const float4 g01 = *reinterpret_cast<const float4 *>(&s_taps[it]);
                const float4 g23 = *reinterpret_cast<const float4 *>(&s_taps[it + 2]);
                const float gr[4] = {g01.x, g01.z, g23.x, g23.z};
                const float gi[4] = {g01.y, g01.w, g23.y, g23.w};
#pragma unroll
                for (int r = 0; r < CH_R; ++r) {
                    float xr[4], xi[4];
                    load4<FMT>(a.raw, start[r] + tc + it, xr, xi);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc_re[r] = fmaf(gr[j], xr[j], acc_re[r]);
                        acc_re[r] = fmaf(-gi[j], xi[j], acc_re[r]);
                        acc_im[r] = fmaf(gr[j], xi[j], acc_im[r]);
                        acc_im[r] = fmaf(gi[j], xr[j], acc_im[r]);
                    }
                }
            }
        } else {
            // edge block (history / end of block / ragged tail): same loop, but each lane decides per
            // 4-frame group whether it may use the vector load; only groups that straddle the block
            // boundary fall back to guarded scalar loads.
            for (int it = lane0; it < cnt; it += LANE_STRIDE) {
                const float4 g01 = *reinterpret_cast<const float4 *>(&s_taps[it]);
                const float4 g23 = *reinterpret_cast<const float4 *>(&s_taps[it + 2]);
                const float gr[4] = {g01.x, g01.z, g23.x, g23.z};
                const float gi[4] = {g01.y, g01.w, g23.y, g23.w};
#pragma unroll
                for (int r = 0; r < CH_R; ++r) {
                    if (o0 + r >= a.n_out) continue;
                    const long long f = start[r] + tc + it;
                    float xr[4], xi[4];
                    if (f >= 0 && f + 4 <= a.n_frames) {
                        load4<FMT>(a.raw, f, xr, xi);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            // padded taps are zero, but their frames may lie past the block: never touch them
                            const float2 x = (tc + it + j < a.L) ? load_guarded<FMT>(a, f + j) : make_float2(0.f, 0.f);
                            xr[j] = x.x;
                            xi[j] = x.y;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc_re[r] = fmaf(gr[j], xr[j], acc_re[r]);
                        acc_re[r] = fmaf(-gi[j], xi[j], acc_re[r]);
                        acc_im[r] = fmaf(gr[j], xi[j], acc_im[r]);
                        acc_im[r] = fmaf(gi[j], xr[j], acc_im[r]);
                    }
                }
            }
        }
    }

    // cross-lane reduction; lane r finishes output r
    float my_re = 0.f, my_im = 0.f;
#pragma unroll
    for (int r = 0; r < CH_R; ++r) {
        const float sr = wave_sum(acc_re[r]);
        const float si = wave_sum(acc_im[r]);
        if (lane == r) {
            my_re = sr;
            my_im = si;
        }
    }
    if constexpr (SPLITK) {
        // fixed-order sum of the 8 waves' partial dot products (deterministic)
        if (lane < CH_R) {
            s_part[wave][lane][0] = my_re;
            s_part[wave][lane][1] = my_im;
        }
        __syncthreads();
        if (wave != 0) return;
        if (lane < CH_R) {
            my_re = my_im = 0.f;
#pragma unroll
            for (int w = 0; w < CH_WAVES; ++w) {
                my_re += s_part[w][lane][0];
                my_im += s_part[w][lane][1];
            }
        }
    }
    if (lane < CH_R && o0 + lane < a.n_out) {
        if (a.conj_sum) my_im = -my_im;
        float yr = my_re, yi = my_im;
        if (a.rotate) {
            const unsigned long long m = static_cast<unsigned long long>(a.m_first + o0 + lane);
            const unsigned long long ph = a.rot_base + m * a.rot_step;  // wraps mod 2^64 == mod 1 turn
            const double frac = static_cast<double>(ph >> 11) * (1.0 / 9007199254740992.0);  // 2^-53
            double s, c;
            sincospi(2.0 * frac, &s, &c);
            const float cf = static_cast<float>(c), sf = static_cast<float>(s);
            yr = my_re * cf - my_im * sf;
            yi = my_re * sf + my_im * cf;
        }
        const float zr = yr * a.sc_re - yi * a.sc_im;
        const float zi = yr * a.sc_im + yi * a.sc_re;
        a.out[o0 + lane] = make_float2(zr, zi);
    }
}

template <int FMT>
__global__ void k_history_update(const void *hist, const void *raw, long long n_frames, int keep, void *next)
{
    // next[i] = (hist | raw)[n_frames + i], i in [0, keep): hist occupies [0, keep)
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= keep) return;
    const long long j = n_frames + i;
    constexpr int FB = (FMT == IQA_FMT_S16) ? 4 : (FMT == IQA_FMT_U8 ? 2 : 8);
    const char *src;
    bool zero = false;
    if (j >= keep) {
        src = reinterpret_cast<const char *>(raw) + (j - keep) * FB;
    } else if (hist != nullptr) {
        src = reinterpret_cast<const char *>(hist) + j * FB;
    } else {
        src = nullptr;
        zero = true;
    }
    char *dst = reinterpret_cast<char *>(next) + i * FB;
    if (zero) {
        if constexpr (FMT == IQA_FMT_U8) {
            dst[0] = static_cast<char>(128);  // u8 zero level
            dst[1] = static_cast<char>(128);
        } else {
#pragma unroll
            for (int b = 0; b < FB; ++b) dst[b] = 0;
        }
    } else {
#pragma unroll
        for (int b = 0; b < FB; ++b) dst[b] = src[b];
    }
}

}  // namespace iqa

using namespace iqa;

extern "C" int64_t iqa_taps_padded_len(int32_t ntaps)
{
    if (ntaps <= 0) return 0;
    return (static_cast<int64_t>(ntaps) + CH_TAP_ALIGN - 1) / CH_TAP_ALIGN * CH_TAP_ALIGN;
}

extern "C" int iqa_channelize(const iqa_chan_params *p, const void *taps_dev, const void *raw_dev, int64_t n_frames,
                              int64_t consumed, const void *hist_dev, int64_t m_first, int64_t n_out, void *z_out_dev,
                              void *stream)
{
    if (p == nullptr) return fail_inval("params is NULL");
    if (p->ntaps <= 0) return fail_inval("ntaps must be positive");
    if (p->decimation < 1) return fail_inval("decimation must be >= 1");
    if (frame_bytes(p->fmt) == 0) return fail_inval("unknown sample format");
    if (n_out < 0 || n_frames < 0 || consumed < 0 || m_first < 0) return fail_inval("negative size");
    if (n_out == 0) return IQA_OK;
    if (taps_dev == nullptr || raw_dev == nullptr || z_out_dev == nullptr) return fail_inval("NULL device pointer");
    // host-side shape check before launching a hand-written kernel: the newest frame any
    // output needs must lie inside this block, the oldest inside (hist | raw).
    const int64_t newest = (m_first + n_out - 1) * p->decimation - consumed;
    const int64_t oldest = m_first * static_cast<int64_t>(p->decimation) - consumed - (p->ntaps - 1);
    if (newest >= n_frames) return fail_inval("outputs requested beyond the frames supplied");
    if (newest < 0) return fail_inval("outputs requested before this block");
    if (oldest < -(static_cast<int64_t>(p->ntaps) - 1)) return fail_inval("outputs need frames older than the history");

    ChanArgs a;
    a.taps = static_cast<const float2 *>(taps_dev);
    a.raw = raw_dev;
    a.hist = hist_dev;
    a.out = static_cast<float2 *>(z_out_dev);
    a.n_frames = n_frames;
    a.consumed = consumed;
    a.m_first = m_first;
    a.n_out = n_out;
    a.L = p->ntaps;
    a.Lpad = static_cast<int>(iqa_taps_padded_len(p->ntaps));
    a.D = p->decimation;
    a.conj_sum = p->conj_sum;
    a.rotate = p->rotate;
    a.rot_step = p->rot_step;
    a.rot_base = p->rot_base;
    a.sc_re = p->out_scale_re;
    a.sc_im = p->out_scale_im;

    // short launches cannot fill 256 CUs with 32-output blocks: split the taps across the block's waves instead
    const bool splitk = n_out < 16384;
    const int per_block = splitk ? CH_R : CH_OUT_PER_BLOCK;
    const int64_t blocks = (n_out + per_block - 1) / per_block;
    if (blocks > 0x7fffffffLL) return fail_inval("too many outputs for one launch");
    dim3 grid(static_cast<unsigned>(blocks)), block(CH_THREADS);
    hipStream_t s = as_stream(stream);
    if (splitk) {
        switch (p->fmt) {
            case IQA_FMT_S16: hipLaunchKernelGGL((k_channelize_v1<IQA_FMT_S16, true>), grid, block, 0, s, a); break;
            case IQA_FMT_U8: hipLaunchKernelGGL((k_channelize_v1<IQA_FMT_U8, true>), grid, block, 0, s, a); break;
            default: hipLaunchKernelGGL((k_channelize_v1<IQA_FMT_F32, true>), grid, block, 0, s, a); break;
        }
    } else {
        switch (p->fmt) {
            case IQA_FMT_S16: hipLaunchKernelGGL((k_channelize_v1<IQA_FMT_S16, false>), grid, block, 0, s, a); break;
            case IQA_FMT_U8: hipLaunchKernelGGL((k_channelize_v1<IQA_FMT_U8, false>), grid, block, 0, s, a); break;
            default: hipLaunchKernelGGL((k_channelize_v1<IQA_FMT_F32, false>), grid, block, 0, s, a); break;
        }
    }
    return check_launch("k_channelize_v1");
}

extern "C" int iqa_history_update(int32_t fmt, int32_t ntaps, const void *hist_dev, const void *raw_dev,
                                  int64_t n_frames, void *hist_next_dev, void *stream)
{
    if (frame_bytes(fmt) == 0) return fail_inval("unknown sample format");
    if (ntaps <= 0 || n_frames < 0) return fail_inval("bad sizes");
    const int keep = ntaps - 1;
    if (keep == 0) return IQA_OK;
    if (hist_next_dev == nullptr || (raw_dev == nullptr && n_frames > 0)) return fail_inval("NULL device pointer");
    if (hist_next_dev == hist_dev) return fail_inval("hist_next must not alias hist");
    dim3 grid((keep + 255) / 256), block(256);
    hipStream_t s = as_stream(stream);
    switch (fmt) {
        case IQA_FMT_S16:
            hipLaunchKernelGGL(k_history_update<IQA_FMT_S16>, grid, block, 0, s, hist_dev, raw_dev, (long long)n_frames, keep, hist_next_dev);
            break;
        case IQA_FMT_U8:
            hipLaunchKernelGGL(k_history_update<IQA_FMT_U8>, grid, block, 0, s, hist_dev, raw_dev, (long long)n_frames, keep, hist_next_dev);
            break;
        default:
            hipLaunchKernelGGL(k_history_update<IQA_FMT_F32>, grid, block, 0, s, hist_dev, raw_dev, (long long)n_frames, keep, hist_next_dev);
            break;
    }
    return check_launch("k_history_update");
}

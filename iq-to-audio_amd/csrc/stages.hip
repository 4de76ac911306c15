// stages.hip -- stand-alone NCO mixer stage + library plumbing (errors, version).
//
// Replaces reference src/iq_to_audio/processing.py:289-297 (ComplexOscillator.mix) for
// callers that use the pluggable stages one at a time.  The fused path
// (channelize.hip) never materialises the mixed stream; this kernel exists so that the
// stage API is complete and so that the fused path can be checked stage by stage.
#include "common.h"

#include <algorithm>

#include <cstring>

namespace iqa {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

template <int FMT>
__global__ void k_oscillator_mix(const void *in, long long n, int iq_order, double phase0, double step, float2 *out)
{
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a, b;
    if constexpr (FMT == IQA_FMT_S16) {
        const int v = reinterpret_cast<const int *>(in)[i];
        a = static_cast<float>(static_cast<short>(v & 0xffff)) * (1.0f / 32768.0f);
        b = static_cast<float>(v >> 16) * (1.0f / 32768.0f);
    } else if constexpr (FMT == IQA_FMT_U8) {
        const unsigned short v = reinterpret_cast<const unsigned short *>(in)[i];
        a = (static_cast<float>(v & 0xff) - 128.0f) * (1.0f / 128.0f);
        b = (static_cast<float>(v >> 8) - 128.0f) * (1.0f / 128.0f);
    } else {
        const float2 v = reinterpret_cast<const float2 *>(in)[i];
        a = v.x;
        b = v.y;
    }
    // IQReader._extract_iq (processing.py:268-279): even/odd -> I/Q, optional swap, optional -Q
    float xr = (iq_order & 1) ? b : a;
    float xi = (iq_order & 1) ? a : b;
    if (iq_order & 2) xi = -xi;
    // float64 phase ramp phase0 + step*i, oscillator rounded to complex64, complex64 product
    const double ph = fma(step, static_cast<double>(i), phase0);
    double s, c;
    sincos(ph, &s, &c);
    const float cf = static_cast<float>(c), sf = static_cast<float>(s);
    out[i] = make_float2(xr * cf - xi * sf, xr * sf + xi * cf);
}

// Audio egress without the runtime's blit kernel: hipMemcpyAsync D2H is a copy kernel that fills the GPU with
// waves parked on PCIe stores, and a one-block-per-CU kernel launched beside it waits until they are gone.
// A handful of workgroups writing 1 KiB per wave-instruction straight into mapped pinned host memory move the
// few MB of PCM16 at the same PCIe rate while the channelizer of the next capture runs on all other CUs.
__global__ __launch_bounds__(256) void k_trickle_copy(const uint4 *src, uint4 *dst, long long n16, const unsigned char *src_tail,
                                                       unsigned char *dst_tail, int tail_bytes)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
    if (blockIdx.x == 0 && static_cast<int>(threadIdx.x) < tail_bytes) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

}  // namespace iqa

using namespace iqa;

extern "C" int iqa_trickle_copy(const void *src_dev, void *dst_mapped, int64_t nbytes, int32_t workgroups, void *stream)
{
    if (nbytes < 0) return fail_inval("negative length");
    if (nbytes == 0) return IQA_OK;
    if (!src_dev || !dst_mapped) return fail_inval("NULL pointer");
    if ((reinterpret_cast<uintptr_t>(src_dev) | reinterpret_cast<uintptr_t>(dst_mapped)) & 15) return fail_inval("pointers must be 16-byte aligned");
    if (workgroups < 1) workgroups = 8;
    const int64_t n16 = nbytes / 16;
    const int tail = static_cast<int>(nbytes - n16 * 16);
    hipLaunchKernelGGL(k_trickle_copy, dim3(static_cast<unsigned>(workgroups)), dim3(256), 0, as_stream(stream),
                       static_cast<const uint4 *>(src_dev), static_cast<uint4 *>(dst_mapped), (long long)n16,
                       static_cast<const unsigned char *>(src_dev) + n16 * 16, static_cast<unsigned char *>(dst_mapped) + n16 * 16,
                       tail);
    return check_launch("k_trickle_copy");
}

extern "C" int iqa_abi_version(void) { return IQA_ABI_VERSION; }

extern "C" const char *iqa_last_error(void) { return g_err; }

extern "C" int iqa_oscillator_mix(int32_t fmt, int32_t iq_order, const void *in_dev, int64_t n, double phase0,
                                  double step, void *out_dev, void *stream)
{
    if (frame_bytes(fmt) == 0) return fail_inval("unknown sample format");
    if (iq_order < 0 || iq_order > 3) return fail_inval("Unsupported iq_order");
    if (n < 0) return fail_inval("negative length");
    if (n == 0) return IQA_OK;
    if (!in_dev || !out_dev) return fail_inval("NULL device pointer");
    dim3 grid(static_cast<unsigned>((n + 255) / 256)), block(256);
    hipStream_t s = as_stream(stream);
    switch (fmt) {
        case IQA_FMT_S16:
            hipLaunchKernelGGL(k_oscillator_mix<IQA_FMT_S16>, grid, block, 0, s, in_dev, (long long)n, (int)iq_order, phase0, step, static_cast<float2 *>(out_dev));
            break;
        case IQA_FMT_U8:
            hipLaunchKernelGGL(k_oscillator_mix<IQA_FMT_U8>, grid, block, 0, s, in_dev, (long long)n, (int)iq_order, phase0, step, static_cast<float2 *>(out_dev));
            break;
        default:
            hipLaunchKernelGGL(k_oscillator_mix<IQA_FMT_F32>, grid, block, 0, s, in_dev, (long long)n, (int)iq_order, phase0, step, static_cast<float2 *>(out_dev));
            break;
    }
    return check_launch("k_oscillator_mix");
}

// ---- float32 captures that are integer captures in disguise ---------------------------------------------------------
// SDR software stores int16 / 12-bit / int8 ADC samples as float32 scaled by a power of two (k / 32768, k / 2048,
// k / 128): such a cf32 capture IS an int16 capture and can take the matrix-core channelizers, which need integer data.
// One pass: every value is multiplied by 32768 (exact in float32), written as int16, and the flag word is set if any
// value was NOT an integer in [-32768, 32767] -- then the int16 copy is not the capture and the caller stays on the
// float32 kernel.  The conversion the reference sees is ffmpeg's f32le -> float (no scaling), so nothing changes for it.
namespace iqa {
__global__ __launch_bounds__(256) void k_f32_to_s16_exact(const float4 *in, long long n4, const float *in_tail, int n_tail,
                                                          short *out, int *flag)
{
    bool bad = false;
    auto conv = [&](float x) -> short {
        const float sc = x * 32768.0f;  // a power of two: exact
        const float r = rintf(sc);
        bad |= !(r == sc) || r > 32767.0f || r < -32768.0f;  // (NaN fails r == sc)
        return static_cast<short>(fminf(fmaxf(r, -32768.0f), 32767.0f));
    };
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = in[i];
        short4 o;
        o.x = conv(v.x);
        o.y = conv(v.y);
        o.z = conv(v.z);
        o.w = conv(v.w);
        reinterpret_cast<short4 *>(out)[i] = o;
    }
    if (blockIdx.x == 0 && static_cast<int>(threadIdx.x) < n_tail) out[4 * n4 + threadIdx.x] = conv(in_tail[threadIdx.x]);
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
}  // namespace iqa

// float32 capture -> TWO int16 planes: x = 2^shift (hi + lo / 32768) / 32768 with hi = rint(S x), lo = rint(32768 (S x - hi)),
// S = 2^(15 - shift), |lo| <= 16384 -- exact to 2^(shift - 31) of full scale (every step but the last rint is exact in float32: multiplications by
// powers of two, a difference of neighbours).  The channel filter is linear, so the matrix-core int16 channelizers give
// z = z(hi) + 2^-15 z(lo): what SDR software writes as (u - 127.5) / 127.5 or k / 32767, resampled or filtered recordings,
// any float capture within +-1.  flag bit 0: a value outside [-1, 1 - 2^-16] or a NaN (not representable: the caller
// stays on the float32 kernel); bit 1: some lo != 0 (otherwise the capture IS an int16 capture and one pass suffices).
namespace iqa {
__global__ __launch_bounds__(256) void k_f32_split_s16(const float4 *in, long long n4, const float *in_tail, int n_tail, float scale,
                                                       short *hi_out, short *lo_out, int *flag)
{
    int bits = 0;
    auto conv = [&](float x, short &hi, short &lo) {
        const float sc = x * scale;  // a power of two: exact
        const float h = rintf(sc);
        if (!(h >= -32768.0f && h <= 32767.0f)) bits |= 1;  // (NaN fails both comparisons)
        const float r = (sc - h) * 32768.0f;  // |sc - h| <= 0.5, exact; times a power of two, exact
        const float l = rintf(r);
        if (l != 0.0f) bits |= 2;
        hi = static_cast<short>(fminf(fmaxf(h, -32768.0f), 32767.0f));
        lo = static_cast<short>(fminf(fmaxf(l, -16384.0f), 16384.0f));
    };
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = in[i];
        short4 h, l;
        conv(v.x, h.x, l.x);
        conv(v.y, h.y, l.y);
        conv(v.z, h.z, l.z);
        conv(v.w, h.w, l.w);
        reinterpret_cast<short4 *>(hi_out)[i] = h;
        reinterpret_cast<short4 *>(lo_out)[i] = l;
    }
    if (blockIdx.x == 0 && static_cast<int>(threadIdx.x) < n_tail) {
        short h, l;
        conv(in_tail[threadIdx.x], h, l);
        hi_out[4 * n4 + threadIdx.x] = h;
        lo_out[4 * n4 + threadIdx.x] = l;
    }
    const unsigned long long b0 = __ballot(bits & 1), b1 = __ballot(bits & 2);
    if ((b0 | b1) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, (b0 ? 1 : 0) | (b1 ? 2 : 0));
}
}  // namespace iqa

extern "C" int iqa_f32_split_s16(const void *f32_dev, int64_t n_values, int32_t shift, void *hi_out_dev, void *lo_out_dev, void *flag_dev,
                                 void *stream)
{
    if (n_values < 0) return fail_inval("negative length");
    if (shift < 0 || shift > 15) return fail_inval("shift must be in 0..15");
    if (n_values == 0) return IQA_OK;
    if (!f32_dev || !hi_out_dev || !lo_out_dev || !flag_dev) return fail_inval("NULL device pointer");
    if ((reinterpret_cast<uintptr_t>(f32_dev) & 15) || (reinterpret_cast<uintptr_t>(hi_out_dev) & 7) || (reinterpret_cast<uintptr_t>(lo_out_dev) & 7))
        return fail_inval("input must be 16-byte aligned, outputs 8-byte aligned");
    const long long n4 = n_values / 4;
    const int n_tail = static_cast<int>(n_values - 4 * n4);
    const float *tail = static_cast<const float *>(f32_dev) + 4 * n4;
    const unsigned blocks = static_cast<unsigned>(std::min<long long>(std::max<long long>((n4 + 255) / 256, 1), 256 * 16));
    hipLaunchKernelGGL(iqa::k_f32_split_s16, dim3(blocks), dim3(256), 0, iqa::as_stream(stream), static_cast<const float4 *>(f32_dev), n4, tail,
                       n_tail, std::ldexp(1.0f, 15 - shift), static_cast<short *>(hi_out_dev), static_cast<short *>(lo_out_dev), static_cast<int *>(flag_dev));
    return iqa::check_launch("k_f32_split_s16");
}

extern "C" int iqa_f32_to_s16_exact(const void *f32_dev, int64_t n_values, void *s16_out_dev, void *flag_dev, void *stream)
{
    if (n_values < 0) return fail_inval("negative length");
    if (n_values == 0) return IQA_OK;
    if (!f32_dev || !s16_out_dev || !flag_dev) return fail_inval("NULL device pointer");
    if ((reinterpret_cast<uintptr_t>(f32_dev) & 15) || (reinterpret_cast<uintptr_t>(s16_out_dev) & 7))
        return fail_inval("input must be 16-byte aligned, output 8-byte aligned");
    const long long n4 = n_values / 4;
    const int n_tail = static_cast<int>(n_values - 4 * n4);
    const float *tail = static_cast<const float *>(f32_dev) + 4 * n4;
    const unsigned blocks = static_cast<unsigned>(std::min<long long>(std::max<long long>((n4 + 255) / 256, 1), 256 * 16));
    hipLaunchKernelGGL(iqa::k_f32_to_s16_exact, dim3(blocks), dim3(256), 0, iqa::as_stream(stream), static_cast<const float4 *>(f32_dev), n4,
                       tail, n_tail, static_cast<short *>(s16_out_dev), static_cast<int *>(flag_dev));
    return iqa::check_launch("k_f32_to_s16_exact");
}

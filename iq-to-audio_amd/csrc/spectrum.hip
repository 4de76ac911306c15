// spectrum.hip -- windowed power spectra of a complex stream.
//
// Replaces reference src/iq_to_audio/spectrum.py: _SlidingFFT.psd (:143-171), compute_psd (:15-45) and the frame
// loop of streaming_waterfall (:58-93) for a batch of frames at once:
//
//   frame f = samples[first + f*hop : first + f*hop + use]          (use <= nfft; compute_psd zero-pads short input)
//   X = FFT_nfft( complex128(frame) * window )                      (float64, as the reference: numpy promotes)
//   psd_db[f][k] = 10 log10( |X[k']|^2 / scale + 1e-18 ),  k' = (k + nfft/2) mod nfft      (fftshift, :161/:37)
//
// The FFT is rocFFT's (through hipFFT, double complex, batched, in place); windowing + ingest conversion and the
// |.|^2 / log / fftshift epilogue are the two kernels around it.  Plans are cached per (device, nfft, batch) in a small
// LRU behind a mutex -- beside the per-kernel "LDS limit raised on device d" bits and the thread-local error string the
// only state the library keeps; the mutex is held while a plan's stream is set and its FFT enqueued.
#include "common.h"

#include <hipfft/hipfft.h>

#include <list>
#include <mutex>
#include <utility>

namespace iqa {

template <int FMT>
__device__ __forceinline__ void load_sample(const void *in, long long i, int iq_order, double &re, double &im)
{
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
    re = static_cast<double>(xr);
    im = static_cast<double>(xi);
}

// work[f][i] = complex128(sample[first + f*hop + i]) * window[i] for i < use, 0 for use <= i < nfft
template <int FMT>
__global__ __launch_bounds__(256) void k_psd_window(const void *in, long long first, long long hop, int use, int nfft,
                                                     int iq_order, const double *window, double2 *work)
{
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nfft) return;
    double re = 0.0, im = 0.0;
    if (i < use) {
        load_sample<FMT>(in, first + static_cast<long long>(f) * hop + i, iq_order, re, im);
        const double w = window[i];
        re *= w;
        im *= w;
    }
    work[static_cast<long long>(f) * nfft + i] = make_double2(re, im);
}

// out[f][k] = 10 log10(|X[f][(k + nfft/2) mod nfft]|^2 / scale + 1e-18); optionally sum_db[k] += out[f][k] over the batch
__global__ __launch_bounds__(256) void k_psd_finish(const double2 *work, int nfft, int n_frames, double inv_scale,
                                                     double *out, float *out_f32, double *sum_db)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nfft) return;
    const int src = (k + ((nfft + 1) >> 1)) % nfft;  // numpy fftshift: out[k] = in[(k - n//2) mod n]
    double acc = 0.0;
    for (int f = 0; f < n_frames; ++f) {
        const double2 x = work[static_cast<long long>(f) * nfft + src];
        const double p = (x.x * x.x + x.y * x.y) * inv_scale;
        const double db = 10.0 * log10(fabs(p) + 1e-18);
        if (out) out[static_cast<long long>(f) * nfft + k] = db;
        if (out_f32) out_f32[static_cast<long long>(f) * nfft + k] = static_cast<float>(db);
        acc += db;
    }
    if (sum_db) sum_db[k] += acc;
}

// Waterfall reduction (_WaterfallAggregator._maybe_reduce, spectrum.py:190-208): rows pairwise averaged in float64
// and rounded back to float32; an odd last row is copied.
__global__ __launch_bounds__(256) void k_pair_average(const float *in, int n_rows, int n_cols, float *out)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;  // output row
    if (c >= n_cols) return;
    const long long a = static_cast<long long>(2 * r) * n_cols + c;
    float v = in[a];
    if (2 * r + 1 < n_rows) v = static_cast<float>((static_cast<double>(v) + static_cast<double>(in[a + n_cols])) / 2.0);
    out[static_cast<long long>(r) * n_cols + c] = v;
}

// rocFFT plans are expensive (milliseconds) and bound to the device they were made on: a small LRU keyed by
// (device, nfft, batch).  A plan carries its stream, so the lock is held from hipfftSetStream to the end of
// hipfftExec* (both are host-side enqueues; the FFT itself runs asynchronously on the caller's stream).  Evicted
// plans are destroyed.
struct PlanCache {
    static constexpr size_t kMax = 8;
    struct Entry {
        int dev, nfft, batch;
        hipfftHandle handle;
        hipStream_t last_stream;  // where this plan's most recent transform was enqueued (hipfftSetStream)
    };
    std::mutex mu;
    std::list<Entry> lru;  // front = most recently used
};
static PlanCache &plan_cache()
{
    static PlanCache c;
    return c;
}

// with c.mu held
static int get_plan_locked(PlanCache &c, int nfft, int batch, hipfftHandle *out)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (auto it = c.lru.begin(); it != c.lru.end(); ++it) {
        if (it->dev == dev && it->nfft == nfft && it->batch == batch) {
            c.lru.splice(c.lru.begin(), c.lru, it);
            *out = c.lru.front().handle;
            return IQA_OK;
        }
    }
    hipfftHandle h;
    int n[1] = {nfft};
    const hipfftResult r = hipfftPlanMany(&h, 1, n, nullptr, 1, nfft, nullptr, 1, nfft, HIPFFT_Z2Z, batch);
    if (r != HIPFFT_SUCCESS) {
        set_error("hipfftPlanMany(nfft=%d, batch=%d) failed: %d", nfft, batch, static_cast<int>(r));
        return IQA_EHIP;
    }
    c.lru.push_front({dev, nfft, batch, h, nullptr});
    while (c.lru.size() > PlanCache::kMax) {
        // the evicted plan's last transform may still be running on its stream: wait for that stream before the plan's
        // work buffers go away (an eviction is rare: more than kMax distinct (nfft, batch) shapes in use)
        int cur = 0;
        (void)hipGetDevice(&cur);
        const PlanCache::Entry &old = c.lru.back();
        if (old.dev != cur) (void)hipSetDevice(old.dev);
        (void)hipStreamSynchronize(old.last_stream);
        if (old.dev != cur) (void)hipSetDevice(cur);
        (void)hipfftDestroy(old.handle);
        c.lru.pop_back();
    }
    *out = h;
    return IQA_OK;
}

}  // namespace iqa

using namespace iqa;

extern "C" int iqa_psd_frames(int32_t fmt, int32_t iq_order, const void *samples_dev, int64_t n_samples, int64_t first,
                              int64_t hop, int32_t n_frames, int32_t nfft, int32_t use, const void *window_dev, double scale,
                              void *work_dev, void *psd_db_dev, void *psd_db_f32_dev, void *sum_db_dev, void *stream)
{
    if (frame_bytes(fmt) == 0) return fail_inval("unknown sample format");
    if (iq_order < 0 || iq_order > 3) return fail_inval("Unsupported iq_order");
    if (nfft < 2 || use < 1 || use > nfft) return fail_inval("need 1 <= use <= nfft, nfft >= 2");
    if (n_frames < 0 || hop < 1 || first < 0) return fail_inval("bad frame geometry");
    if (n_frames == 0) return IQA_OK;
    if (first + static_cast<int64_t>(n_frames - 1) * hop + use > n_samples) return fail_inval("frames reach past the samples");
    if (!samples_dev || !window_dev || !work_dev) return fail_inval("NULL device pointer");
    if (!(scale > 0.0)) return fail_inval("scale must be positive");
    hipStream_t s = as_stream(stream);
    const dim3 grid((nfft + 255) / 256, n_frames), block(256);
    double2 *work = static_cast<double2 *>(work_dev);
    const double *win = static_cast<const double *>(window_dev);
    switch (fmt) {
        case IQA_FMT_S16:
            hipLaunchKernelGGL(k_psd_window<IQA_FMT_S16>, grid, block, 0, s, samples_dev, (long long)first, (long long)hop, (int)use, (int)nfft, (int)iq_order, win, work);
            break;
        case IQA_FMT_U8:
            hipLaunchKernelGGL(k_psd_window<IQA_FMT_U8>, grid, block, 0, s, samples_dev, (long long)first, (long long)hop, (int)use, (int)nfft, (int)iq_order, win, work);
            break;
        default:
            hipLaunchKernelGGL(k_psd_window<IQA_FMT_F32>, grid, block, 0, s, samples_dev, (long long)first, (long long)hop, (int)use, (int)nfft, (int)iq_order, win, work);
            break;
    }
    if (check_launch("k_psd_window") != IQA_OK) return IQA_EHIP;
    {
        PlanCache &cache = plan_cache();
        std::lock_guard<std::mutex> lock(cache.mu);
        hipfftHandle plan;
        const int rc = get_plan_locked(cache, nfft, n_frames, &plan);
        if (rc != IQA_OK) return rc;
        if (hipfftSetStream(plan, s) != HIPFFT_SUCCESS) {
            set_error("hipfftSetStream failed");
            return IQA_EHIP;
        }
        cache.lru.front().last_stream = s;  // (get_plan_locked moved this plan to the front)
        if (hipfftExecZ2Z(plan, reinterpret_cast<hipfftDoubleComplex *>(work), reinterpret_cast<hipfftDoubleComplex *>(work),
                          HIPFFT_FORWARD) != HIPFFT_SUCCESS) {
            set_error("hipfftExecZ2Z failed");
            return IQA_EHIP;
        }
    }
    hipLaunchKernelGGL(k_psd_finish, dim3((nfft + 255) / 256), block, 0, s, work, (int)nfft, (int)n_frames, 1.0 / scale,
                       static_cast<double *>(psd_db_dev), static_cast<float *>(psd_db_f32_dev), static_cast<double *>(sum_db_dev));
    return check_launch("k_psd_finish");
}

extern "C" int iqa_pair_average_rows(const void *rows_dev, int32_t n_rows, int32_t n_cols, void *out_dev, void *stream)
{
    if (n_rows < 0 || n_cols < 1) return fail_inval("bad matrix shape");
    if (n_rows == 0) return IQA_OK;
    if (!rows_dev || !out_dev) return fail_inval("NULL device pointer");
    const int out_rows = (n_rows + 1) / 2;
    hipLaunchKernelGGL(k_pair_average, dim3((n_cols + 255) / 256, out_rows), dim3(256), 0, as_stream(stream),
                       static_cast<const float *>(rows_dev), (int)n_rows, (int)n_cols, static_cast<float *>(out_dev));
    return check_launch("k_pair_average");
}

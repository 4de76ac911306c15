// mfma_common.h -- types, argument block and the output emission shared by the int8-MFMA channelizer
// kernels (channelize_mfma.hip: per-lane and per-wave-staged data paths; channelize_ring.hip: block-wide
// contiguous LDS-DMA ring).  See channelize_mfma.hip for the mathematics.
#pragma once

#include "common.h"

namespace iqa {

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef v4i_t v4i_a4 __attribute__((aligned(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));

constexpr int MF_WAVES = 8;  // one block per CU, two waves per SIMD, tap fragments shared by all eight
constexpr int MF_THREADS = MF_WAVES * kWave;
constexpr int MF_Q = 64;          // q slots per output component (needs ceil(L/D) <= 64)
constexpr int MF_ROWTILES = 4;    // 2 components x 64 q = 128 rows
constexpr int MF_KSTEP_BYTES = MF_ROWTILES * 2 * 1024;  // tap fragments per k step

struct MfmaArgs {
    const v4i_t *afrag;  // [ksteps][rowtile 4][piece 2][lane 64] 16-byte tap fragments
    const int *raw;      // capture frames as dwords (lo half = I, hi half = Q)
    float2 *out;         // out[i] = z[m_lo + i]
    long long consumed, m_lo, n_out;
    int D, ksteps, range, debug;
    int k_first;      // first k step of this pass (K split over passes when the tap fragments exceed LDS)
    int col_shift;    // 64 * q-group of this pass (filters with ceil(L/D) > 64 are split into q-groups)
    int finalize;     // 1: add partial_in, rotate/scale and store z; 0: store the raw sums to partial_out
    const double2 *partial_in;
    double2 *partial_out;
    unsigned long long *stamps;  // diagnostics only (debug bit 1): per-wave cycle anatomy
    double unit, c_re, c_im;
    int conj_sum, rotate;
    unsigned long long rot_step, rot_base;
    float sc_re, sc_im;
    double rot64_re, rot64_im;  // exp(j*2*pi*64*rot_step): rotation between outputs 64 apart (ring kernel emission)
    int raw_partials;  // ring kernels, int32 sums, finalize == 0, no partial_in: partial_out holds int2 {256*S1+S2 re, im} (8 B per output)
    int high_taps_only;  // ring kernels, int16 data: the low tap byte q2 is zero throughout (the first lane of a "fine" / "full"
                         // tap-row group, dsp_plan.plan_mfma(residual=True)): the q2*hi MFMA of every k step is skipped
    // lane pairs (ring kernel, two lanes per workgroup): this lane's own tile t is staged in round t + pair_shift (the
    // stream is the one of the pair's lane with the LARGER tap-row group, which starts 2 tiles earlier per group), and
    // the workgroup runs pair_extra rounds beyond a lane's own tiles
    int pair_shift, pair_extra;
    // lane pairs: pacing of the workgroups that stream the same range (see ring_main): a word per (range, unit) in a
    // library-owned device buffer, tagged with this launch's token
    unsigned int *pace;
    unsigned int pace_token;
    int pace_slot, pace_units;
};

// The last steps of every matrix-core emission, written with explicit roundings so that the kernels that share them
// (per-lane kernel, ring kernels, the combine kernel of the multi-lane path) give the SAME bits for the same sums --
// left to the compiler, `a*b + c*d` contracts to an fma in one kernel and not in another.
__device__ __forceinline__ double mfma_scaled_sum(double v, double c, double unit)  // (256 v + c) * unit
{
    return __dmul_rn(__fma_rn(v, 256.0, c), unit);
}
// conjugation, rotation and ingest-order scaling in FLOAT64, ONE rounding to float32 at the very end: with the "full"
// precision's sums (error ~1e-9 of full scale) z then carries the same float32 rounding as the reference's own
// complex128 -> complex64 cast (processing.py:339) on all but a few per cent of the samples; rotating in float32 after an
// early cast left ~4e-8 RMS (three roundings at |z| ~ 0.7), which the reference's SSB AGC amplifies (decoders/ssb.py:75-77).
__device__ __forceinline__ float2 mfma_finish(double d_re, double d_im, int conj_sum, int rotate, double cw, double sw, float sc_re,
                                              float sc_im)
{
    if (conj_sum) d_im = -d_im;
    double yr = d_re, yi = d_im;
    if (rotate) {
        yr = __fma_rn(d_re, cw, -__dmul_rn(d_im, sw));
        yi = __fma_rn(d_re, sw, __dmul_rn(d_im, cw));
    }
    // (out_scale is one of 1, j, -j: exact)
    const double sr = static_cast<double>(sc_re), si = static_cast<double>(sc_im);
    return make_float2(static_cast<float>(__fma_rn(yr, sr, -__dmul_rn(yi, si))), static_cast<float>(__fma_rn(yr, si, __dmul_rn(yi, sr))));
}

// emission shared by both kernels: output m0+i sits at position 64+i of the S1/S2 arrays
template <int THREADS>
__device__ __forceinline__ void mfma_emit(const MfmaArgs &a, const int *s_acc, int acc_len, int cnt, long long i0,
                                          long long m0, int tid)
{
    for (int i = tid; i < cnt; i += THREADS) {
        const int pos = MF_Q + i;
        const double s1r = s_acc[pos], s1i = s_acc[acc_len + pos];
        const double s2r = s_acc[2 * acc_len + pos], s2i = s_acc[3 * acc_len + pos];
        double d_re = mfma_scaled_sum(s1r * 256.0 + s2r, a.c_re, a.unit);  // (65536 S1 + 256 S2 + c) * unit, exact up to the product
        double d_im = mfma_scaled_sum(s1i * 256.0 + s2i, a.c_im, a.unit);
        if (a.partial_in != nullptr) {
            const double2 pr = a.partial_in[i0 + i];
            d_re = __dadd_rn(d_re, pr.x);
            d_im = __dadd_rn(d_im, pr.y);
        }
        if (!a.finalize) {
            a.partial_out[i0 + i] = make_double2(d_re, d_im);
            continue;
        }
        double cw = 1.0, sw = 0.0;
        if (a.rotate) {
            const unsigned long long m = static_cast<unsigned long long>(m0 + i);
            const unsigned long long ph = a.rot_base + m * a.rot_step;
            const double frac = static_cast<double>(ph >> 11) * (1.0 / 9007199254740992.0);
            sincospi(2.0 * frac, &sw, &cw);
        }
        a.out[i0 + i] = mfma_finish(d_re, d_im, a.conj_sum, a.rotate, cw, sw, a.sc_re, a.sc_im);
    }
}

// the ring kernels (channelize_ring.hip)
int mfma_ring_mode(int decimation, int k_first, int k_count, bool acc64, bool u8);  // 0 none, 1 contiguous, 2 row-staged slots
bool mfma_ring_supported(int decimation);                                  // mode 1 possible for this decimation
size_t mfma_ring_lds_bytes(int ksteps, bool rows, bool u8);                // LDS of a block: data ring + window of sums
int mfma_ring_launch(const MfmaArgs &a, unsigned blocks, size_t lds_bytes, hipStream_t stream, bool rows, bool u8);  // IQA_* status

// one (channel, tap-row group) of a multi-lane launch: the per-lane part of MfmaArgs
struct MfmaLane {
    const v4i_t *afrag;
    float2 *out;
    const double2 *partial_in;
    double2 *partial_out;
    double unit, c_re, c_im;
    unsigned long long rot_step, rot_base;
    double rot64_re, rot64_im;
    float sc_re, sc_im;
    int col_shift, finalize, conj_sum, rotate, raw_partials, high_taps_only;
};
int mfma_ring_launch_multi(const MfmaArgs &common, const MfmaLane *lanes, int n_lanes, size_t lds_bytes, hipStream_t stream, bool rows,
                           bool u8, unsigned *blocks_out, bool pairs = false, bool acc64 = false);
bool mfma_ring_pairs_supported(int decimation, int k_first, int k_count, bool u8, bool acc64 = false);  // two lanes per workgroup (contiguous slots, no loader waves)

}  // namespace iqa

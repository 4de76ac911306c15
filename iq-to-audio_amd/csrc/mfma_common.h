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
};

// emission shared by both kernels: output m0+i sits at position 64+i of the S1/S2 arrays
template <int THREADS>
__device__ __forceinline__ void mfma_emit(const MfmaArgs &a, const int *s_acc, int acc_len, int cnt, long long i0,
                                          long long m0, int tid)
{
    for (int i = tid; i < cnt; i += THREADS) {
        const int pos = MF_Q + i;
        const double s1r = s_acc[pos], s1i = s_acc[acc_len + pos];
        const double s2r = s_acc[2 * acc_len + pos], s2i = s_acc[3 * acc_len + pos];
        double d_re = (s1r * 65536.0 + s2r * 256.0 + a.c_re) * a.unit;
        double d_im = (s1i * 65536.0 + s2i * 256.0 + a.c_im) * a.unit;
        if (a.partial_in != nullptr) {
            const double2 pr = a.partial_in[i0 + i];
            d_re += pr.x;
            d_im += pr.y;
        }
        if (!a.finalize) {
            a.partial_out[i0 + i] = make_double2(d_re, d_im);
            continue;
        }
        float my_re = static_cast<float>(d_re);
        float my_im = static_cast<float>(d_im);
        if (a.conj_sum) my_im = -my_im;
        float yr = my_re, yi = my_im;
        if (a.rotate) {
            const unsigned long long m = static_cast<unsigned long long>(m0 + i);
            const unsigned long long ph = a.rot_base + m * a.rot_step;
            const double frac = static_cast<double>(ph >> 11) * (1.0 / 9007199254740992.0);
            double s, c;
            sincospi(2.0 * frac, &s, &c);
            const float cf = static_cast<float>(c), sf = static_cast<float>(s);
            yr = my_re * cf - my_im * sf;
            yi = my_re * sf + my_im * cf;
        }
        a.out[i0 + i] = make_float2(yr * a.sc_re - yi * a.sc_im, yr * a.sc_im + yi * a.sc_re);
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
    int col_shift, finalize, conj_sum, rotate;
};
int mfma_ring_launch_multi(const MfmaArgs &common, const MfmaLane *lanes, int n_lanes, size_t lds_bytes, hipStream_t stream, bool rows,
                           bool u8, unsigned *blocks_out);

}  // namespace iqa

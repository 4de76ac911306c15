// channelize_mfma.hip -- int8-MFMA form of the fused ingest+mix+FIR+decimate kernel (gfx950).
//
// Same mathematics as channelize.hip (reference processing.py:268-279, 289-297, 325-346,
// 354-360): z[m] = rot(m) * sum_k g[k] x[mD-k].  Here the decimating FIR is recast as a
// dense integer GEMM so that it runs on the matrix cores at the int8 rate, with exact
// integer accumulation:
//
//   rows of D frames:  X[b][kap] = v[2(bD+1) + kap],  kap in [0,2D)  (raw int16 I,Q,I,Q,... --
//                      row b IS a contiguous run of the capture, rows are 2D values apart)
//   tap rows:          W[q][rho] = g[qD-1-rho], q = 1..P (P = ceil(L/D) <= 64), real form
//                      A[(re,q)][2rho]=Re, [2rho+1]=-Im ; A[(im,q)][2rho]=Im, [2rho+1]=Re
//   GEMM:              G[row][b] = sum_kap A[row][kap] * X[b][kap]        (128 x Ncols x 2D)
//   diagonal sum:      S[m] = sum_q G[(.,q)][m-q]
//
// int16 data are split exactly into bytes v = 256*hi + lo' + 128 (hi = v>>8, lo' = (v&255)-128,
// both int8), taps are quantised to 16-bit fixed point T = 256*q1 + q2 (q1,q2 int8, one global
// unit u), and three int8 MFMAs per (row tile, k step) accumulate
//      ACC1 += q1*hi        ACC2 += q1*lo' + q2*hi        (q2*lo' dropped: < 1e-6 of full scale)
// so that  S = u*(65536*S1 + 256*S2 + 128*sum(T)).  All accumulation up to S1/S2 is exact
// int32 (|S1| <= 2^14 L, |S2| <= 2^15 L < 2^31 for L <= 32769), hence bit-reproducible.
//
// Mapping: block = 4 waves; a block owns `range` consecutive outputs and walks the
// range+63 data columns that touch them in tiles of 32 columns, one tile per wave at a time.
// Per k step (32 int8 along K) a wave issues 2 unaligned global_load_dwordx4 per lane for the
// data fragment (straight from HBM/L2 -- no LDS staging: a row is contiguous memory), splits
// hi/lo bytes with v_perm_b32, reads 8 tap fragments from LDS (pre-swizzled by the host into
// fragment order, conflict-free ds_read_b128) and issues 12 v_mfma_i32_32x32x32_i8.
// The G tile never leaves registers except as ds_add_u32 into the block's S1/S2 arrays
// (address = lane part + immediate), which are converted, rotated and stored once at the end.
//
// Lane maps of v_mfma_i32_32x32x32_i8 were verified on hardware with probe/mfma_i8_probe.hip:
//   A[row=l&31][k=16(l>>5)+j], B[k=16(l>>5)+j][col=l&31], C: col=l&31, row=(r&3)+8(r>>2)+4(l>>5).
#include "mfma_common.h"

#include <atomic>
#include <cmath>
#include <type_traits>

namespace iqa {

__global__ __launch_bounds__(MF_THREADS, 2) void k_channelize_mfma_s16(MfmaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;

    const long long i0 = static_cast<long long>(blockIdx.x) * a.range;  // first output of this block (relative)
    const int cnt = static_cast<int>(min(static_cast<long long>(a.range), a.n_out - i0));
    const long long m0 = a.m_lo + i0;
    const int tiles = (cnt + 63 + 31) >> 5;  // data columns b in [m0-64, m0+cnt-2], rounded up to tiles of 32
    const int acc_len = tiles * 32 + MF_Q + 4;

    const bool stamp = (a.debug & 2) && a.stamps != nullptr;
    unsigned long long t_begin = 0, t_loop0 = 0, t_scatter = 0, t_loop1 = 0;
    if (stamp) t_begin = __builtin_amdgcn_s_memtime();
    v4i_t *s_a = reinterpret_cast<v4i_t *>(smem);
    int *s_acc = reinterpret_cast<int *>(smem + static_cast<size_t>(a.ksteps) * MF_KSTEP_BYTES);
    // s_acc layout: [S1re | S1im | S2re | S2im], each acc_len ints
    {
        // tap fragments -> LDS, eight loads in flight per thread (a plain load/store loop serialises on latency)
        const int n16 = a.ksteps * (MF_KSTEP_BYTES / 16);
        for (int i0f = tid; i0f < n16; i0f += MF_THREADS * 8) {
            v4i_t tmp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0f + u * MF_THREADS;
                tmp[u] = (i < n16) ? a.afrag[i] : v4i_t{0, 0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0f + u * MF_THREADS;
                if (i < n16) s_a[i] = tmp[u];
            }
        }
    }
    for (int i = tid; i < 4 * acc_len; i += MF_THREADS) s_acc[i] = 0;
    __syncthreads();

    // Software pipeline over the flat sequence of (tile, k step) pairs of this wave, unrolled by two with
    // two static data slots A/B: right after a slot has been consumed (byte split) the load of the pair two
    // steps ahead is issued into the SAME registers, and each row tile's tap fragments for the next pair
    // are requested from LDS right after that row tile's three MFMAs.  No register is copied while its
    // load is in flight (a copy would force the wait the pipeline is there to avoid).
    int pf_tile = wave, pf_ks = 0;
    auto row_ptr = [&](int tile) -> const int * {
        const long long b = m0 - MF_Q - a.col_shift + tile * 32 + col;  // this lane's data column (global row index)
        return a.raw + (b * a.D + 1 - a.consumed) + 16 * a.k_first + 8 * h;  // dword index of kap = 32 k_first + 16h
    };
    const int *pf_row = row_ptr(min(pf_tile, tiles - 1));
#define MF_LOAD_PAIR(X0, X1)                                                        \
    {                                                                               \
        X0 = *reinterpret_cast<const v4i_a4 *>(pf_row + 16 * pf_ks);                \
        X1 = *reinterpret_cast<const v4i_a4 *>(pf_row + 16 * pf_ks + 4);            \
        if (++pf_ks == a.ksteps) {                                                  \
            pf_ks = 0;                                                              \
            pf_tile += MF_WAVES;                                                    \
            pf_row = row_ptr(min(pf_tile, tiles - 1)); /* past the end: re-read */  \
        }                                                                           \
    }
    v4i_t a0, a1, b0, b1;
    MF_LOAD_PAIR(a0, a1);
    MF_LOAD_PAIR(b0, b1);
    v4i_t fr[2 * MF_ROWTILES];
    {
        const v4i_t *fa = s_a + lane;
#pragma unroll
        for (int i = 0; i < 2 * MF_ROWTILES; ++i) fr[i] = fa[i * 64];
    }
    v16i_t acc1[MF_ROWTILES], acc2[MF_ROWTILES];
#pragma unroll
    for (int rt = 0; rt < MF_ROWTILES; ++rt) {
        acc1[rt] = v16i_t{0};
        acc2[rt] = v16i_t{0};
    }
    const v16i_t zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (stamp) t_loop0 = __builtin_amdgcn_s_memtime();
    int t = wave, ks = 0;  // the pair being multiplied
    const bool skip_scatter = (a.debug & 1) != 0;

#define MF_PAIR(X0, X1)                                                                              \
    {                                                                                                \
        v4i_t hi, lo;                                                                                \
        hi.x = __builtin_amdgcn_perm(X0.y, X0.x, 0x07050301);                                        \
        hi.y = __builtin_amdgcn_perm(X0.w, X0.z, 0x07050301);                                        \
        hi.z = __builtin_amdgcn_perm(X1.y, X1.x, 0x07050301);                                        \
        hi.w = __builtin_amdgcn_perm(X1.w, X1.z, 0x07050301);                                        \
        lo.x = __builtin_amdgcn_perm(X0.y, X0.x, 0x06040200) ^ 0x80808080;                           \
        lo.y = __builtin_amdgcn_perm(X0.w, X0.z, 0x06040200) ^ 0x80808080;                           \
        lo.z = __builtin_amdgcn_perm(X1.y, X1.x, 0x06040200) ^ 0x80808080;                           \
        lo.w = __builtin_amdgcn_perm(X1.w, X1.z, 0x06040200) ^ 0x80808080;                           \
        MF_LOAD_PAIR(X0, X1);                                                                        \
        const int kn = (ks + 1 < a.ksteps) ? ks + 1 : 0;                                             \
        const v4i_t *fa = s_a + kn * (MF_KSTEP_BYTES / 16) + lane;                                   \
        if (ks == 0) { /* first step of a tile: C = 0 constant, no accumulator clearing needed */     \
            _Pragma("unroll") for (int rt = 0; rt < MF_ROWTILES; ++rt)                               \
            {                                                                                        \
                acc1[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr[2 * rt], hi, zero16, 0, 0, 0);   \
                acc2[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr[2 * rt], lo, zero16, 0, 0, 0);   \
                acc2[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr[2 * rt + 1], hi, acc2[rt], 0, 0, 0); \
                fr[2 * rt] = fa[(2 * rt) * 64];                                                      \
                fr[2 * rt + 1] = fa[(2 * rt + 1) * 64];                                              \
            }                                                                                        \
        } else {                                                                                     \
            _Pragma("unroll") for (int rt = 0; rt < MF_ROWTILES; ++rt)                               \
            {                                                                                        \
                acc1[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr[2 * rt], hi, acc1[rt], 0, 0, 0); \
                acc2[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr[2 * rt], lo, acc2[rt], 0, 0, 0); \
                acc2[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr[2 * rt + 1], hi, acc2[rt], 0, 0, 0); \
                fr[2 * rt] = fa[(2 * rt) * 64];                                                      \
                fr[2 * rt + 1] = fa[(2 * rt + 1) * 64];                                              \
            }                                                                                        \
        }                                                                                            \
        if (++ks == a.ksteps) { /* tile complete: diagonal scatter into the block's S1/S2 arrays */   \
            unsigned long long ts0 = 0;                                                              \
            if (stamp) ts0 = __builtin_amdgcn_s_memtime();                                           \
            if (!skip_scatter) {                                                                     \
                int *base = s_acc + (t * 32 + col + 4 * h + 1);                                      \
                _Pragma("unroll") for (int rt = 0; rt < MF_ROWTILES; ++rt)                           \
                {                                                                                    \
                    int *p1 = base + (rt >> 1) * acc_len + (rt & 1) * 32;                            \
                    int *p2 = p1 + 2 * acc_len;                                                      \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r)                                   \
                    {                                                                                \
                        const int off = (r & 3) + 8 * (r >> 2);                                      \
                        atomicAdd(p1 + off, acc1[rt][r]);                                            \
                        atomicAdd(p2 + off, acc2[rt][r]);                                            \
                    }                                                                                \
                }                                                                                    \
            } else {                                                                                 \
                _Pragma("unroll") for (int rt = 0; rt < MF_ROWTILES; ++rt)                           \
                    asm volatile("" ::"v"(acc1[rt]), "v"(acc2[rt]));                                 \
            }                                                                                        \
            if (stamp) {                                                                             \
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                   \
                t_scatter += __builtin_amdgcn_s_memtime() - ts0;                                     \
            }                                                                                        \
            ks = 0;                                                                                  \
            t += MF_WAVES;                                                                           \
        }                                                                                            \
    }

    while (t < tiles) {
        MF_PAIR(a0, a1);
        if (t >= tiles) break;
        MF_PAIR(b0, b1);
    }
#undef MF_PAIR
#undef MF_LOAD_PAIR
    if (stamp) t_loop1 = __builtin_amdgcn_s_memtime();
    __syncthreads();

    mfma_emit<MF_THREADS>(a, s_acc, acc_len, cnt, i0, m0, tid);
    if (stamp && lane == 0) {
        // stamps go to a buffer nothing else reads: {prologue, main loop incl. scatter, scatter only, tail, tiles}
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        unsigned long long *o = a.stamps + (static_cast<size_t>(blockIdx.x) * MF_WAVES + wave) * 8;
        o[0] = t_loop0 - t_begin;
        o[1] = t_loop1 - t_loop0;
        o[2] = t_scatter;
        o[3] = t_end - t_loop1;
        o[4] = static_cast<unsigned long long>((tiles - wave + MF_WAVES - 1) / MF_WAVES);
    }
}

}  // namespace iqa

using namespace iqa;

extern "C" int64_t iqa_mfma_ring_bytes(int32_t decimation)
{
    // LDS bytes of a ring-kernel block with contiguous slots for this decimation, or 0 when that form does not apply
    if (decimation < 1 || !mfma_ring_supported(decimation)) return 0;
    return static_cast<int64_t>(mfma_ring_lds_bytes((2 * decimation + 31) / 32, false, false));
}

extern "C" int64_t iqa_mfma_ring_lds_bytes(int32_t fmt, int32_t decimation, int32_t k_first, int32_t k_count, int32_t acc32)
{
    // LDS bytes of a workgroup of the ring kernel that iqa_mfma_ring_mode selects for this pass (0: none applies)
    if (decimation < 1 || (fmt != IQA_FMT_S16 && fmt != IQA_FMT_U8)) return 0;
    const int ks_all = (2 * decimation + 31) / 32;
    const int ks = k_count > 0 ? k_count : ks_all - k_first;
    const int mode = mfma_ring_mode(decimation, k_first, ks, acc32 == 0, fmt == IQA_FMT_U8);
    return mode ? static_cast<int64_t>(mfma_ring_lds_bytes(ks, mode == 2, fmt == IQA_FMT_U8)) : 0;
}

extern "C" int32_t iqa_mfma_ring_mode(int32_t fmt, int32_t decimation, int32_t k_first, int32_t k_count, int32_t acc32)
{
    if (decimation < 1 || (fmt != IQA_FMT_S16 && fmt != IQA_FMT_U8)) return 0;
    return mfma_ring_mode(decimation, k_first, k_count, acc32 == 0, fmt == IQA_FMT_U8);
}

extern "C" int64_t iqa_mfma_afrag_bytes(int32_t decimation)
{
    // bytes of tap fragments for ALL k steps of one q-group (a pass may use a sub-range of k steps)
    if (decimation < 1) return 0;
    const int64_t ksteps = (2 * static_cast<int64_t>(decimation) + 31) / 32;
    return ksteps * MF_KSTEP_BYTES;
}

extern "C" int iqa_channelize_mfma(const iqa_chan_params *p, const iqa_mfma_params *q, const void *afrag_dev,
                                   const void *raw_dev, int64_t n_frames, int64_t consumed, int64_t m_first,
                                   int64_t n_out, void *z_out_dev, void *stream)
{
    if (p == nullptr || q == nullptr) return fail_inval("params is NULL");
    const bool u8 = p->fmt == IQA_FMT_U8;  // uint8 captures: row-staged ring kernel only (reserved = 64|128)
    if (p->fmt != IQA_FMT_S16 && !(u8 && (q->reserved & 64)))
        return fail_inval("the MFMA channelizer takes int16 captures (and uint8 ones with the ring variant) only");
    if (p->ntaps <= 0 || p->decimation < 1) return fail_inval("bad ntaps/decimation");
    const int64_t D = p->decimation;
    // consumed may be negative: raw_dev then starts |consumed| frames BEFORE global frame 0 (a lead-in of zeros --
    // the filter's initial state -- in front of the capture, so that the first outputs are interior outputs too)
    if (n_out < 0 || n_frames < 0 || m_first < 0) return fail_inval("negative size");
    if (n_out == 0) return IQA_OK;
    if (!afrag_dev || !raw_dev || !z_out_dev) return fail_inval("NULL device pointer");
    const int ksteps_all = static_cast<int>((2 * D + 31) / 32);
    const int k_first = q->k_first;
    const int ksteps = q->k_count > 0 ? q->k_count : ksteps_all - k_first;
    if (k_first < 0 || ksteps <= 0 || k_first + ksteps > ksteps_all) return fail_inval("bad k-step range");
    if (q->q_group < 0) return fail_inval("bad q group");
    if (!q->finalize && !q->partial_out_dev) return fail_inval("non-final pass needs partial_out_dev");
    int range = q->outputs_per_block;
    if (range <= 0 || (range & 31)) return fail_inval("outputs_per_block must be a positive multiple of 32");
    // every frame the kernel touches must lie inside [0, n_frames): columns b in [m_first-64, last], each read
    // from frame b*D+1 for 16*ksteps frames (the k padding reads past the row into the next one)
    const int64_t blocks = (n_out + range - 1) / range;
    const int64_t last_cnt = n_out - (blocks - 1) * range;
    const int64_t last_tiles = (last_cnt + 63 + 31) / 32;
    const int64_t col_shift = static_cast<int64_t>(MF_Q) * q->q_group;
    const int64_t b_min = m_first - MF_Q - col_shift;
    const int64_t b_max = m_first + (blocks - 1) * range - MF_Q - col_shift + last_tiles * 32 - 1;
    const int64_t f_min = b_min * D + 1 - consumed;
    const int64_t f_max = b_max * D + 1 - consumed + 16LL * (k_first + ksteps) - 1;
    // full blocks are also bounded by their own tile count
    const int64_t full_tiles = (static_cast<int64_t>(range) + 63 + 31) / 32;
    const int64_t b_max_full = blocks > 1 ? m_first + (blocks - 2) * range - MF_Q - col_shift + full_tiles * 32 - 1 : b_max;
    const int64_t f_max_full = b_max_full * D + 1 - consumed + 16LL * (k_first + ksteps) - 1;
    if (f_min < 0 || f_max >= n_frames || f_max_full >= n_frames)
        return fail_inval("MFMA channelizer range reads outside the block (use iqa_channelize for the edges)");
    const int64_t acc_len = full_tiles * 32 + MF_Q + 4;
    const bool ring = (q->reserved & 64) != 0;  // block-wide contiguous LDS-DMA ring (channelize_ring.hip)
    size_t lds = static_cast<size_t>(ksteps) * MF_KSTEP_BYTES + 4 * acc_len * sizeof(int);
    int ring_mode = 0;
    if (ring) {
        ring_mode = mfma_ring_mode(static_cast<int>(D), k_first, ksteps, !(q->reserved & 128), u8);
        if (ring_mode == 0)
            return fail_inval("the ring kernel does not cover this (decimation, k-step range, sum width): see iqa_mfma_ring_mode");
        if (ring_mode == 1) {
            // a contiguous slot is filled in whole 1 KiB chunks: every tile reads 2048*ksteps bytes from its first frame
            const int64_t slot_frames = 512LL * ksteps;
            const int64_t t_last = m_first + (blocks - 1) * range - MF_Q - col_shift + (last_tiles - 1) * 32;
            const int64_t t_full = blocks > 1 ? m_first + (blocks - 2) * range - MF_Q - col_shift + (full_tiles - 1) * 32 : t_last;
            if (t_last * D + 1 - consumed + slot_frames > n_frames || t_full * D + 1 - consumed + slot_frames > n_frames)
                return fail_inval("ring kernel range reads outside the block (use iqa_channelize for the edges)");
        }  // row-staged slots read exactly what the per-lane kernel reads: covered by the checks above
        lds = mfma_ring_lds_bytes(ksteps, ring_mode == 2, u8);  // ring + sliding window: independent of outputs_per_block
    }
    if (lds > 160 * 1024) return fail_inval("tap fragments + accumulators exceed 160 KiB of LDS");

    MfmaArgs a;
    a.afrag = static_cast<const v4i_t *>(afrag_dev);
    a.raw = static_cast<const int *>(raw_dev);
    a.out = static_cast<float2 *>(z_out_dev);
    a.consumed = consumed;
    a.m_lo = m_first;
    a.n_out = n_out;
    a.D = static_cast<int>(D);
    a.ksteps = ksteps;
    a.range = range;
    a.k_first = k_first;
    a.col_shift = static_cast<int>(col_shift);
    a.finalize = q->finalize;
    a.partial_in = static_cast<const double2 *>(q->partial_in_dev);
    a.partial_out = static_cast<double2 *>(q->partial_out_dev);
    a.debug = q->reserved;
    a.stamps = static_cast<unsigned long long *>(q->debug_stamps);
    a.unit = q->unit;
    a.c_re = q->c_re;
    a.c_im = q->c_im;
    a.conj_sum = p->conj_sum;
    a.rotate = p->rotate;
    a.rot_step = p->rot_step;
    a.rot_base = p->rot_base;
    a.sc_re = p->out_scale_re;
    a.sc_im = p->out_scale_im;
    {
        // rotation between outputs 64 apart, for the ring kernel's recurrence: frac(64*rot_step / 2^64) turns
        const unsigned long long st = p->rot_step * 64ULL;
        const double turns = static_cast<double>(st >> 11) * (1.0 / 9007199254740992.0);
        a.rot64_re = std::cos(2.0 * M_PI * turns);
        a.rot64_im = std::sin(2.0 * M_PI * turns);
    }
    a.raw_partials = 0;
    a.high_taps_only = 0;  // (single-lane launches always compute the q2*hi product: reserved bit 8 is accepted and ignored)
    if (ring) {
        return mfma_ring_launch(a, static_cast<unsigned>(blocks), lds, as_stream(stream), ring_mode == 2, u8);
    }
    {
        // the 160 KiB dynamic-LDS limit is a per-device attribute (one bit per device id; racing threads both set it)
        static std::atomic<unsigned long long> done{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done.load(std::memory_order_acquire) & bit)) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_channelize_mfma_s16),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) {
                set_error("k_channelize_mfma_s16: cannot raise the dynamic LDS limit on device %d: %s", dev, hipGetErrorString(e));
                return IQA_EHIP;
            }
            done.fetch_or(bit, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL(k_channelize_mfma_s16, dim3(static_cast<unsigned>(blocks)), dim3(MF_THREADS), lds,
                       as_stream(stream), a);
    return check_launch("k_channelize_mfma_s16");
}

// ---- several lanes (channels x tap-row groups) of one capture in ONE launch of the ring kernel ------------------------

static int channelize_mfma_lanes(int32_t fmt, int32_t decimation, int32_t k_first, int32_t k_count, int32_t outputs_per_block,
                                 const iqa_mfma_lane *lanes, int32_t n_lanes, const void *raw_dev, int64_t n_frames, int64_t consumed,
                                 int64_t m_first, int64_t n_out, void *stream, bool pairs)
{
    if (lanes == nullptr || n_lanes < 1) return fail_inval("no lanes");
    const bool u8 = fmt == IQA_FMT_U8;
    if (fmt != IQA_FMT_S16 && !u8) return fail_inval("the MFMA channelizer takes int16 and uint8 captures only");
    if (decimation < 1) return fail_inval("bad decimation");
    const int64_t D = decimation;
    if (n_out < 0 || n_frames < 0 || m_first < 0) return fail_inval("negative size");
    if (n_out == 0) return IQA_OK;
    if (!raw_dev) return fail_inval("NULL device pointer");
    const int ksteps_all = static_cast<int>((2 * D + 31) / 32);
    const int ksteps = k_count > 0 ? k_count : ksteps_all - k_first;
    if (k_first < 0 || ksteps <= 0 || k_first + ksteps > ksteps_all) return fail_inval("bad k-step range");
    const int range = outputs_per_block;
    if (range <= 0 || (range & 31)) return fail_inval("outputs_per_block must be a positive multiple of 32");
    // 64-bit sums (iqa_mfma_lane.reserved bit 0, the same for every lane of a launch): fragments without the int32 bound
    const bool acc64 = (lanes[0].reserved & 1) != 0;
    for (int i = 1; i < n_lanes; ++i)
        if (lanes[i].afrag_dev && ((lanes[i].reserved & 1) != 0) != acc64) return fail_inval("the lanes of a launch share the sum width (reserved bit 0)");
    const int ring_mode = mfma_ring_mode(static_cast<int>(D), k_first, ksteps, acc64, u8);
    if (ring_mode == 0) return fail_inval("the ring kernel does not cover this (decimation, k-step range, sum width): see iqa_mfma_ring_mode");
    if (acc64 && ring_mode != 1) return fail_inval("64-bit sums: contiguous slots only");
    if (pairs && !mfma_ring_pairs_supported(static_cast<int>(D), k_first, ksteps, u8, acc64))
        return fail_inval("lane pairs are not available for this (format, decimation, k-step range, sum width): see iqa_mfma_ring_lanes");
    // every frame any lane touches must lie inside [0, n_frames): the lane with the largest tap-row group reads the
    // earliest data rows, group 0 the latest (see iqa_channelize_mfma for the geometry)
    int q_max = 0;
    for (int i = 0; i < n_lanes; ++i) {
        if (lanes[i].q_group < 0) return fail_inval("bad q group");
        if (!lanes[i].afrag_dev) {
            if (pairs && (i & 1)) continue;  // a pair without a second lane: that half of the workgroup idles
            return fail_inval("lane without tap fragments");
        }
        if (lanes[i].finalize ? !lanes[i].z_out_dev : !lanes[i].partial_out_dev) return fail_inval("lane without an output buffer");
        q_max = lanes[i].q_group > q_max ? lanes[i].q_group : q_max;
    }
    const int64_t blocks = (n_out + range - 1) / range;
    const int64_t last_cnt = n_out - (blocks - 1) * range;
    const int64_t last_tiles = (last_cnt + 63 + 31) / 32, full_tiles = (static_cast<int64_t>(range) + 63 + 31) / 32;
    const int64_t f_min = (m_first - MF_Q - static_cast<int64_t>(MF_Q) * q_max) * D + 1 - consumed;
    const int64_t tail = ring_mode == 1 ? 512LL * ksteps : 16LL * (k_first + ksteps);  // frames read from a row's first frame
    const int64_t t_last = m_first + (blocks - 1) * range - MF_Q + (last_tiles - 1) * 32;  // first row of the last tile, group 0
    const int64_t t_full = blocks > 1 ? m_first + (blocks - 2) * range - MF_Q + (full_tiles - 1) * 32 : t_last;
    const int64_t row_span = ring_mode == 1 ? 0 : 31;  // row-staged slots fetch each of the tile's 32 rows separately
    if (f_min < 0 || (t_last + row_span) * D + 1 - consumed + tail > n_frames || (t_full + row_span) * D + 1 - consumed + tail > n_frames)
        return fail_inval("multi-lane ring launch reads outside the block (use iqa_channelize for the edges)");
    const size_t lds = mfma_ring_lds_bytes(ksteps, ring_mode == 2, u8);
    if (lds == 0 || lds > 160 * 1024) return fail_inval("ring + window exceed 160 KiB of LDS");

    MfmaArgs a{};
    a.raw = static_cast<const int *>(raw_dev);
    a.consumed = consumed;
    a.m_lo = m_first;
    a.n_out = n_out;
    a.D = static_cast<int>(D);
    a.ksteps = ksteps;
    a.range = range;
    a.k_first = k_first;
    a.debug = 64 | (acc64 ? 0 : 128);
    MfmaLane packed[16];
    if (n_lanes > 16) return fail_inval("at most 16 lanes per launch");
    for (int i = 0; i < n_lanes; ++i) {
        const iqa_mfma_lane &s = lanes[i];
        MfmaLane &l = packed[i];
        l.afrag = static_cast<const v4i_t *>(s.afrag_dev);
        l.out = static_cast<float2 *>(s.z_out_dev);
        l.partial_in = static_cast<const double2 *>(s.partial_in_dev);
        l.partial_out = static_cast<double2 *>(s.partial_out_dev);
        l.unit = s.unit;
        l.c_re = s.c_re;
        l.c_im = s.c_im;
        l.rot_step = s.rot_step;
        l.rot_base = s.rot_base;
        const unsigned long long st = s.rot_step * 64ULL;
        const double turns = static_cast<double>(st >> 11) * (1.0 / 9007199254740992.0);
        l.rot64_re = std::cos(2.0 * M_PI * turns);
        l.rot64_im = std::sin(2.0 * M_PI * turns);
        l.sc_re = s.out_scale_re;
        l.sc_im = s.out_scale_im;
        l.col_shift = MF_Q * s.q_group;
        l.finalize = s.finalize;
        l.conj_sum = s.conj_sum;
        l.rotate = s.rotate;
        l.raw_partials = (s.raw_partials != 0 && !s.finalize && !s.partial_in_dev) ? 1 : 0;
        if (s.raw_partials && !l.raw_partials) return fail_inval("raw partials need finalize == 0 and no partial_in");
        if (s.raw_partials && acc64) return fail_inval("raw partials are int32 sums: not with 64-bit sums");
        l.high_taps_only = (s.reserved & 2) != 0 ? 1 : 0;
    }
    return mfma_ring_launch_multi(a, packed, n_lanes, lds, as_stream(stream), ring_mode == 2, u8, nullptr, pairs, acc64);
}

extern "C" int iqa_channelize_mfma_multi(int32_t fmt, int32_t decimation, int32_t k_first, int32_t k_count,
                                         int32_t outputs_per_block, const iqa_mfma_lane *lanes, int32_t n_lanes,
                                         const void *raw_dev, int64_t n_frames, int64_t consumed, int64_t m_first,
                                         int64_t n_out, void *stream)
{
    return channelize_mfma_lanes(fmt, decimation, k_first, k_count, outputs_per_block, lanes, n_lanes, raw_dev, n_frames, consumed,
                                 m_first, n_out, stream, false);
}

extern "C" int iqa_channelize_mfma_pairs(int32_t fmt, int32_t decimation, int32_t k_first, int32_t k_count,
                                         int32_t outputs_per_block, const iqa_mfma_lane *lanes, int32_t n_lanes,
                                         const void *raw_dev, int64_t n_frames, int64_t consumed, int64_t m_first,
                                         int64_t n_out, void *stream)
{
    return channelize_mfma_lanes(fmt, decimation, k_first, k_count, outputs_per_block, lanes, n_lanes, raw_dev, n_frames, consumed,
                                 m_first, n_out, stream, true);
}

extern "C" int32_t iqa_mfma_ring_lanes(int32_t fmt, int32_t decimation, int32_t k_first, int32_t k_count, int32_t acc32)
{
    if (decimation < 1 || (fmt != IQA_FMT_S16 && fmt != IQA_FMT_U8)) return 0;
    const bool u8 = fmt == IQA_FMT_U8, acc64 = acc32 == 0;
    const int ks_all = (2 * decimation + 31) / 32;
    const int ks = k_count > 0 ? k_count : ks_all - k_first;
    const int mode = mfma_ring_mode(decimation, k_first, ks, acc64, u8);
    if (mode == 0 || (acc64 && mode != 1)) return 0;
    return 1 | (mfma_ring_pairs_supported(decimation, k_first, ks, u8, acc64) ? 2 : 0);
}

extern "C" int32_t iqa_mfma_ring_pairs(int32_t fmt, int32_t decimation, int32_t k_first, int32_t k_count)
{
    if (decimation < 1 || fmt != IQA_FMT_S16) return 0;
    const int ks_all = (2 * decimation + 31) / 32;
    return mfma_ring_pairs_supported(decimation, k_first, k_count > 0 ? k_count : ks_all - k_first, false) ? 1 : 0;
}

// Sum of the partial sums of a filter's tap-row groups (each written by its own lane of a multi-lane launch), then the
// same conversion, rotation and scaling as the kernels' own emission: z[m_first + i].
namespace iqa {
struct CombineArgs {
    const double2 *part[16];
    double unit[16], c_re[16], c_im[16];  // raw != 0: part[k] holds int2 sums, scaled here exactly as the kernels' emission does
    int raw;
    int n_parts;
    float2 *out;
    long long m_first, n_out;
    int conj_sum, rotate;
    unsigned long long rot_step, rot_base;
    float sc_re, sc_im;
};

__global__ __launch_bounds__(256) void k_mfma_combine(CombineArgs a)
{
    const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= a.n_out) return;
    auto part = [&](int k) -> double2 {
        if (!a.raw) return a.part[k][i];
        const int2 v = reinterpret_cast<const int2 *>(a.part[k])[i];
        return make_double2(mfma_scaled_sum(static_cast<double>(v.x), a.c_re[k], a.unit[k]),
                            mfma_scaled_sum(static_cast<double>(v.y), a.c_im[k], a.unit[k]));
    };
    double2 d = part(0);
    for (int k = 1; k < a.n_parts; ++k) {  // in group order: the chained single-lane passes add them in this order too
        const double2 v = part(k);
        d.x = __dadd_rn(v.x, d.x);
        d.y = __dadd_rn(v.y, d.y);
    }
    double cs = 1.0, sn = 0.0;
    if (a.rotate) {
        const unsigned long long ph = a.rot_base + static_cast<unsigned long long>(a.m_first + i) * a.rot_step;
        sincospi(2.0 * (static_cast<double>(ph >> 11) * (1.0 / 9007199254740992.0)), &sn, &cs);
    }
    a.out[i] = mfma_finish(d.x, d.y, a.conj_sum, a.rotate, cs, sn, a.sc_re, a.sc_im);
}
}  // namespace iqa

extern "C" int iqa_mfma_combine(const iqa_chan_params *p, const void *const *partials_dev, int32_t n_partials,
                                const double *raw_scale, int64_t m_first, int64_t n_out, void *z_out_dev, void *stream)
{
    if (p == nullptr || partials_dev == nullptr) return fail_inval("params is NULL");
    if (n_partials < 1 || n_partials > 16) return fail_inval("1..16 partial buffers");
    if (n_out < 0 || m_first < 0) return fail_inval("negative size");
    if (n_out == 0) return IQA_OK;
    if (!z_out_dev) return fail_inval("NULL device pointer");
    CombineArgs a{};
    for (int k = 0; k < n_partials; ++k) {
        if (!partials_dev[k]) return fail_inval("NULL partial buffer");
        a.part[k] = static_cast<const double2 *>(partials_dev[k]);
    }
    a.n_parts = n_partials;
    a.raw = raw_scale != nullptr;
    for (int k = 0; k < n_partials && raw_scale; ++k) {
        a.unit[k] = raw_scale[3 * k];
        a.c_re[k] = raw_scale[3 * k + 1];
        a.c_im[k] = raw_scale[3 * k + 2];
    }
    a.out = static_cast<float2 *>(z_out_dev);
    a.m_first = m_first;
    a.n_out = n_out;
    a.conj_sum = p->conj_sum;
    a.rotate = p->rotate;
    a.rot_step = p->rot_step;
    a.rot_base = p->rot_base;
    a.sc_re = p->out_scale_re;
    a.sc_im = p->out_scale_im;
    hipLaunchKernelGGL(k_mfma_combine, dim3(static_cast<unsigned>((n_out + 255) / 256)), dim3(256), 0, as_stream(stream), a);
    return check_launch("k_mfma_combine");
}

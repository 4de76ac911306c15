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

// ---------------------------------------------------------------------------------------------------
// LDS-staged variant.  Same arithmetic, different data path: instead of every lane loading 32 bytes of
// its own row into VGPRs (64 separate L1 look-ups per wave-instruction, ~55 % TA busy), a wave copies
// its tile's k-chunk -- 32 rows x 64 bytes -- into a PRIVATE 2 KiB LDS slot with two
// global_load_lds_dwordx4 (LDS-DMA, no VGPRs): four consecutive lanes fetch the four 16-byte chunks of
// one row's 64-byte segment, so a wave-instruction touches 16 segments instead of 64 rows.  The image
// is XOR-swizzled on the SOURCE side (LDS-DMA writes base + lane*16 linearly): slot s of row r holds
// chunk s ^ ((r>>2)&3), which makes the two ds_read_b128 of the MFMA lane (col, h) conflict-free.
// A ring of ST_NB slots per wave keeps ST_NB-1 k steps of DMA in flight behind a hand-counted vmcnt
// (the slots are private to the wave, so no barrier is involved).
constexpr int ST_SLOT_BYTES = 2048;
typedef __attribute__((address_space(3))) void lds_void_t;

// WAVES: waves per block (one block per CU); ST_NB: ring depth; JIT_FRAGS: read each row tile's tap
// fragments right before its MFMAs (8 live registers instead of 32) so that three waves fit per SIMD.
template <int WAVES, int ST_NB, bool JIT_FRAGS>
__global__ __launch_bounds__(WAVES *kWave, WAVES / 4) void k_channelize_mfma_s16_staged(MfmaArgs a)
{
    constexpr int THREADS = WAVES * kWave;
    constexpr int ST_BYTES = WAVES * ST_NB * ST_SLOT_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;

    const long long i0 = static_cast<long long>(blockIdx.x) * a.range;
    const int cnt = static_cast<int>(min(static_cast<long long>(a.range), a.n_out - i0));
    const long long m0 = a.m_lo + i0;
    const int tiles = (cnt + 63 + 31) >> 5;
    const int acc_len = tiles * 32 + MF_Q + 4;

    v4i_t *s_a = reinterpret_cast<v4i_t *>(smem);
    char *s_stage = smem + static_cast<size_t>(a.ksteps) * MF_KSTEP_BYTES + wave * (ST_NB * ST_SLOT_BYTES);
    int *s_acc = reinterpret_cast<int *>(smem + static_cast<size_t>(a.ksteps) * MF_KSTEP_BYTES + ST_BYTES);
    {
        const int n16 = a.ksteps * (MF_KSTEP_BYTES / 16);
        for (int i0f = tid; i0f < n16; i0f += THREADS * 8) {
            v4i_t tmp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0f + u * THREADS;
                tmp[u] = (i < n16) ? a.afrag[i] : v4i_t{0, 0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0f + u * THREADS;
                if (i < n16) s_a[i] = tmp[u];
            }
        }
    }
    for (int i = tid; i < 4 * acc_len; i += THREADS) s_acc[i] = 0;
    __syncthreads();

    // producer role of this lane: row (lane>>2) [+16 for the second instruction], chunk (lane&3)^f(row)
    const int drow = lane >> 2;
    const int dchunk = (lane & 3) ^ ((drow >> 2) & 3);  // f(row) == f(row+16)
    const long long row_bytes = static_cast<long long>(a.D) * 4;
    const long long doff0 = drow * row_bytes + dchunk * 16;
    const long long doff1 = doff0 + 16 * row_bytes;
    // consumer role: MFMA lane (col, h) wants chunks 2h and 2h+1 of row `col`
    const int f = (col >> 2) & 3;
    const int rd0 = col * 64 + ((2 * h) ^ f) * 16;
    const int rd1 = col * 64 + ((2 * h + 1) ^ f) * 16;

    auto tile_base = [&](int tile) -> const char * {
        const long long b0 = m0 - MF_Q - a.col_shift + static_cast<long long>(tile) * 32;  // first data row of the tile
        return reinterpret_cast<const char *>(a.raw + (b0 * a.D + 1 - a.consumed) + 16 * a.k_first);
    };
    int pf_tile = wave, pf_ks = 0, pf_slot = 0;
    const char *pf_base = tile_base(min(pf_tile, tiles - 1));
    auto issue = [&]() {
        const char *g = pf_base + pf_ks * 64;
        char *dst = s_stage + pf_slot * ST_SLOT_BYTES;
        __builtin_amdgcn_global_load_lds(g + doff0, (lds_void_t *)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(g + doff1, (lds_void_t *)(dst + 1024), 16, 0, 0);
        if (++pf_ks == a.ksteps) {
            pf_ks = 0;
            pf_tile += WAVES;
            pf_base = tile_base(min(pf_tile, tiles - 1));  // past the end: re-read the last tile (in bounds, unused)
        }
        pf_slot = (pf_slot + 1 == ST_NB) ? 0 : pf_slot + 1;
    };
#pragma unroll
    for (int s = 0; s < ST_NB; ++s) issue();

    v4i_t fr[2 * MF_ROWTILES];
    if constexpr (!JIT_FRAGS) {
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
    int cur_slot = 0;
    const bool stamp = (a.debug & 2) && a.stamps != nullptr;  // diagnostics build path only
    unsigned long long c_vm = 0, c_lds = 0, c_dma = 0, c_mfma = 0, c_sc = 0, tp = 0;
#define ST_STAMP(ACC)                                                           \
    if (stamp) {                                                                \
        __builtin_amdgcn_sched_barrier(0);                                      \
        const unsigned long long tn = __builtin_amdgcn_s_memtime();             \
        ACC += tn - tp;                                                         \
        tp = tn;                                                                \
        __builtin_amdgcn_sched_barrier(0);                                      \
    }
    if (stamp) tp = __builtin_amdgcn_s_memtime();
    for (int t = wave; t < tiles; t += WAVES) {
        for (int ks = 0; ks < a.ksteps; ++ks) {
            // the DMA pair of this step has landed once at most 2*(ST_NB-1) younger ones are outstanding
            if constexpr (ST_NB == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (ST_NB == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            static_assert(ST_NB >= 2 && ST_NB <= 4, "vmcnt literals above are 2*(ST_NB-1)");
            ST_STAMP(c_vm);
            const char *sl = s_stage + cur_slot * ST_SLOT_BYTES;
            const v4i_t d0 = *reinterpret_cast<const v4i_t *>(sl + rd0);
            const v4i_t d1 = *reinterpret_cast<const v4i_t *>(sl + rd1);
            v4i_t hi, lo;
            hi.x = __builtin_amdgcn_perm(d0.y, d0.x, 0x07050301);
            hi.y = __builtin_amdgcn_perm(d0.w, d0.z, 0x07050301);
            hi.z = __builtin_amdgcn_perm(d1.y, d1.x, 0x07050301);
            hi.w = __builtin_amdgcn_perm(d1.w, d1.z, 0x07050301);
            lo.x = __builtin_amdgcn_perm(d0.y, d0.x, 0x06040200) ^ 0x80808080;
            lo.y = __builtin_amdgcn_perm(d0.w, d0.z, 0x06040200) ^ 0x80808080;
            lo.z = __builtin_amdgcn_perm(d1.y, d1.x, 0x06040200) ^ 0x80808080;
            lo.w = __builtin_amdgcn_perm(d1.w, d1.z, 0x06040200) ^ 0x80808080;
            // the slot has been read into registers (the byte split above waited for it): refill it
            asm volatile("" ::"v"(hi), "v"(lo) : "memory");
            ST_STAMP(c_lds);
            if (!(a.debug & 16)) issue();  // bit 4: timing experiment without the data stream
            ST_STAMP(c_dma);
            cur_slot = (cur_slot + 1 == ST_NB) ? 0 : cur_slot + 1;
            const int kn = (ks + 1 < a.ksteps) ? ks + 1 : 0;
            const v4i_t *fa = s_a + kn * (MF_KSTEP_BYTES / 16) + lane;
            if (a.debug & 32) {  // bit 5: timing experiment without the matrix work
                asm volatile("" ::"v"(hi), "v"(lo));
                ST_STAMP(c_mfma);
                continue;
            }
            if constexpr (JIT_FRAGS) {
                const v4i_t *fc = s_a + ks * (MF_KSTEP_BYTES / 16) + lane;
                if (ks == 0) {
#pragma unroll
                    for (int rt = 0; rt < MF_ROWTILES; ++rt) {
                        const v4i_t q1 = fc[(2 * rt) * 64], q2 = fc[(2 * rt + 1) * 64];
                        acc1[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(q1, hi, zero16, 0, 0, 0);
                        acc2[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(q1, lo, zero16, 0, 0, 0);
                        acc2[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(q2, hi, acc2[rt], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int rt = 0; rt < MF_ROWTILES; ++rt) {
                        const v4i_t q1 = fc[(2 * rt) * 64], q2 = fc[(2 * rt + 1) * 64];
                        acc1[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(q1, hi, acc1[rt], 0, 0, 0);
                        acc2[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(q1, lo, acc2[rt], 0, 0, 0);
                        acc2[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(q2, hi, acc2[rt], 0, 0, 0);
                    }
                }
            } else if (ks == 0) {
#pragma unroll
                for (int rt = 0; rt < MF_ROWTILES; ++rt) {
                    acc1[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr[2 * rt], hi, zero16, 0, 0, 0);
                    acc2[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr[2 * rt], lo, zero16, 0, 0, 0);
                    acc2[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr[2 * rt + 1], hi, acc2[rt], 0, 0, 0);
                    fr[2 * rt] = fa[(2 * rt) * 64];
                    fr[2 * rt + 1] = fa[(2 * rt + 1) * 64];
                }
            } else {
#pragma unroll
                for (int rt = 0; rt < MF_ROWTILES; ++rt) {
                    acc1[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr[2 * rt], hi, acc1[rt], 0, 0, 0);
                    acc2[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr[2 * rt], lo, acc2[rt], 0, 0, 0);
                    acc2[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr[2 * rt + 1], hi, acc2[rt], 0, 0, 0);
                    fr[2 * rt] = fa[(2 * rt) * 64];
                    fr[2 * rt + 1] = fa[(2 * rt + 1) * 64];
                }
            }
            ST_STAMP(c_mfma);
        }
        if (a.debug & 1) {  // bit 0: timing experiment without the LDS scatter
#pragma unroll
            for (int rt = 0; rt < MF_ROWTILES; ++rt) asm volatile("" ::"v"(acc1[rt]), "v"(acc2[rt]));
            continue;
        }
        int *base = s_acc + (t * 32 + col + 4 * h + 1);
#pragma unroll
        for (int rt = 0; rt < MF_ROWTILES; ++rt) {
            int *p1 = base + (rt >> 1) * acc_len + (rt & 1) * 32;
            int *p2 = p1 + 2 * acc_len;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int off = (r & 3) + 8 * (r >> 2);
                atomicAdd(p1 + off, acc1[rt][r]);
                atomicAdd(p2 + off, acc2[rt][r]);
            }
        }
        ST_STAMP(c_sc);
    }
#undef ST_STAMP
    if (stamp && lane == 0) {
        unsigned long long *o = a.stamps + (static_cast<size_t>(blockIdx.x) * WAVES + wave) * 8;
        o[0] = c_vm;
        o[1] = c_lds;
        o[2] = c_dma;
        o[3] = c_mfma;
        o[4] = static_cast<unsigned long long>((tiles - wave + WAVES - 1) / WAVES);
        o[5] = c_sc;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the trailing (unused) DMAs before the LDS is reused
    __syncthreads();
    mfma_emit<THREADS>(a, s_acc, acc_len, cnt, i0, m0, tid);
}

}  // namespace iqa

using namespace iqa;

extern "C" int64_t iqa_mfma_ring_bytes(int32_t decimation)
{
    // LDS bytes of the ring kernel's data ring for this decimation, or 0 when the ring kernel does not apply
    if (decimation < 1 || !mfma_ring_supported(decimation)) return 0;
    return static_cast<int64_t>(mfma_ring_lds_bytes((2 * decimation + 31) / 32));
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
    if (p->fmt != IQA_FMT_S16) return fail_inval("the MFMA channelizer takes int16 captures only");
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
    const bool staged = !ring && (q->reserved & 4) != 0;
    const bool waves12 = staged && (q->reserved & 8) != 0;  // 12-wave blocks (three waves per SIMD), ring of 3
    const size_t st_bytes = staged ? (waves12 ? 12 * 3 : 8 * 4) * static_cast<size_t>(ST_SLOT_BYTES) : 0;
    size_t lds = static_cast<size_t>(ksteps) * MF_KSTEP_BYTES + 4 * acc_len * sizeof(int) + st_bytes;
    if (ring) {
        if (!mfma_ring_supported(static_cast<int>(D)) || k_first != 0 || ksteps != ksteps_all)
            return fail_inval("the ring kernel needs D % 4 == 0, D <= 256 and all k steps in one pass");
        // a slot is filled in whole 1 KiB chunks: every tile reads 2048*ksteps bytes from its first frame
        const int64_t slot_frames = 512LL * ksteps;
        const int64_t t_last = m_first + (blocks - 1) * range - MF_Q - col_shift + (last_tiles - 1) * 32;
        const int64_t t_full = blocks > 1 ? m_first + (blocks - 2) * range - MF_Q - col_shift + (full_tiles - 1) * 32 : t_last;
        if (t_last * D + 1 - consumed + slot_frames > n_frames || t_full * D + 1 - consumed + slot_frames > n_frames)
            return fail_inval("ring kernel range reads outside the block (use iqa_channelize for the edges)");
        lds = mfma_ring_lds_bytes(ksteps);  // ring + sliding window: independent of outputs_per_block
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
    if (ring) {
        mfma_ring_launch(a, static_cast<unsigned>(blocks), lds, as_stream(stream));
        return check_launch("k_channelize_mfma_s16_ring");
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_channelize_mfma_s16),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_channelize_mfma_s16_staged<8, 4, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_channelize_mfma_s16_staged<12, 3, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (staged) {
        // the second DMA of a step reads 16 rows further than the per-lane loads of the plain kernel: same rows, checked above
        if (waves12)
            hipLaunchKernelGGL((k_channelize_mfma_s16_staged<12, 3, true>), dim3(static_cast<unsigned>(blocks)),
                               dim3(12 * kWave), lds, as_stream(stream), a);
        else
            hipLaunchKernelGGL((k_channelize_mfma_s16_staged<8, 4, false>), dim3(static_cast<unsigned>(blocks)),
                               dim3(MF_THREADS), lds, as_stream(stream), a);
        return check_launch("k_channelize_mfma_s16_staged");
    }
    hipLaunchKernelGGL(k_channelize_mfma_s16, dim3(static_cast<unsigned>(blocks)), dim3(MF_THREADS), lds,
                       as_stream(stream), a);
    return check_launch("k_channelize_mfma_s16");
}

// channelize_ring.hip -- int8-MFMA channelizer with a block-wide contiguous LDS-DMA ring (gfx950).
//
// Same mathematics, tap fragments, accumulators and emission as channelize_mfma.hip (reference
// processing.py:268-279, 289-297, 325-346, 354-360); what changes is how the capture reaches the matrix
// cores.  The kernels of channelize_mfma.hip let every lane (or every group of four lanes) fetch its own
// row, so one wave-instruction touches 16-64 different 416-byte rows: measured, that address pattern alone
// caps the data path at ~3.8 TB/s.  Here the block streams the capture exactly as it lies in memory:
//
//   * the data columns a block needs are ONE contiguous run of the capture; it is cut into tiles of 32 rows
//     (32*D frames) and copied into LDS slots by global_load_lds_dwordx4 in chunks of 1 KiB per
//     wave-instruction (64 lanes x 16 consecutive bytes): 8-9 cache lines per instruction instead of 64;
//   * 8 waves, two per SIMD.  Wave (rt, cp): rt = one of the four tap-row tiles (32 of the 128 rows),
//     cp = column-tile parity.  A round = two column tiles (cp 0/1); the four waves of a parity read the
//     same slot (ds_read_b128 at row stride 4*D) and each multiplies it with ITS 32 tap rows, whose
//     fragments (8*KS registers) stay in registers for the whole block -- no tap traffic through LDS;
//   * ring of R rounds (as many as LDS holds, <= 5): the DMAs of round r+R-1 are issued during round r, one
//     per k step, by two waves of each parity (one issuing wave per SIMD), behind a counted s_waitcnt vmcnt;
//     one s_barrier per round publishes the landed slots and frees the oldest;
//   * the sums live in a 512-position sliding window (one int32 = 256*S1 + S2 per component): a round's
//     scatter reaches 127 positions ahead, the 64 outputs completed two rounds ago are converted, rotated
//     and stored by one wave while the others multiply, and their slots are reused 8 rounds later.  So a
//     block is not bounded by LDS: every CU gets ONE contiguous range of the launch (persistent blocks:
//     one prologue per CU, 63 rows of halo per CU).
//
// Round 3: up to 13 k steps (and in every row-staged int16 kernel) the LDS-DMA copies are replaced by LOADER WAVES that fetch
// the tiles into registers, split every int16 into its high byte and its biased low byte once per tile and write two byte
// planes per slot; the multiplying waves read their MFMA operands as they lie in LDS (IQA_RING_SPLIT_STAGE below says what
// that bought).  The ring of such a kernel is two slots per parity in LDS plus two or three rounds of loads in the loaders'
// registers; the loaders also emit.  The text above describes the LDS-DMA form, which remains for uint8 captures and for
// 14..16 k steps.  Up to 8 k steps a row whose last k step is at most half full ends in 32x32x16 MFMAs (the ..._half kernels).
//
// A contiguous slot is 1024*(2*KS + 1) bytes >= 32 rows at a pitch of D/4 (+1) 16-byte units (KS = ceil(2D/32) k steps;
// the K padding of the last k step reads on into the next row, against zero taps); a tile takes 2*KS + 1 DMA
// instructions; it needs D % 4 == 0 (16-byte aligned rows for ds_read_b128) and KS <= 16.
// Every other decimation (odd D, D > 256 in several k-step ranges) uses row-staged slots, see RingGeo.
#include "mfma_common.h"

// cache-policy bits of the LDS-DMA loads (gfx940+: 1 = sc0, 2 = nt, 16 = sc1); a build-time knob for experiments
#ifndef IQA_RING_DMA_AUX
#define IQA_RING_DMA_AUX 0
#endif
// 1: a scheduling barrier behind every k step of the multiplying loop (what the loop needed while it also issued one
// LDS-DMA per k step: the compiler will not move a ds_read across a DMA).  The refills now go out in front of the loop,
// and the compiler's own interleaving of fragment reads, byte splits and MFMAs is faster: probe/kstep_probe.hip measures
// 1483 ns per tile pair and SIMD against 1842 ns with the barriers (13 k steps, two waves per SIMD, no DMA, no scatter).
#ifndef IQA_RING_SCHED_BARRIER
#define IQA_RING_SCHED_BARRIER 0
#endif
// 1: in the kernels without loader waves (9..16 k steps: D = 132..256) the two column-tile parities run HALF A ROUND
// apart: two workgroup barriers per round, each the tile boundary of one parity and a mid-tile barrier of the other.
// A SIMD holds one wave of each parity; with one barrier per round both reach their tile boundary -- results out of the
// matrix pipe, 16 LDS adds, the barrier, the first fragment reads of the next tile -- at the same time and the matrix
// pipe idles for that long; half a round apart, one of them is always in the middle of its 3*KS MFMAs (one wave alone
// keeps the pipe 96 % busy: probe/kstep_probe.hip).  MEASURED and left off: 9.48 against 9.15 ms on the five-target launch,
// 1.057 against 1.055 ms on one channel at D = 208 -- tile boundaries are not what holds this kernel back (DESIGN.md
// section 6).  Build-time knob for A/B measurements only.
// Loader waves (extra waves that feed the ring and emit) up to this many k steps; 14..16 k steps need more registers than
// twelve waves leave each other (168), there the multiplying waves issue LDS-DMAs themselves.  With LDS-DMA staging
// (IQA_RING_SPLIT_STAGE=0) round 2 measured loaders at 13 k steps as 4 % faster on a five-target launch of single lanes but
// with L2 misses of 1.1-1.4x the capture instead of 1.003x (the lanes of a range lose their lock-step); banks at 9..16 k
// steps run as lane PAIRS (no loader waves, paced), so this concerns the single-lane kernels.
#ifndef IQA_RING_LOADERS_MAX_KS
#define IQA_RING_LOADERS_MAX_KS 13
#endif
// A/B knobs for the waves' interplay on a SIMD (diagnostic builds): IQA_RING_DEFER 0 = parity-1 waves scatter their tile
// right behind it like parity 0; IQA_RING_PRIO 1 = parity-1 waves run at raised priority (s_setprio 1), 2 = parity-0 waves.
#ifndef IQA_RING_DEFER
#define IQA_RING_DEFER 1
#endif
#ifndef IQA_RING_PRIO
#define IQA_RING_PRIO 0
#endif
#ifndef IQA_RING_STAGGER
#define IQA_RING_STAGGER 0
#endif
#ifndef IQA_RING_ROUNDS_MAX
#define IQA_RING_ROUNDS_MAX 5  // ring depth of the single-lane kernels in rounds of two tiles, where LDS allows it
#endif
// Lane pairs: data tiles per round.  1 (default): one tile per round, two issuing waves (RingGeo PAIR below).  2: a round is
// TWO tiles and every wave multiplies both with its lane's tap rows -- half as many round barriers per tile, the ring fed
// exactly as in the single-lane kernel (one issuing wave per SIMD), one emitting wave per lane every round.  Built,
// bit-identical on every pair test, and MEASURED SLOWER on config 3 (same box, alternating: 9.12 against 8.78 ms with
// every target "fast", 14.15 against 13.67 ms at the product's precisions; profiles/r03_ab_pair_tiles.txt): the round
// barrier is not what the pair kernel waits for (the probe said the same).  A/B knob only.
#ifndef IQA_RING_PAIR_TILES
#define IQA_RING_PAIR_TILES 1
#endif
#ifndef IQA_RING_PAIR_ROUNDS
#define IQA_RING_PAIR_ROUNDS 3  // ring depth of the lane-pair kernel in rounds (2..5).  Its speed does not depend on it (2, 3, 5:
                                // 9.19 / 9.08 / 9.08 ms at config 3); at 3 a workgroup leaves 59 KB of a CU's LDS to the small
                                // kernels of the previous capture's tail (the resampler wants 20), at 5 only 3
#endif

// 1: the kernels with loader waves over contiguous slots (<= IQA_RING_LOADERS_MAX_KS k steps: config 2) stage the capture as
// BYTE PLANES: four loader waves fetch the tiles into registers (global_load_dwordx4, three rounds ahead), split every int16
// into its high byte and its biased low byte ONCE and write the two planes into the slot; the multiplying waves read their
// matrix operands straight out of the planes.  With raw tiles in LDS (0: LDS-DMA, the older form) each of a parity's four
// multiplying waves split the same fragment again: 12 VALU instructions per k step and wave beside 3 MFMAs -- measured on the
// sustained config-2 loop as 8.5 % of the kernel's time (diagnostic build "no byte splits", profiles/r03_sustained_ablation_splits.txt).
#ifndef IQA_RING_SPLIT_STAGE
#define IQA_RING_SPLIT_STAGE 1
#endif
#ifndef IQA_RING_HALF_STEP
#define IQA_RING_HALF_STEP 1  // 0: the last k step of a row is always a whole 32x32x32 MFMA (A/B)
#endif
#ifndef IQA_RING_PD
#define IQA_RING_PD 2
#endif
#ifndef IQA_RING_SPLIT_CONTIG
#define IQA_RING_SPLIT_CONTIG 0  // 1: a parity's two loader waves take the first and the second half of a tile instead of every second piece (A/B)
#endif
#ifndef IQA_RING_PAIR_NLOADERS
#define IQA_RING_PAIR_NLOADERS 4
#endif
#ifndef IQA_RING_PAIR_LOADERS
#define IQA_RING_PAIR_LOADERS 1  // 0: lane pairs keep their issuing / emitting multiplying waves and LDS-DMA (A/B)
#endif
#ifndef IQA_RING_SPLIT_F3_MAX_KS
#define IQA_RING_SPLIT_F3_MAX_KS 0  // three rounds of loads in flight up to this many k steps, two beyond.  Two everywhere: config 2's
                                    // kernel measures the same with two and three (0.547 / 0.549 ms, profiles/r03b_ab_rounds_in_flight.txt),
                                    // and with two a workgroup of the 7-k-step kernel takes 3 x 120 registers per SIMD instead of 3 x 152:
                                    // the 80-register kernels of a capture's tail fit beside it
#endif

#include <atomic>
#include <mutex>
#include <cmath>
#include <type_traits>

namespace iqa {

typedef __attribute__((address_space(3))) void ring_lds_t;

constexpr int RG_WAVES = 8;

constexpr int RG_MAX_KS = 16;
constexpr int RG_ROWS_MAX_KS_C = 11;  // most k steps a pass of the row-staged kernels takes
constexpr int RG_W = 512;      // sliding window of output sums (positions mod 512), per component
constexpr int RG_GUARD = 64;   // a tile's scatter reaches at most 59 positions past its lane base: aliases of slots 0..63
constexpr int RG_AS = RG_W + RG_GUARD;
constexpr int RG_ACC_BYTES = 2 * RG_AS * 8;  // sized for the 64-bit sums; the 32-bit form uses half of it
constexpr int RG_EMIT_WAVE = 2;  // (rt 2, parity 0): not an issuing wave
// Parity-1 waves scatter a tile one round late (at the start of the next round, while their SIMD partner of parity 0
// is already multiplying -- the two waves of a SIMD then alternate between the matrix pipe and the LDS instead of
// meeting in both), so a group of 64 outputs is final three rounds after its first tile.
constexpr int RG_EMIT_LAG = 3;
constexpr unsigned int RG_PACE_AHEAD = 2;  // publications (of every second round) a workgroup may be in front of its range's slowest
constexpr int RG_PACE_SPINS = 20000;
constexpr int RG_PACE_WORDS = 16384;       // words of the pacing buffer: ranges x units of a launch must fit
constexpr int RG_PAIR_IDLE = 1 << 28;  // MfmaArgs::pair_shift of the half of a pair that has no lane

// One LDS-DMA instruction: 64 lanes x 16 bytes, global -> LDS (global_load_lds_dwordx4).  The 16-byte form exists on
// gfx950 only; the HOST pass of the compilation checks the builtin's size argument against ITS target and, inside
// function templates, turns that into spurious "no matching function" errors -- it never needs the body.
__device__ __forceinline__ void ring_dma16(const void *src, ring_lds_t *dst)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(src, dst, 16, 0, IQA_RING_DMA_AUX);
#else
    (void)src;
    (void)dst;
#endif
}

__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) const void *)p));
}

// Two ways a tile (32 data rows) sits in a ring slot:
//   contiguous (ROWS = false): the 32*D frames of the tile, one contiguous run of the capture, fetched by 2*KS + 1
//     DMA instructions of 1 KiB (64 lanes x 16 bytes) -- needs 16-byte aligned rows (D % 4 == 0) and all k steps of a
//     row in one pass (KS <= 16).  In LDS the rows sit at a pitch of an ODD number of 16-byte units (the row's own D/4
//     units, plus one of padding when that is even): the fragment reads of a wave -- ds_read_b128, lane = row, served
//     in four groups of 16 lanes -- then touch 16 different bank quads per group.  At the capture's own pitch they are
//     2-way conflicts at D = 104 and 4-way at D = 208 (16 LDS cycles per wave-instruction instead of 4: with 26 of them
//     per wave and tile the LDS, not the matrix pipe, bounded a CU as soon as the data came from L2).  An LDS-DMA lane
//     writes to (instruction base + 16 * lane) but fetches from any address, so the padding costs nothing but the
//     per-lane source offsets (`ring_src_off`, computed once per workgroup);
//   row-staged (ROWS = true): row j's k-step range of THIS pass, 64*KS bytes, at a pitch of 64*KS + 16 bytes (16 bytes
//     of padding: rows 4 banks apart, conflict-free reads), filled 64 units per DMA instruction like the contiguous slots (RingGeo::NI_ROWS).  Any D, and the
//     k-step ranges of a long row (D = 521: 33 k steps in three passes of 11) each fetch only their own third of every
//     row, so three passes read the capture about once instead of three times.  Always with loader waves.
// U8 (row-staged only): uint8 I/Q captures -- a k step is 32 bytes of a row, one 16-byte fragment per lane, the
// offset-binary bytes become int8 with one XOR, and there is a single data piece: two MFMAs per k step (q1*v, q2*v).
// PAIR (contiguous slots without loader waves only): the workgroup holds TWO lanes -- the four waves of parity 0 the tap
// rows of one, the four of parity 1 those of the other -- and all eight read the SAME staged tile: one tile per round
// instead of two, so the ring is twice as deep in rounds at the same LDS (five rounds instead of two at 13 k steps: the
// refill of a slot has four rounds to land instead of one) and a tile crosses L2 -> LDS once per two lanes; two windows
// of sums.  The lanes of a pair share their tap-row group index (the staged stream depends on it).
template <int KS, bool ROWS, bool U8 = false, bool PAIR = false>
struct RingGeo {
    static_assert(ROWS || !U8, "uint8 captures use row-staged slots");
    static_assert(!PAIR || (!ROWS && !U8), "lane pairs: contiguous slots only");
    static constexpr bool PAIR1 = PAIR && IQA_RING_PAIR_TILES == 1;  // lane pairs, one tile per round
    static constexpr bool PAIR2 = PAIR && IQA_RING_PAIR_TILES != 1;  // lane pairs, two tiles per round (every wave takes both)
    static constexpr int TPR = PAIR1 ? 1 : 2;  // tiles per round
    static constexpr int ACCS = PAIR ? 2 : 1;  // windows of sums
    static constexpr int KBYTES = U8 ? 32 : 64;  // bytes of a row per k step
    static constexpr int PITCH = KBYTES * KS + 16;
    static constexpr bool PADDED = KS <= 13;  // contiguous slots: rows at an odd pitch in LDS (conflict-free fragment reads)
    static constexpr int NI = 2 * KS + 1;  // contiguous slots: 1 KiB DMA instructions per tile (32 rows at a padded pitch)
    // row-staged slots are filled like the contiguous ones: 64 consecutive 16-byte units of the slot per DMA instruction,
    // every lane fetching ITS unit from wherever it lies in the capture (unit q = unit q % UNITS_ROW of row q / UNITS_ROW; a
    // row's padding unit and the units behind row 31 re-fetch a neighbour) -- 32 (4 KS + 1) / 64 instructions per tile
    // instead of one per row with 4 KS of 64 lanes active (2 k steps: 5 instead of 32)
    static constexpr int UNITS_ROW = PITCH / 16;
    static constexpr int NI_ROWS = (32 * UNITS_ROW + 63) / 64;
    static constexpr int SLOT_RAW = ROWS ? (32 * PITCH > 1024 * NI_ROWS ? 32 * PITCH : 1024 * NI_ROWS) : 1024 * NI;
    // extra waves that feed the ring and emit (a wave then has 168 registers).  Lane pairs: with byte-plane staging up to 13 k
    // steps -- four loader waves share the round's ONE tile, the first two emit a lane each
    static constexpr bool LOADERS = (!PAIR && (ROWS || KS <= IQA_RING_LOADERS_MAX_KS)) ||
                                    (PAIR1 && IQA_RING_SPLIT_STAGE != 0 && IQA_RING_PAIR_LOADERS != 0 && KS <= 13);
    // byte-plane staging (IQA_RING_SPLIT_STAGE): a slot is [high bytes: 32 rows at PLANE_PITCH][biased low bytes: ditto], the
    // same 1024 * NI bytes; the pitch is an odd number of 16-byte units (conflict-free ds_read_b128, lane = row).  FOUR
    // loader waves, two per parity (wave `half` of a parity takes the tile's 1 KiB pieces 2 j + half, j < KS), SPLIT_F rounds
    // of loads in flight in their registers (4 KS SPLIT_F of them), two slots per parity in LDS (one read, one written).
    static constexpr bool SPLIT = LOADERS && !U8 && (IQA_RING_SPLIT_STAGE != 0);
    static constexpr int PLANE_PITCH = 32 * KS + 16;
    static constexpr int SPLIT_F = ((PAIR && IQA_RING_PAIR_NLOADERS == 4) || KS <= IQA_RING_SPLIT_F3_MAX_KS) ? 3 : 2;  // (a loader wave has 168 registers: 4 KS SPLIT_F of data, 2 KS of offsets, the emission)
    // (lane pairs: IQA_RING_PAIR_NLOADERS loader waves, 2 or 4.  Two: ten waves per workgroup -- SIMDs 2 and 3 hold two waves
    // each and keep 176 registers free, room for a 256-thread workgroup of the small kernels of a capture's tail)
    static constexpr int NLOADERS = SPLIT ? (PAIR ? IQA_RING_PAIR_NLOADERS : 4) : (LOADERS ? 2 : 0);
    static constexpr int SLOT = SPLIT ? 64 * PLANE_PITCH : SLOT_RAW;
    static constexpr int NDMA = ROWS ? NI_ROWS : (LOADERS ? NI : KS + 1);  // DMAs per issuing wave and round
    static constexpr int FIT = (160 * 1024 - ACCS * RG_ACC_BYTES) / (TPR * SLOT);
    static constexpr int RMAX = 63 / NDMA + 2;  // (R - 2) * NDMA must fit the 6-bit vmcnt
    static constexpr int R0 = FIT < RMAX ? FIT : RMAX;
    static constexpr int RCAP = PAIR ? IQA_RING_PAIR_ROUNDS : IQA_RING_ROUNDS_MAX;
    static constexpr int R = SPLIT ? 2 : R0 > RCAP ? RCAP : (R0 < 2 ? 2 : R0);  // rounds (of two tiles; PAIR: of one) the ring holds
    static constexpr int THREADS = (RG_WAVES + NLOADERS) * kWave;
    static constexpr int LDS_BYTES = R * TPR * SLOT + ACCS * RG_ACC_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "ring + window exceed LDS");
    static_assert((R - 2) * NDMA <= 63, "vmcnt is a 6-bit counter");
};

struct RingCtx {
    char *smem;
    int *s_acc;           // [Sre | Sim], RG_AS entries each, output position p at slot p mod RG_W.  ACC64: one int64
                          // S1*2^32 + S2 per entry (16-bit taps, exact); else one int32 256*S1 + S2 (~14-bit taps)
    const char *stream0;  // row-staged slots: this lane's source byte of row 0 of tile 0; contiguous: tile 0's first byte
    long long tile_bytes, i0, m0;
    int tiles, rounds, cnt, lane_off, rt, cp, col, h, lane;
    int row_units, pitch_units;  // contiguous slots: 16-byte units per data row in the capture / in LDS (odd)
    int tshift;  // lane pairs: this wave's own tile t is the staged tile of round t + tshift (0 without pairs)
    int extra;   // lane pairs: rounds the workgroup runs beyond a lane's own tiles (the second lane works that far behind)
    int half_last;  // byte-plane kernels, contiguous slots: the row's last k step holds <= 16 values -- it is a 32x32x16 MFMA
};

// Source offset (bytes from the tile's first byte) of the 16 bytes lane `lane` of DMA instruction `idx` fetches: the
// instruction fills LDS units 64*idx .. 64*idx + 63 of the slot, unit q holds unit q % pitch of row q / pitch; the
// padding unit of a row and the units behind row 31 re-fetch a neighbour (they are read, if at all, against zero taps).
__device__ __forceinline__ int ring_src_off(int idx, int lane, int row_units, int pitch_units)
{
    const int q = 64 * idx + lane;
    int r = q / pitch_units;
    int u = q - r * pitch_units;
    if (r > 31) {
        r = 31;
        u = row_units - 1;
    }
    return (r * row_units + min(u, row_units - 1)) * 16;
}

// Emission state of the emitting wave: lane l owns position 64 k + 1 + l of group k; its rotation advances by
// the host-computed step of 64 outputs per group (float64 recurrence, ~1e-16 per step).
struct RingEmit {
    double wc, ws;
};

// A group of 64 finished outputs leaves the window in two steps, so that the multiplying wave that also emits (the
// kernels without loader waves) can put its own matrix work between them: `ring_emit_load` reads the sums out of the
// window (and the partial sums of earlier passes out of memory) and clears the slots; `ring_emit_store` -- a dozen
// k steps later, when those reads have long returned -- scales, rotates and stores.  Back to back they are the one-step
// emission of the loader waves.  (Done in one piece at the top of a round, the chain LDS read -> wait -> float64 maths ->
// store made the emitting wave ~500 cycles late for its tile, and seven waves waited for it at every barrier.)
struct RingEmitRegs {
    double v_re, v_im;  // 256*S1 + S2 per component
    double2 pr;         // partial sums of the earlier passes (0 when there are none)
    int i;              // output index inside the block
};

template <bool ACC64, bool EARLY>
__device__ __forceinline__ void ring_emit_load(const MfmaArgs &a, const RingCtx &c, int k, RingEmitRegs &g)
{
    const int pos = 64 * k + 1 + c.lane;
    g.i = pos - MF_Q;
    const int s = pos & (RG_W - 1);
    if constexpr (ACC64) {
        long long *acc = reinterpret_cast<long long *>(c.s_acc);
        long long sr = acc[s], si = acc[RG_AS + s];
        if (s < RG_GUARD) {
            sr += acc[RG_W + s];
            si += acc[RG_AS + RG_W + s];
            acc[RG_W + s] = 0;
            acc[RG_AS + RG_W + s] = 0;
        }
        acc[s] = 0;  // the slot is scattered into again 8 rounds from now
        acc[RG_AS + s] = 0;
        // the adds were (S1 << 32) + sign-extended S2: the low word is S2 itself (|S2| < 2^31), the rest is S1
        const int s2r = static_cast<int>(sr), s2i = static_cast<int>(si);
        const long long s1r = (sr - s2r) >> 32, s1i = (si - s2i) >> 32;
        g.v_re = static_cast<double>(s1r) * 256.0 + static_cast<double>(s2r);
        g.v_im = static_cast<double>(s1i) * 256.0 + static_cast<double>(s2i);
    } else {
        int *acc = c.s_acc;
        int sr = acc[s], si = acc[RG_AS + s];
        if (s < RG_GUARD) {
            sr += acc[RG_W + s];
            si += acc[RG_AS + RG_W + s];
            acc[RG_W + s] = 0;
            acc[RG_AS + RG_W + s] = 0;
        }
        acc[s] = 0;
        acc[RG_AS + s] = 0;
        g.v_re = static_cast<double>(sr);
        g.v_im = static_cast<double>(si);
    }
    g.pr = make_double2(0.0, 0.0);
    // EARLY: the earlier passes' partial sums are requested here, a dozen k steps before they are used (the emitting
    // MULTIPLYING wave: it issues no LDS-DMAs, a wait for this load costs it nothing).  A loader wave must not: the wait
    // for the load would be a vmcnt(0) in the middle of its counted DMA sequence -- the whole ring drained once per round
    // (measured: 0.653 against 0.585 ms at D = 104).  It loads them where it uses them, inside the branch.
    if constexpr (EARLY) {
        if (a.partial_in != nullptr && g.i >= 0 && g.i < c.cnt) g.pr = a.partial_in[c.i0 + g.i];
    }
}

template <bool EARLY>
__device__ __forceinline__ void ring_emit_store(const MfmaArgs &a, const RingCtx &c, RingEmit &e, const RingEmitRegs &g)
{
    const int i = g.i;
    if (i >= 0 && i < c.cnt) {
        double d_re = mfma_scaled_sum(g.v_re, a.c_re, a.unit);
        double d_im = mfma_scaled_sum(g.v_im, a.c_im, a.unit);
        if (a.partial_in != nullptr) {
            const double2 pr = EARLY ? g.pr : a.partial_in[c.i0 + i];
            d_re = __dadd_rn(d_re, pr.x);
            d_im = __dadd_rn(d_im, pr.y);
        }
        if (!a.finalize) {
            if (a.raw_partials)  // the integer sums themselves (exact: int32 by construction); iqa_mfma_combine scales them
                reinterpret_cast<int2 *>(a.partial_out)[c.i0 + i] = make_int2(static_cast<int>(g.v_re), static_cast<int>(g.v_im));
            else
                a.partial_out[c.i0 + i] = make_double2(d_re, d_im);
        } else {
            a.out[c.i0 + i] = mfma_finish(d_re, d_im, a.conj_sum, a.rotate, e.wc, e.ws, a.sc_re, a.sc_im);
        }
    }
    const double nc = e.wc * a.rot64_re - e.ws * a.rot64_im;
    e.ws = fma(e.wc, a.rot64_im, e.ws * a.rot64_re);
    e.wc = nc;
}

template <bool ACC64>
__device__ __forceinline__ void ring_emit_group(const MfmaArgs &a, const RingCtx &c, RingEmit &e, int k)
{
    RingEmitRegs g;
    ring_emit_load<ACC64, false>(a, c, k, g);
    ring_emit_store<false>(a, c, e, g);
}

template <int KS, bool ROWS, bool U8, bool PAIR = false>
__device__ __forceinline__ void ring_wait_and_barrier(int younger)
{
    // an issuing wave's DMAs of round r have landed once only those of the younger rounds are outstanding
    using G = RingGeo<KS, ROWS, U8, PAIR>;
    constexpr int R = G::R, N = G::NDMA;
    if (R >= 5 && younger == 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(R >= 5 ? 3 * N : 0) : "memory");
    else if (R >= 4 && younger == 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(R >= 4 ? 2 * N : 0) : "memory");
    else if (R >= 3 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(R >= 3 ? N : 0) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
}

// A loader wave (parity cp): per round, wait for its DMAs of this round, join the barrier, refill the slot the
// previous round has left with the tile R-1 rounds ahead (all 2*KS chunks), and emit the group of 64 outputs that
// became complete two rounds ago when that group's parity is its own.
template <int KS, int DBG, bool ACC64, bool ROWS, bool U8>
__device__ __forceinline__ void ring_loader(const MfmaArgs &a, const RingCtx &c)
{
    using G = RingGeo<KS, ROWS, U8>;
    constexpr int R = G::R, SLOT = G::SLOT;
    constexpr bool STREAM = !(DBG & 16);
    const int cp = c.cp;
    int soff[ROWS ? G::NI_ROWS : G::NI];
    if constexpr (!ROWS) {
#pragma unroll
        for (int i = 0; i < G::NI; ++i) soff[i] = ring_src_off(i, c.lane, c.row_units, c.pitch_units);
    } else {
        const int row_bytes = static_cast<int>(c.tile_bytes >> 5);
#pragma unroll
        for (int i = 0; i < G::NI_ROWS; ++i) {
            const int q = 64 * i + c.lane;
            int r = q / G::UNITS_ROW;
            int u = q - r * G::UNITS_ROW;
            if (r > 31) {
                r = 31;
                u = G::UNITS_ROW - 2;
            }
            soff[i] = r * row_bytes + min(u, G::UNITS_ROW - 2) * 16;  // (the last data unit again for the padding unit)
        }
    }
    auto issue_tile = [&](int tile, int slot) {
        const char *src = c.stream0 + static_cast<long long>(min(tile, c.tiles - 1)) * c.tile_bytes;
        char *dst = c.smem + (slot * 2 + cp) * SLOT;
        if constexpr (ROWS) {
#pragma unroll
            for (int i = 0; i < G::NI_ROWS; ++i)
                ring_dma16(src + soff[i], (ring_lds_t *)(dst + i * 1024));
        } else {
#pragma unroll
            for (int i = 0; i < G::NI; ++i)
                ring_dma16(src + soff[i], (ring_lds_t *)(dst + i * 1024));
        }
    };
    if (STREAM) {
#pragma unroll
        for (int rr = 0; rr < R - 1; ++rr) issue_tile(2 * rr + cp, rr);
    }
    RingEmit em{1.0, 0.0};
    double st_re = a.rot64_re * a.rot64_re - a.rot64_im * a.rot64_im, st_im = 2.0 * a.rot64_re * a.rot64_im;  // 128 outputs
    if (a.finalize && a.rotate) {
        const unsigned long long m = static_cast<unsigned long long>(c.m0 + (64 * cp + 1 + c.lane - MF_Q));  // group cp
        const unsigned long long ph = a.rot_base + m * a.rot_step;
        sincospi(2.0 * (static_cast<double>(ph >> 11) * (1.0 / 9007199254740992.0)), &em.ws, &em.wc);
    }
    MfmaArgs a2 = a;  // ring_emit_group advances the rotation by (rot64_re, rot64_im): this wave owns every other group
    a2.rot64_re = st_re;
    a2.rot64_im = st_im;
    int slot = 0;
    for (int r = 0; r < c.rounds; ++r) {
        if (STREAM) ring_wait_and_barrier<KS, ROWS, U8>(min(R - 2, c.rounds - 1 - r));
        else asm volatile("s_barrier" ::: "memory");
        if (STREAM && r + R - 1 < c.rounds) issue_tile(2 * (r + R - 1) + cp, (slot == 0) ? R - 1 : slot - 1);
        if (r >= RG_EMIT_LAG && ((r - RG_EMIT_LAG) & 1) == cp) {
            // every wave has ISSUED the LDS adds of its round r-2 tile before this barrier (parity 1 defers a tile's
            // adds to the start of the next round); a wave has at most 15 LDS operations outstanding and they complete
            // in order, so the adds of the tiles of round r-3 and earlier -- everything that reaches positions
            // <= 64 (r-2) -- have landed.
            asm volatile("" ::: "memory");
            ring_emit_group<ACC64>(a2, c, em, r - RG_EMIT_LAG);
            asm volatile("" ::: "memory");
        }
        slot = (slot + 1 == R) ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int k_last = (c.cnt + 62) >> 6;
    for (int k = max(c.rounds - RG_EMIT_LAG, 0); k <= k_last; ++k)
        if ((k & 1) == cp) ring_emit_group<ACC64>(a2, c, em, k);
}

// A loader wave of the byte-plane kernels (RingGeo::SPLIT): parity cp, half `half`.  Round r: behind the barrier it waits
// for ITS pieces of the tile of round r + 1 (requested SPLIT_F rounds ago), splits them and writes the planes of slot
// (r + 1) % 2 -- the slot round r - 1 has just left --, requests the tile of round r + 1 + SPLIT_F into the registers that
// became free, and (half 0) emits like ring_loader.  Plain loads: the compiler counts vmcnt itself, the emission's store
// included.  Pieces beyond the tile's last unit (a tile is D / 8 pieces, a wave pair fetches 2 KS) re-fetch the last unit
// and are not written.
struct __attribute__((packed, aligned(4))) ring_raw16 {  // 16 bytes of the capture: dword-aligned (the stream starts at frame row*D + 1)
    int x, y, z, w;
};

// PAIR: the four loader waves (HALF 0..3) share the round's one tile, piece 4 j + HALF each; waves 0 and 1 emit lane 0's and
// lane 1's outputs (c.cp = the lane; a, c.s_acc, c.tshift are that lane's), every group of their lane.
template <int KS, int DBG, bool ACC64, bool ROWS, int HALF, bool PAIR = false>
__device__ __forceinline__ void ring_loader_split(const MfmaArgs &a, const RingCtx &c)
{
    constexpr bool EMITTER = PAIR ? HALF < 2 : HALF == 0;
    // pieces of a tile this wave takes: every second one, or -- from 12 k steps on, where 8 KS registers of data in flight
    // and the emission's float64 state do not fit one wave -- the emitting wave (half 0) the first KS - 2, the other the rest
    constexpr bool UNEVEN = !PAIR && (KS >= 12 || IQA_RING_SPLIT_CONTIG != 0);
    constexpr int P0 = KS >= 12 ? KS - 2 : KS;
    constexpr int STRIDE = PAIR ? RingGeo<KS, ROWS, false, PAIR>::NLOADERS : 2;
    constexpr int NP = UNEVEN ? (HALF ? 2 * KS - P0 : P0) : (2 * KS + STRIDE - 1) / STRIDE;
    auto piece = [](int j) { return UNEVEN ? (HALF ? P0 + j : j) : STRIDE * j + HALF; };
    using G = RingGeo<KS, ROWS, false, PAIR>;
    constexpr int SLOT = G::SLOT, PP = G::PLANE_PITCH, F = G::SPLIT_F;
    constexpr bool STREAM = !(DBG & 16);
    const int cp = c.cp;
    // 16-byte units of a data row that are staged: the whole row (contiguous slots: rows follow one another in the capture)
    // or this pass's 4 KS units of it (row-staged slots); a tile is 32 rows of them, unit q = unit q % upr of row q / upr
    const int upr = ROWS ? 4 * KS : c.row_units;
    const int row_bytes = static_cast<int>(c.tile_bytes >> 5);
    // piece j of this wave is unit q = 64 piece(j) + lane = unit u of row `row`.  Up to 8 k steps its source and
    // destination offsets sit in two tables; longer rows keep 16 bits (row, u) per piece and spend a few
    // VALU instructions on them -- 2 KS registers the loader does not have beside 8 KS of data in flight.  (Walking (row, u) from
    // piece to piece instead, with its lane-divergent carry loop, made the loaders the slowest waves of the workgroup:
    // 1.55 against 1.09 ms at 13 k steps.)
    constexpr bool TABLE = KS <= 8;
    int doff[TABLE ? NP : 1], soff[TABLE ? NP : 1];
    unsigned rowu[TABLE ? 1 : (NP + 1) / 2];  // two pieces per word: u (6 bits: a row holds <= 64 units), row (5), "not written" (1)
    if constexpr (!TABLE) {
#pragma unroll
        for (int j = 0; j < (NP + 1) / 2; ++j) rowu[j] = 0;
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int q = 64 * piece(j) + c.lane;
        const int row = q / upr, u = q - row * upr;
        if constexpr (TABLE) {
            doff[j] = row < 32 ? row * PP + 8 * u : -1;
            soff[j] = row < 32 ? row * row_bytes + 16 * u : 31 * row_bytes + 16 * (upr - 1);
        } else {
            const unsigned f = row < 32 ? static_cast<unsigned>((row << 6) | u) : static_cast<unsigned>((31 << 6) | (upr - 1) | (1 << 11));
            rowu[j >> 1] |= f << (16 * (j & 1));
        }
    }
    int4 buf[F][NP];
    auto request = [&](int round, auto set) {
        constexpr int S = decltype(set)::value;
        // (PAIR: the stream runs on past this lane's own last tile for the partner that works pair_extra rounds behind)
        const char *src = c.stream0 + static_cast<long long>(PAIR ? min(round, c.rounds - 1) : min(2 * round + cp, c.tiles - 1)) * c.tile_bytes;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            int so;
            if constexpr (TABLE) so = soff[j];
            else {
                const unsigned f = rowu[j >> 1] >> (16 * (j & 1));
                so = static_cast<int>((f >> 6) & 31u) * row_bytes + 16 * static_cast<int>(f & 63u);
            }
            const ring_raw16 t = *reinterpret_cast<const ring_raw16 *>(src + so);
            buf[S][j] = make_int4(t.x, t.y, t.z, t.w);
        }
    };
    auto stage = [&](int slot, auto set) {
        constexpr int S = decltype(set)::value;
        char *hi0 = c.smem + (PAIR ? slot : slot * 2 + cp) * SLOT;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int4 d = buf[S][j];
            int2 hi, lo;
            hi.x = __builtin_amdgcn_perm(d.y, d.x, 0x07050301);
            hi.y = __builtin_amdgcn_perm(d.w, d.z, 0x07050301);
            lo.x = __builtin_amdgcn_perm(d.y, d.x, 0x06040200) ^ 0x80808080;
            lo.y = __builtin_amdgcn_perm(d.w, d.z, 0x06040200) ^ 0x80808080;
            int dq;
            if constexpr (TABLE) dq = doff[j];
            else {
                const unsigned f = rowu[j >> 1] >> (16 * (j & 1));
                dq = (f & 2048u) ? -1 : static_cast<int>((f >> 6) & 31u) * PP + 8 * static_cast<int>(f & 63u);
            }
            if (dq >= 0) {
                *reinterpret_cast<int2 *>(hi0 + dq) = hi;
                *reinterpret_cast<int2 *>(hi0 + 32 * PP + dq) = lo;
            }
        }
    };
    // prologue: the tile of round 0 into slot 0 (before the block's first barrier), rounds 1 .. F requested
    if (STREAM) {
        request(0, std::integral_constant<int, 0>{});
        stage(0, std::integral_constant<int, 0>{});
        request(1, std::integral_constant<int, 1 % F>{});
        request(2, std::integral_constant<int, 2 % F>{});
        if constexpr (F == 3) request(3, std::integral_constant<int, 0>{});
    }
    static_assert(F == 2 || F == 3, "the prologue and the unrolled loop below are written for two or three rounds in flight");
    RingEmit em{1.0, 0.0};
    double st_re = a.rot64_re * a.rot64_re - a.rot64_im * a.rot64_im, st_im = 2.0 * a.rot64_re * a.rot64_im;  // 128 outputs
    if (EMITTER && a.finalize && a.rotate) {
        const unsigned long long m = static_cast<unsigned long long>(c.m0 + ((PAIR ? 0 : 64 * cp) + 1 + c.lane - MF_Q));  // its first group
        const unsigned long long ph = a.rot_base + m * a.rot_step;
        sincospi(2.0 * (static_cast<double>(ph >> 11) * (1.0 / 9007199254740992.0)), &em.ws, &em.wc);
    }
    MfmaArgs a2 = a;  // ring_emit_group advances the rotation by (rot64_re, rot64_im): without pairs this wave owns every other group
    if constexpr (!PAIR) {
        a2.rot64_re = st_re;
        a2.rot64_im = st_im;
    }
    // the group of 64 outputs this wave emits in round r, or -1.  PAIR: a lane's group k (its tiles 2k, 2k + 1) in round
    // 2k + 5 (lane 0) / 2k + 6 (lane 1) of the lane's own count, as ring_main's emitting waves did
    auto group_of = [&](int r) {
        if constexpr (PAIR) {
            const int re = r - c.tshift, lag = 5 + cp;
            return (re >= lag && ((re - lag) & 1) == 0) ? (re - lag) >> 1 : -1;
        } else {
            return (r >= RG_EMIT_LAG && ((r - RG_EMIT_LAG) & 1) == cp) ? r - RG_EMIT_LAG : -1;
        }
    };
    // The partial sums of the earlier passes (a long row's k-step ranges) are requested ONE ROUND AHEAD of their emission and
    // in front of that round's tile request: loads return in order, so waiting for a load issued behind the tile requests
    // -- where the emission uses it -- would wait for every tile in flight: the prefetch drained every other round.
    // (Up to 11 k steps; at 12 and 13 the four registers are not there, and the partial sums are loaded where they are used.)
    constexpr bool PR_AHEAD = PAIR || KS <= 11;
    double2 pr_next = make_double2(0.0, 0.0);
    auto round_body = [&](int r, auto set) {
        // (the planes of round r were written before this barrier: lgkmcnt(0) in front of it)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (STREAM && r + 1 < c.rounds) stage((r + 1) & 1, set);
        const double2 pr_now = pr_next;
        if (PR_AHEAD && EMITTER && a.partial_in != nullptr) {
            const int kn = group_of(r + 1);
            if (kn >= 0) {
                const int i = 64 * kn + 1 + c.lane - MF_Q;
                pr_next = (i >= 0 && i < c.cnt) ? a.partial_in[c.i0 + i] : make_double2(0.0, 0.0);
            }
        }
        if (STREAM) request(r + 1 + F, set);  // (beyond the last tile: the last tile again, never staged)
        const int k = EMITTER ? group_of(r) : -1;
        if (k >= 0) {
            asm volatile("" ::: "memory");
            if constexpr (PR_AHEAD) {
                RingEmitRegs g;  // (see ring_loader for why these sums are final)
                ring_emit_load<ACC64, false>(a2, c, k, g);
                g.pr = pr_now;
                ring_emit_store<true>(a2, c, em, g);
            } else {
                ring_emit_group<ACC64>(a2, c, em, k);
            }
            asm volatile("" ::: "memory");
        }
    };
    // round r stages the tile of round r + 1, which sits in register set (r + 1) % F
    int r = 0;
    if constexpr (F == 3) {
        for (; r + 3 <= c.rounds; r += 3) {
            round_body(r, std::integral_constant<int, 1>{});
            round_body(r + 1, std::integral_constant<int, 2>{});
            round_body(r + 2, std::integral_constant<int, 0>{});
        }
        if (r < c.rounds) {
            round_body(r, std::integral_constant<int, 1>{});
            if (r + 1 < c.rounds) round_body(r + 1, std::integral_constant<int, 2>{});
        }
    } else {
        for (; r + 2 <= c.rounds; r += 2) {
            round_body(r, std::integral_constant<int, 1>{});
            round_body(r + 1, std::integral_constant<int, 0>{});
        }
        if (r < c.rounds) round_body(r, std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (EMITTER) {
        const int k_last = (c.cnt + 62) >> 6;
        if constexpr (PAIR) {
            const int own = c.rounds - c.tshift, lag = 5 + cp;  // rounds in this lane's own count
            const int k_next = own > lag ? ((own - 1 - lag) >> 1) + 1 : 0;
            if (c.tshift < RG_PAIR_IDLE)  // (the idle half of a pair without a second lane emits nothing)
                for (int k = k_next; k <= k_last; ++k) ring_emit_group<ACC64>(a2, c, em, k);
        } else {
            for (int k = max(c.rounds - RG_EMIT_LAG, 0); k <= k_last; ++k)
                if ((k & 1) == cp) ring_emit_group<ACC64>(a2, c, em, k);
        }
    }
}

// The main loop of a multiplying wave.  ISSUER (kernels without loader waves): this wave also feeds the ring
// (chunks 2i + (rt & 1) of its parity's slot).  EMIT (ditto): this wave also converts, rotates and stores the 64
// outputs that became complete two rounds ago.
// DBG bits (diagnostic instantiations only): 1 = no scatter, 16 = no data stream, 32 = no matrix work, 4 = every second
// multiplying wave without fragment reads and byte splits, 8 = no byte splits.
// SKIP: what this wave does with the q2*hi product (the third matrix instruction of a k step): 0 = always computes it;
// 2 = never (its lane's low tap byte is zero throughout: the first lane of a "fine" / "full" tap-row group); 1 = asks the
// lane's flag at run time (both bodies in the code: ~30 registers more, so only the kernels of such launches carry it).
template <int KS, int DBG, bool ACC64, bool ROWS, bool U8, bool ISSUER, bool EMIT, bool DEFER, bool PAIR, typename SKIPT, bool HALF = false>
__device__ __forceinline__ void ring_main(const MfmaArgs &a, const RingCtx &c, const v4i_t (&fq)[KS][2], SKIPT)
{
    constexpr int SKIP = SKIPT::value;
    using G = RingGeo<KS, ROWS, U8, PAIR>;
    constexpr int R = G::R, SLOT = G::SLOT;
    static_assert(!(ROWS && ISSUER), "row-staged slots are always fed by loader waves");
    // PAIR: round r works on tile r (both parities: two lanes' tap rows), whose slot is ring slot r mod R; a lane's group k
    // of 64 outputs (tiles 2k, 2k + 1) is emitted by that lane's emitting wave in round 2k + 5 (parity 0) / 2k + 6 (parity
    // 1): the two lanes' emissions fall into alternate rounds
    const int RG_EMIT_LAG_PAIR = 5 + c.cp;  // (PAIR1 only)
    constexpr bool PAIR1 = G::PAIR1, PAIR2 = G::PAIR2;
    // PAIR2: a round is tiles 2r, 2r + 1 of the STAGED stream (the pair's first lane's), in slots 2 * slot + j like the
    // single-lane kernel's; every wave multiplies both with its lane's tap rows; a lane's own tiles are the staged ones of
    // tshift rounds earlier (one round per tap-row group of difference), its group k of 64 outputs is the two tiles of its
    // round k and is emitted by the lane's emitting wave in round k + RG_EMIT_LAG of that lane's count, every round.
    constexpr bool STREAM = ISSUER && !(DBG & 16);
    const int rt = c.rt, cp = c.cp;
    // the two issuing waves of a parity share the tile's 2*KS + 1 DMA instructions: wave p = rt & 1 issues numbers
    // p, p + 2, ...; both issue KS + 1 (the counted s_waitcnt wants one number for both), so the odd wave's last one
    // repeats the even wave's last (same bytes to the same place)
    // Source offsets of this wave's KS + 1 instructions: a table in registers where the rows are padded (G::PADDED);
    // the longest rows (KS >= 14, already at the 256-register limit -- a spill would put scratch loads into the vmcnt
    // sequence the counted waits rely on) keep the capture's own pitch: unit q of the slot is unit q of the tile.
    int soff[(ISSUER && G::PADDED) ? KS + 1 : 1];
    if constexpr (ISSUER && G::PADDED) {
#pragma unroll
        for (int i = 0; i <= KS; ++i) soff[i] = ring_src_off(min(2 * i + (rt & 1), 2 * KS), c.lane, c.row_units, c.pitch_units);
    } else {
        soff[0] = c.lane * 16;
    }
    const int odd_kib = (rt & 1) * 1024;
    auto issue = [&](int tile, int slot, int i) {  // `i` is a compile-time constant at every call site
        // (PAIR: the stream runs on past this lane's own last tile for the partner that works pair_extra rounds behind)
        const char *tile0 = c.stream0 + static_cast<long long>(min(tile, (PAIR1 ? c.rounds : PAIR2 ? c.tiles + 2 * c.extra : c.tiles) - 1)) * c.tile_bytes;
        char *slot0 = c.smem + (PAIR1 ? slot : slot * 2 + cp) * SLOT;
        const int at = (i < KS) ? odd_kib + i * 2048 : 2 * KS * 1024;  // instruction numbers p, p + 2, ..., then 2*KS
        if constexpr (ISSUER && G::PADDED) {
            ring_dma16(tile0 + soff[i], (ring_lds_t *)(slot0 + at));
        } else {
            ring_dma16(tile0 + soff[0] + at, (ring_lds_t *)(slot0 + at));
        }
    };
    if (STREAM) {
#pragma unroll
        for (int rr = 0; rr < R - 1; ++rr)
#pragma unroll
            for (int i = 0; i <= KS; ++i) issue(PAIR1 ? rr : 2 * rr + cp, rr, i);
    }
    RingEmit em{1.0, 0.0};
    if (EMIT && a.finalize && a.rotate) {
        const unsigned long long m = static_cast<unsigned long long>(c.m0 + (1 + c.lane - MF_Q));  // group 0
        const unsigned long long ph = a.rot_base + m * a.rot_step;
        sincospi(2.0 * (static_cast<double>(ph >> 11) * (1.0 / 9007199254740992.0)), &em.ws, &em.wc);
    }
    const v16i_t zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // diagonal scatter of a finished tile into the window of sums (row = q, column = data row: output = row + column).
    // Written as inline asm on purpose: the compiler drains vmcnt to 0 before any LDS store it can see while LDS-DMAs
    // are in flight (it cannot tell the ring from the window), which would serialise the whole prefetch pipeline once
    // per round.  These adds touch the window only, never the ring.
    auto scatter = [&](int t, const v16i_t &acc1, const v16i_t &acc2) {
        const int slot_idx = ((t * 32 + c.col + 4 * c.h + 1) & (RG_W - 1)) + (rt >> 1) * RG_AS + (rt & 1) * 32;
        if constexpr (ACC64) {
            // exact for 16-bit taps: ONE 64-bit add of (S1 << 32) + sign-extended S2 per element (3 dwords through the
            // LDS data path instead of 2 x 2 for separate S1/S2 adds); the carry out of the low word is undone at emission
            const unsigned p = lds_addr(reinterpret_cast<long long *>(c.s_acc) + slot_idx);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const long long comb = (static_cast<long long>(acc1[q]) << 32) + static_cast<long long>(acc2[q]);
                const int off = 8 * ((q & 3) + 8 * (q >> 2));
                asm volatile("ds_add_u64 %0, %1 offset:%2" ::"v"(p), "v"(comb), "n"(off));
            }
        } else {
            const unsigned p = lds_addr(c.s_acc + slot_idx);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                // 256*S1 + S2 in ONE int32 (the host bounds the tap magnitudes so that this cannot overflow for any
                // input, dsp_plan.plan_mfma(acc32=True)); the shift-add is a VALU op the compiler can see, so the
                // MFMA -> VALU hazard distance is its business and the asm reads a VALU result
                const int comb = (acc1[q] << 8) + acc2[q];
                const int off = 4 * ((q & 3) + 8 * (q >> 2));
                asm volatile("ds_add_u32 %0, %1 offset:%2" ::"v"(p), "v"(comb), "n"(off));
            }
        }
    };
    // STAGGER (see IQA_RING_STAGGER): barrier 2r is the tile boundary of parity 0's round r and falls into the middle of
    // parity 1's tile of round r - 1; barrier 2r + 1 is parity 1's boundary and parity 0's mid-tile barrier.  Every wave
    // executes the same 2 * rounds + 1 barriers: parity 1 one in front of its loop, parity 0 one behind it.
    constexpr bool STAGGER = (IQA_RING_STAGGER != 0) && !G::LOADERS;
    // (the older, weaker way of keeping a SIMD's two waves out of step.  Not in the byte-plane kernels: with three waves per SIMD
    // and no byte splits it buys nothing -- config 2's kernel 0.558 either way, 13 k steps 0.975 / 0.979 ms -- and its 32 held
    // registers made the 64-bit-sum lane pairs spill: config 3 at the product's precisions 13.84 -> 12.65 ms without,
    // profiles/r03_ab_defer.txt)
    constexpr bool DEFER_ADDS = DEFER && !STAGGER && (IQA_RING_DEFER != 0) && !G::SPLIT;
    v16i_t held1 = zero16, held2 = zero16;  // DEFER_ADDS: the previous tile's sums, scattered at the start of the next round
    int held_t = -1;
    int slot = 0;
    if (STAGGER && cp == 1) asm volatile("s_barrier" ::: "memory");
    if (IQA_RING_PRIO == 1 && cp == 1) asm volatile("s_setprio 1");
    if (IQA_RING_PRIO == 2 && cp == 0) asm volatile("s_setprio 1");
    // DBG & 2 (diagnostic builds): per wave, cycles spent waiting in front of / at the round barrier and cycles between
    // barriers, summed over the rounds -> a.stamps[(workgroup * 8 + wave) * 4 + {0: wait, 1: work, 2: rounds, 3: first tile stamp}]
    unsigned long long st_wait = 0, st_work = 0, st_prev = 0;
    bool pace_on = PAIR && a.pace != nullptr && a.pace_units > 1 && c.rounds < (PAIR2 ? 4000 : 8000);  // (12 bits of publication count)
    for (int r = 0; r < c.rounds; ++r) {
        unsigned long long st0 = 0;
        if (DBG & 2) {
            st0 = __builtin_amdgcn_s_memtime();
            if (r) st_work += st0 - st_prev;
        }
        if (STREAM) ring_wait_and_barrier<KS, ROWS, U8, PAIR>(min(R - 2, c.rounds - 1 - r));
        else asm volatile("s_barrier" ::: "memory");
        if (DBG & 2) {
            st_prev = __builtin_amdgcn_s_memtime();
            st_wait += st_prev - st0;
        }
        if constexpr (PAIR && !ISSUER && !EMIT && !DEFER) {
            // Pacing (wave rt 3 of parity 0: no counted waits of its own to disturb).  The workgroups of a range share the
            // capture through their XCD's L2, which at this rate keeps a line for ~10 us = a few rounds; single-lane
            // workgroups stay that close by themselves (the one in front misses in L2 and waits for HBM, the others hit),
            // a pair's DMAs are off its critical path and nothing holds the workgroups of a range together: 1.9x - 4.3x
            // the capture in HBM reads.  So every second round a workgroup publishes its round and the one that is more
            // than RG_PACE_AHEAD publications in front of the slowest STARTED workgroup of its range waits for it -- the
            // slowest never waits, a workgroup that has not started is not waited for: no cycle.
            if (c.rt == 3 && c.cp == 0 && pace_on && (PAIR2 || (r & 1) == 0)) {  // (every second tile either way)
                const unsigned int mine = static_cast<unsigned int>(PAIR2 ? r : r >> 1);
                if (c.lane == 0) __hip_atomic_store(a.pace + a.pace_slot, (a.pace_token << 12) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int spin = 0;; ++spin) {
                    unsigned int q = mine;
                    if (c.lane < a.pace_units) {
                        const unsigned int v = __hip_atomic_load(a.pace + (a.pace_slot - a.pace_slot % a.pace_units) + c.lane, __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT);
                        if ((v >> 12) == a.pace_token) q = v & 0xFFFu;
                    }
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) q = min(q, static_cast<unsigned int>(__shfl_xor(static_cast<int>(q), o, kWave)));
                    q = static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(q)));
                    if (mine <= q + RG_PACE_AHEAD) break;
                    if (spin >= RG_PACE_SPINS) {  // (a peer that stopped moving: do without)
                        pace_on = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(32);
                }
            }
        }
        const bool pf = STREAM && (r + R - 1 < c.rounds);
        const int pf_tile = PAIR1 ? r + R - 1 : 2 * (r + R - 1) + cp;
        const int pf_slot = (slot == 0) ? R - 1 : slot - 1;  // the slot round r-1 has just left
        // The refill of that slot goes out FIRST, all KS + 1 instructions of it, before this round's matrix work: at
        // 13 k steps the ring holds two rounds only, so a DMA issued late in round r (one per k step, as this loop used
        // to do) had to land before the barrier of round r + 1 -- its whole L2/HBM latency, ~1000 cycles of a ~2500-cycle
        // round, stood exposed at every barrier (measured on the five-target launch: the matrix pipe 52 % busy with no
        // bank conflict and the LDS 23 % busy).  Issued here, a refill has the whole round to land.
        if (pf) {
#pragma unroll
            for (int i = 0; i <= KS; ++i) issue(pf_tile, pf_slot, i);
        }
        const int re = r - c.tshift;  // PAIR: the round in this lane's own count (its tile(s) of round re are staged now)
        const bool emit_now = EMIT && (PAIR1 ? (re >= RG_EMIT_LAG_PAIR && ((re - RG_EMIT_LAG_PAIR) & 1) == 0) : re >= RG_EMIT_LAG);
        RingEmitRegs eg;
        if (emit_now) {
            asm volatile("" ::: "memory");
            // see ring_loader for why these sums are final (PAIR: the group's last tile was round r - 4's, whose adds --
            // deferred by one round at most -- went out before the barrier of round r - 2)
            ring_emit_load<ACC64, true>(a, c, PAIR1 ? (re - RG_EMIT_LAG_PAIR) >> 1 : re - RG_EMIT_LAG, eg);
            asm volatile("" ::: "memory");
        }
        if (DEFER_ADDS && held_t >= 0) {
            scatter(held_t, held1, held2);
            held_t = -1;
        }
        bool store_pending = emit_now;  // the emitting wave finishes its group inside its first tile of the round
        constexpr int NTW = PAIR2 ? 2 : 1;  // tiles this wave multiplies per round
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
        const int t = PAIR1 ? re : PAIR2 ? 2 * re + j : 2 * r + cp;
        if (t >= 0 && t < c.tiles) {
            const bool store_here = store_pending;
            store_pending = false;
            const char *la = c.smem + (PAIR1 ? slot : PAIR2 ? slot * 2 + j : slot * 2 + cp) * SLOT + c.lane_off;
            // The data fragments are read PD k steps ahead by hand (an LDS-DMA is a store to LDS as far as the compiler knows,
            // so it never moves a ds_read above an earlier issue(): the refill in front of this loop is a fence for them).
            auto tile_body = [&](auto skip_low) {
                constexpr bool SKIP_LOW = decltype(skip_low)::value;  // q2 == 0 throughout: no q2*hi product
                constexpr int PD = KS < IQA_RING_PD ? KS : IQA_RING_PD;  // k steps the fragment reads run ahead of the MFMAs
                v16i_t acc1, acc2;
                if constexpr (U8) {
                    // uint8 frames: this lane's 16 bytes of a k step are 8 frames; u ^ 0x80 = u - 128 as int8
                    v4i_t du[KS];
#pragma unroll
                    for (int ks = 0; ks < PD; ++ks) du[ks] = *reinterpret_cast<const v4i_t *>(la + 32 * ks);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        v4i_t v = du[ks];
                        v.x ^= 0x80808080;
                        v.y ^= 0x80808080;
                        v.z ^= 0x80808080;
                        v.w ^= 0x80808080;
                        if (ks + PD < KS) du[ks + PD] = *reinterpret_cast<const v4i_t *>(la + 32 * (ks + PD));
                        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][0], v, ks ? acc1 : zero16, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][1], v, ks ? acc2 : zero16, 0, 0, 0);
                        if (IQA_RING_SCHED_BARRIER) __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                if constexpr (G::SPLIT) {
                    // byte planes: the operands as they lie in the slot (high bytes at la, biased low bytes 32 rows further)
                    v4i_t hh[KS], ll[KS];
                    constexpr bool half_last = HALF;  // (a property of the kernel variant: both bodies in one kernel cost ~25 registers)
                    auto fetch = [&](int ks) {  // `ks` is a compile-time constant at every call site
                        if (ks == KS - 1 && half_last) {  // this lane's 8 values of the half step: 8 h bytes into the step, not 16 h
                            const int2 h2 = *reinterpret_cast<const int2 *>(la - 8 * c.h + 32 * ks);
                            const int2 l2 = *reinterpret_cast<const int2 *>(la - 8 * c.h + 32 * G::PLANE_PITCH + 32 * ks);
                            hh[ks].x = h2.x;
                            hh[ks].y = h2.y;
                            ll[ks].x = l2.x;
                            ll[ks].y = l2.y;
                        } else {
                            hh[ks] = *reinterpret_cast<const v4i_t *>(la + 32 * ks);
                            ll[ks] = *reinterpret_cast<const v4i_t *>(la + 32 * G::PLANE_PITCH + 32 * ks);
                        }
                    };
#pragma unroll
                    for (int ks = 0; ks < PD; ++ks) fetch(ks);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const v4i_t hi = hh[ks], lo = ll[ks];
                        if (ks + PD < KS) fetch(ks + PD);
                        if (DBG & 32) {
                            asm volatile("" ::"v"(hi), "v"(lo));
                        } else if (ks == KS - 1 && half_last) {
                            auto pack = [](int x, int y) { return static_cast<long>((static_cast<unsigned long long>(static_cast<unsigned>(y)) << 32) | static_cast<unsigned>(x)); };
                            const long a1 = pack(fq[ks][0].x, fq[ks][0].y), a2 = pack(fq[ks][1].x, fq[ks][1].y);
                            const long bh = pack(hi.x, hi.y), bl = pack(lo.x, lo.y);
                            acc1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a1, bh, ks ? acc1 : zero16, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a1, bl, ks ? acc2 : zero16, 0, 0, 0);
                            if constexpr (!SKIP_LOW) acc2 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a2, bh, acc2, 0, 0, 0);
                        } else {
                            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][0], hi, ks ? acc1 : zero16, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][0], lo, ks ? acc2 : zero16, 0, 0, 0);
                            if constexpr (!SKIP_LOW) acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][1], hi, acc2, 0, 0, 0);
                        }
                    }
                } else if ((DBG & 4) && (rt & 1)) {
                    // diagnostic: every second wave multiplies register constants -- half the fragment reads and byte splits of
                    // a workgroup, all of its matrix work (what a wave holding 64 tap rows per data fragment would save)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][0], fq[ks][1], ks ? acc1 : zero16, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][0], fq[ks][0], ks ? acc2 : zero16, 0, 0, 0);
                        if constexpr (!SKIP_LOW) acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][1], fq[ks][1], acc2, 0, 0, 0);
                    }
                } else {
                v4i_t dd[KS][2];
#pragma unroll
                for (int ks = 0; ks < PD; ++ks) {
                    dd[ks][0] = *reinterpret_cast<const v4i_t *>(la + 64 * ks);
                    dd[ks][1] = *reinterpret_cast<const v4i_t *>(la + 64 * ks + 16);
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    // the other parity's tile boundary: no memory clobber -- this wave's own fragment reads may move across it
                    if (STAGGER && ks == KS / 2) asm volatile("s_barrier");
                    if (EMIT && ks == KS - 3 && store_here) ring_emit_store<true>(a, c, em, eg);  // (its reads went out before k step 0)
                    const v4i_t d0 = dd[ks][0], d1 = dd[ks][1];
                    v4i_t hi, lo;
                    if (DBG & 8) {  // diagnostic: no byte split
                        hi = d0;
                        lo = d1;
                    } else {
                    hi.x = __builtin_amdgcn_perm(d0.y, d0.x, 0x07050301);
                    hi.y = __builtin_amdgcn_perm(d0.w, d0.z, 0x07050301);
                    hi.z = __builtin_amdgcn_perm(d1.y, d1.x, 0x07050301);
                    hi.w = __builtin_amdgcn_perm(d1.w, d1.z, 0x07050301);
                    lo.x = __builtin_amdgcn_perm(d0.y, d0.x, 0x06040200) ^ 0x80808080;
                    lo.y = __builtin_amdgcn_perm(d0.w, d0.z, 0x06040200) ^ 0x80808080;
                    lo.z = __builtin_amdgcn_perm(d1.y, d1.x, 0x06040200) ^ 0x80808080;
                    lo.w = __builtin_amdgcn_perm(d1.w, d1.z, 0x06040200) ^ 0x80808080;
                    }
                    if (ks + PD < KS) {
                        dd[ks + PD][0] = *reinterpret_cast<const v4i_t *>(la + 64 * (ks + PD));
                        dd[ks + PD][1] = *reinterpret_cast<const v4i_t *>(la + 64 * (ks + PD) + 16);
                    }
                    if (DBG & 32) {
                        asm volatile("" ::"v"(hi), "v"(lo));
                    } else {
                        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][0], hi, ks ? acc1 : zero16, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][0], lo, ks ? acc2 : zero16, 0, 0, 0);
                        if constexpr (!SKIP_LOW) acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][1], hi, acc2, 0, 0, 0);
                    }
                    if (IQA_RING_SCHED_BARRIER) __builtin_amdgcn_sched_barrier(0);
                }
                }
                }
                if (DBG & 32) acc1 = acc2 = zero16;
                if (DBG & 1) {
                    asm volatile("" ::"v"(acc1), "v"(acc2));
                } else if (DEFER_ADDS) {
                    held1 = acc1;
                    held2 = acc2;
                    held_t = t;
                } else {
                    scatter(t, acc1, acc2);
                }
            };
            if constexpr (SKIP == 2 && !U8) {
                tile_body(std::true_type{});
            } else if constexpr (SKIP == 1 && !U8) {  // (a uniform branch: one lane's waves all take the same side)
                if (a.high_taps_only) tile_body(std::true_type{});
                else tile_body(std::false_type{});
            } else {
                tile_body(std::false_type{});
            }
        } else {
            if (STAGGER) asm volatile("s_barrier" ::: "memory");  // no tile this round (odd tile count): the mid-tile barrier alone
        }
        }
        if (store_pending) ring_emit_store<true>(a, c, em, eg);  // (no tile of its own this round)
        slot = (slot + 1 == R) ? 0 : slot + 1;
    }
    if (DEFER_ADDS && held_t >= 0) scatter(held_t, held1, held2);
    if (STAGGER && cp == 0) asm volatile("s_barrier" ::: "memory");
    if ((DBG & 2) && a.stamps != nullptr && c.lane == 0) {
        unsigned long long *o = a.stamps + (static_cast<size_t>(blockIdx.x) * RG_WAVES + (cp * 4 + rt)) * 4;
        o[0] = st_wait;
        o[1] = st_work + (__builtin_amdgcn_s_memtime() - st_prev);
        o[2] = static_cast<unsigned long long>(c.rounds);
    }
    // the last groups: everything has landed behind a full wait and one more barrier
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (EMIT) {
        const int k_last = (c.cnt + 62) >> 6;
        // the groups the loop has not emitted: PAIR emitted group k in round 2k + RG_EMIT_LAG_PAIR
        const int own = c.rounds - c.tshift;  // rounds in this lane's own count
        const int k_next = PAIR1 ? (own > RG_EMIT_LAG_PAIR ? ((own - 1 - RG_EMIT_LAG_PAIR) >> 1) + 1 : 0) : max(own - RG_EMIT_LAG, 0);
        if (!PAIR || c.tshift < RG_PAIR_IDLE)  // (the idle half of a pair without a second lane emits nothing)
            for (int k = k_next; k <= k_last; ++k) ring_emit_group<ACC64>(a, c, em, k);
    }
}

// One block = one contiguous range of outputs of any length (the host gives every CU one range): a persistent
// stream through the ring, sums in a 512-position sliding window, outputs emitted two rounds behind the matrix work.
// SKIPK (kernels of launches in which some lanes have high-byte-only taps, see ring_main's SKIP): lane pairs -- the pair's
// first lane (parity 0) skips the q2*hi product at compile time, the second never does (the host pairs a tap-row group's
// high-byte lane with its residue lane); one lane per workgroup -- every wave asks its lane's flag.
template <int KS, int DBG, bool ACC64, bool ROWS, bool U8, bool PAIR = false, bool SKIPK = false, bool HALF = false>
__device__ __forceinline__ void ring_block(const MfmaArgs &a, long long range_idx)
{
    using G = RingGeo<KS, ROWS, U8, PAIR>;
    constexpr int R = G::R, SLOT = G::SLOT;
    constexpr bool LOADERS = G::LOADERS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    RingCtx c;
    c.lane = tid & 63;
    c.rt = wave & 3;
    c.cp = wave >= RG_WAVES ? (wave - RG_WAVES) & 1 : (wave >> 2) & 1;  // (loader waves: their parity, or the lane of a pair they emit)
    c.col = c.lane & 31;
    c.h = c.lane >> 5;

    c.i0 = range_idx * a.range;
    c.cnt = static_cast<int>(min(static_cast<long long>(a.range), a.n_out - c.i0));
    c.m0 = a.m_lo + c.i0;
    c.tiles = (c.cnt + 63 + 31) >> 5;  // data columns b in [m0-64, m0+cnt-2], rounded up to tiles of 32
    c.rounds = G::PAIR1 ? c.tiles + a.pair_extra : ((c.tiles + 1) >> 1) + (PAIR ? a.pair_extra : 0);
    c.extra = PAIR ? a.pair_extra : 0;
    c.tshift = PAIR ? a.pair_shift : 0;  // (RG_PAIR_IDLE: the idle half of a pair without a second lane -- barriers only)
    c.smem = smem;
    c.s_acc = reinterpret_cast<int *>(smem + R * G::TPR * SLOT);
    for (int i = tid; i < G::ACCS * RG_ACC_BYTES / 4; i += G::THREADS) c.s_acc[i] = 0;
    if (PAIR) c.s_acc += c.cp * (RG_ACC_BYTES / 4);  // each lane of the pair sums into its own window

    // the stream: tile t starts at data row m0 - 64 - col_shift + 32 t, i.e. frame row*D + 1
    constexpr int FB = U8 ? 2 : 4;  // bytes per frame
    const long long row_bytes = static_cast<long long>(FB) * a.D;
    c.tile_bytes = 32 * row_bytes;
    const char *stream = reinterpret_cast<const char *>(a.raw) + FB * ((c.m0 - MF_Q - a.col_shift) * a.D + 1 - a.consumed) +
                         (ROWS ? G::KBYTES * a.k_first : 0);
    c.row_units = static_cast<int>(row_bytes >> 4);
    c.pitch_units = G::PADDED ? (c.row_units | 1) : c.row_units;  // odd: conflict-free fragment reads (see RingGeo)
    c.lane_off = G::SPLIT ? c.col * G::PLANE_PITCH + 16 * c.h : c.col * (ROWS ? G::PITCH : 16 * c.pitch_units) + (G::KBYTES / 2) * c.h;

    if constexpr (LOADERS) {
        if (wave >= RG_WAVES) {
            c.stream0 = stream;
            __syncthreads();
            if constexpr (G::SPLIT && PAIR && G::NLOADERS == 2) {
                if (wave - RG_WAVES) ring_loader_split<KS, DBG, ACC64, ROWS, 1, true>(a, c);
                else ring_loader_split<KS, DBG, ACC64, ROWS, 0, true>(a, c);
            } else if constexpr (G::SPLIT && PAIR) {
                switch (wave - RG_WAVES) {
                    case 0: ring_loader_split<KS, DBG, ACC64, ROWS, 0, true>(a, c); break;
                    case 1: ring_loader_split<KS, DBG, ACC64, ROWS, 1, true>(a, c); break;
                    case 2: ring_loader_split<KS, DBG, ACC64, ROWS, 2, true>(a, c); break;
                    default: ring_loader_split<KS, DBG, ACC64, ROWS, 3, true>(a, c); break;
                }
            } else if constexpr (G::SPLIT) {
                if ((wave - RG_WAVES) >> 1) ring_loader_split<KS, DBG, ACC64, ROWS, 1>(a, c);
                else ring_loader_split<KS, DBG, ACC64, ROWS, 0>(a, c);
            }
            else ring_loader<KS, DBG, ACC64, ROWS, U8>(a, c);
            return;
        }
    }
    // tap fragments of this wave's row tile: registers for the whole block
    v4i_t fq[KS][2];
    {
        const v4i_t *fa = a.afrag + c.rt * 128 + c.lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) fq[ks][pc] = fa[(ks * 8 + pc) * 64];
        // have them arrive here, before the first DMA: a later wait for them would drain the DMA queue with them
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(fq[ks][0]), "v"(fq[ks][1]));
    }
    // A row of 2 D values whose last k step holds at most 16 of them (D = 104: 208 = 6 x 32 + 16): the other half of that
    // step multiplied padding against zero taps.  The byte planes make the half step a v_mfma_i32_32x32x16_i8 -- lanes 0..31
    // take values 0..7 of the step, lanes 32..63 values 8..15: the tap fragment's first 8 bytes, and for the upper lanes the
    // second 8 bytes of the lane 32 below (whose own bytes were the padding's zero taps); the data are two ds_read_b64.
    // (The kernel variant HALF: the launcher picks it for rows of 32 (KS - 1) + 1 .. 16 values.)
    c.half_last = HALF;
    if constexpr (HALF) {
        static_assert(!ROWS && !U8 && !PAIR, "the half last k step exists for single lanes over contiguous byte-plane slots");
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            const v4i_t f = fq[KS - 1][pc];
            const int z0 = __shfl(f.z, c.lane ^ 32, kWave), w0 = __shfl(f.w, c.lane ^ 32, kWave);
            if (c.h) {
                fq[KS - 1][pc].x = z0;
                fq[KS - 1][pc].y = w0;
            }
        }
    }
    __syncthreads();
    c.stream0 = stream;
    if constexpr (PAIR && G::SPLIT) {
        // loader waves feed the ring and emit: every multiplying wave only multiplies (parity 1 defers its adds)
        constexpr int SA = SKIPK ? 2 : 0, SB = 0;  // (parity 0 = the pair's first lane)
        if (c.cp) ring_main<KS, DBG, ACC64, ROWS, U8, false, false, true, true>(a, c, fq, std::integral_constant<int, SB>{});
        else ring_main<KS, DBG, ACC64, ROWS, U8, false, false, false, true>(a, c, fq, std::integral_constant<int, SA>{});
    } else if constexpr (PAIR) {
        // parity 0's first two waves feed the ring (one tile per round: the same KS + 1 instructions per issuing wave and
        // round as without pairs); one wave of either parity emits ITS lane's outputs; parity 1 defers its adds
        // (waves go to SIMDs cyclically: issuers on SIMDs 0 and 1, lane A's emitter -- wave 2 -- on SIMD 2, lane B's -- wave
        // 7 -- on SIMD 3)
        constexpr int SA = SKIPK ? 2 : 0, SB = 0;  // (parity 0 = the pair's first lane)
        if constexpr (G::PAIR2) {
            // the single-lane kernel's roles: one issuing wave per SIMD (rt 0, 1 of parity 0 feed the round's first tile, rt 2, 3
            // of parity 1 its second), one emitting wave per LANE (wave 2 = lane A's rt 2 on SIMD 2, wave 5 = lane B's rt 1 on
            // SIMD 1: neither issues); no deferred adds (two tiles per round keep a SIMD's two waves out of step by themselves)
            if ((c.rt >> 1) == c.cp) {
                if (c.cp) ring_main<KS, DBG, ACC64, ROWS, U8, true, false, false, true>(a, c, fq, std::integral_constant<int, SB>{});
                else ring_main<KS, DBG, ACC64, ROWS, U8, true, false, false, true>(a, c, fq, std::integral_constant<int, SA>{});
            } else if (c.cp == 0 && c.rt == 2) ring_main<KS, DBG, ACC64, ROWS, U8, false, true, false, true>(a, c, fq, std::integral_constant<int, SA>{});
            else if (c.cp == 1 && c.rt == 1) ring_main<KS, DBG, ACC64, ROWS, U8, false, true, false, true>(a, c, fq, std::integral_constant<int, SB>{});
            else if (c.cp) ring_main<KS, DBG, ACC64, ROWS, U8, false, false, false, true>(a, c, fq, std::integral_constant<int, SB>{});
            else ring_main<KS, DBG, ACC64, ROWS, U8, false, false, false, true>(a, c, fq, std::integral_constant<int, SA>{});
        } else
        if (c.cp == 0 && c.rt < 2) ring_main<KS, DBG, ACC64, ROWS, U8, true, false, false, true>(a, c, fq, std::integral_constant<int, SA>{});
        else if (c.cp == 0 && c.rt == 2) ring_main<KS, DBG, ACC64, ROWS, U8, false, true, false, true>(a, c, fq, std::integral_constant<int, SA>{});
        else if (c.cp == 1 && c.rt == 3) ring_main<KS, DBG, ACC64, ROWS, U8, false, true, false, true>(a, c, fq, std::integral_constant<int, SB>{});
        else if (c.cp) ring_main<KS, DBG, ACC64, ROWS, U8, false, false, true, true>(a, c, fq, std::integral_constant<int, SB>{});
        else ring_main<KS, DBG, ACC64, ROWS, U8, false, false, false, true>(a, c, fq, std::integral_constant<int, SA>{});
    } else if constexpr (LOADERS) {
        constexpr int S1 = SKIPK ? 1 : 0;
        if (c.cp) ring_main<KS, DBG, ACC64, ROWS, U8, false, false, true, false, std::integral_constant<int, S1>, HALF>(a, c, fq, std::integral_constant<int, S1>{});
        else ring_main<KS, DBG, ACC64, ROWS, U8, false, false, false, false, std::integral_constant<int, S1>, HALF>(a, c, fq, std::integral_constant<int, S1>{});
    } else {
        constexpr int S1 = SKIPK ? 1 : 0;
        // one issuing wave per SIMD (waves go to SIMDs in a cyclic order of period 4): rt 0,1 of parity 0, rt 2,3 of parity 1
        if ((c.rt >> 1) == c.cp) {
            if (c.cp) ring_main<KS, DBG, ACC64, ROWS, U8, true, false, true, false>(a, c, fq, std::integral_constant<int, S1>{});
            else ring_main<KS, DBG, ACC64, ROWS, U8, true, false, false, false>(a, c, fq, std::integral_constant<int, S1>{});
        } else if (wave == RG_EMIT_WAVE) ring_main<KS, DBG, ACC64, ROWS, U8, false, true, false, false>(a, c, fq, std::integral_constant<int, S1>{});
        else if (c.cp) ring_main<KS, DBG, ACC64, ROWS, U8, false, false, true, false>(a, c, fq, std::integral_constant<int, S1>{});
        else ring_main<KS, DBG, ACC64, ROWS, U8, false, false, false, false>(a, c, fq, std::integral_constant<int, S1>{});
    }
}

template <int KS, int DBG, bool ACC64>
__global__ __launch_bounds__((RingGeo<KS, false>::THREADS), (RingGeo<KS, false>::LOADERS ? 3 : 2)) void k_channelize_mfma_s16_ring(MfmaArgs a)
{
    ring_block<KS, DBG, ACC64, false, false>(a, blockIdx.x);
}

// The variant whose last k step is a 32x32x16 MFMA (rows of 32 (KS - 1) + 1 .. 16 values: D = 104 -> 208 = 6 x 32 + 16).
template <int KS>
__global__ __launch_bounds__((RingGeo<KS, false>::THREADS), 3) void k_channelize_mfma_s16_ring_half(MfmaArgs a)
{
    static_assert(RingGeo<KS, false>::SPLIT, "byte-plane kernels only");
    ring_block<KS, 0, false, false, false, false, false, true>(a, blockIdx.x);
}

// The same block under its own name for short launches (the mixer-sign probes: a few thousand outputs in blocks of
// 64), so that profiles keep the capture-long launches and the probes in separate rows.
template <int KS>
__global__ __launch_bounds__((RingGeo<KS, false>::THREADS), (RingGeo<KS, false>::LOADERS ? 3 : 2)) void k_channelize_mfma_s16_ring_short(MfmaArgs a)
{
    ring_block<KS, 0, false, false, false>(a, blockIdx.x);
}

// Row-staged slots (any D, one k-step range per pass), int32 sums.
template <int KS>
__global__ __launch_bounds__((RingGeo<KS, true>::THREADS), 3) void k_channelize_mfma_s16_ring_rows(MfmaArgs a)
{
    ring_block<KS, 0, false, true, false>(a, blockIdx.x);
}

// Row-staged slots, uint8 I/Q captures (cu8 / RTL-SDR), int32 sums.
template <int KS>
__global__ __launch_bounds__((RingGeo<KS, true, true>::THREADS), 3) void k_channelize_mfma_u8_ring_rows(MfmaArgs a)
{
    ring_block<KS, 0, false, true, true>(a, blockIdx.x);
}


// ---- several channels of ONE capture in one launch (shared ingest) ---------------------------------------------
//
// A lane = one (channel, tap-row group): its own tap fragments, output, scale, rotation.  All lanes of a launch share
// the capture, the decimation, the k-step range and the output range [m_lo, m_lo + n_out).  The reference runs a whole
// pipeline per --ft target over the same file (cli.py:683-710); here the lanes of one stretch of the capture run AT THE
// SAME TIME on the CUs of ONE XCD, so that stretch crosses the fabric once and the other lanes' LDS-DMAs hit in that
// XCD's L2: workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one -- a speed assumption only, nothing
// here depends on it for correctness), so workgroup b takes lane (b / 8) % n_lanes of output range
// ((b / 8) / n_lanes) * 8 + b % 8.  Every lane does the same work per tile, so the lanes of a range stay within a few
// rounds of each other without any synchronisation (4 MiB of L2 per XCD = dozens of rounds of slack).
struct RingLane {
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

constexpr int RG_MAX_LANES = 16;  // (<= 5 targets x <= 3 tap-row groups in the reference's CLI; the table travels as kernel arguments)

struct RingMultiArgs {
    MfmaArgs c;  // what the lanes share; the per-lane fields of `c` are overwritten per workgroup
    int n_lanes;
    RingLane lane[RG_MAX_LANES];
};

template <int KS, bool ROWS, bool U8, bool PAIR = false, bool ACC64 = false, bool SKIPK = false, bool HALF = false>
__device__ __forceinline__ void ring_multi_block(const RingMultiArgs &m)
{
    const int idx = blockIdx.x >> 3;
    // uniform: scalar loads from the kernel-argument segment.  PAIR: the table holds the pairs back to back, a workgroup
    // takes pair (idx mod n_pairs) and its waves of parity cp (waves 4..7: cp = 1) the pair's lane cp
    const int units = PAIR ? m.n_lanes >> 1 : m.n_lanes;
    const int wv = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    // (loader waves 8, 10 carry the first lane's arguments, 9, 11 the second's: waves 8 and 9 emit those lanes)
    const int li = PAIR ? 2 * (idx % units) + (wv >= RG_WAVES ? wv & 1 : (wv >> 2) & 1) : idx % units;
    const long long range_idx = static_cast<long long>(idx / units) * 8 + (blockIdx.x & 7);
    const RingLane &l = m.lane[li];
    MfmaArgs a = m.c;
    a.pair_shift = a.pair_extra = 0;
    a.afrag = l.afrag;
    a.out = l.out;
    a.partial_in = l.partial_in;
    a.partial_out = l.partial_out;
    a.unit = l.unit;
    a.c_re = l.c_re;
    a.c_im = l.c_im;
    a.rot_step = l.rot_step;
    a.rot_base = l.rot_base;
    a.rot64_re = l.rot64_re;
    a.rot64_im = l.rot64_im;
    a.sc_re = l.sc_re;
    a.sc_im = l.sc_im;
    a.col_shift = l.col_shift;
    a.finalize = l.finalize;
    a.conj_sum = l.conj_sum;
    a.rotate = l.rotate;
    a.raw_partials = l.raw_partials;
    a.high_taps_only = l.high_taps_only;
    if constexpr (PAIR) {
        // the pair's first lane has the larger (or the same) tap-row group: ITS stream is staged, the second lane's own
        // tiles arrive 2 rounds per group of difference later; a pair without a second lane (afrag NULL) idles that half
        const RingLane &la = m.lane[li & ~1], &lb = m.lane[li | 1];
        const bool idle_b = lb.afrag == nullptr;
        // (rounds the second lane works behind the first: two tiles per tap-row group of difference = one round of two tiles)
        a.pair_extra = idle_b ? 0 : (la.col_shift - lb.col_shift) >> (RingGeo<KS, ROWS, U8, PAIR>::PAIR2 ? 6 : 5);
        a.col_shift = la.col_shift;
        if (li & 1) {
            a.pair_shift = idle_b ? RG_PAIR_IDLE : a.pair_extra;
            if (idle_b) a.afrag = la.afrag;  // (anything readable: the fragments are loaded, never used)
        }
        a.pace_units = units;
        a.pace_slot = static_cast<int>(range_idx) * units + idx % units;
    }
    if (range_idx * a.range >= a.n_out) return;  // (the last ranges of a short launch)
    ring_block<KS, 0, ACC64, ROWS, U8, PAIR, SKIPK, HALF>(a, range_idx);
}

// ACC64: one int64 (S1 << 32) + S2 per output component instead of one int32 256*S1 + S2 -- 16-bit taps without the int32
// bound, what the "full" precision's lanes (taps + their residue, dsp_plan.plan_mfma(residual=True)) need; contiguous slots only.
template <int KS, bool ACC64 = false, bool SKIPK = false>
__global__ __launch_bounds__((RingGeo<KS, false>::THREADS), (RingGeo<KS, false>::LOADERS ? 3 : 2)) void k_channelize_mfma_s16_ring_multi(RingMultiArgs m)
{
    ring_multi_block<KS, false, false, false, ACC64, SKIPK>(m);
}

// ... with the row's last k step as a 32x32x16 MFMA (see k_channelize_mfma_s16_ring_half).
template <int KS>
__global__ __launch_bounds__((RingGeo<KS, false>::THREADS), 3) void k_channelize_mfma_s16_ring_multi_half(RingMultiArgs m)
{
    static_assert(RingGeo<KS, false>::SPLIT, "byte-plane kernels only");
    ring_multi_block<KS, false, false, false, false, false, true>(m);
}

// Two lanes per workgroup (RingGeo PAIR): the lanes of the table in pairs of equal tap-row group.
template <int KS, bool ACC64 = false, bool SKIPK = false>
__global__ __launch_bounds__((RingGeo<KS, false, false, true>::THREADS), (RingGeo<KS, false, false, true>::LOADERS ? 3 : 2)) void k_channelize_mfma_s16_ring_pairs(RingMultiArgs m)
{
    ring_multi_block<KS, false, false, true, ACC64, SKIPK>(m);
}

template <int KS, bool SKIPK = false>
__global__ __launch_bounds__((RingGeo<KS, true>::THREADS), 3) void k_channelize_mfma_s16_ring_rows_multi(RingMultiArgs m)
{
    ring_multi_block<KS, true, false, false, false, SKIPK>(m);
}

template <int KS>
__global__ __launch_bounds__((RingGeo<KS, true, true>::THREADS), 3) void k_channelize_mfma_u8_ring_rows_multi(RingMultiArgs m)
{
    ring_multi_block<KS, true, true>(m);
}

// The 160 KiB dynamic-LDS limit is a per-device attribute of a kernel: `done` remembers (one bit per device id)
// where it has been raised.  Two threads that race here both set it -- harmless, the call is idempotent.
template <typename K, typename A>
static int ring_launch_kernel(K kernel, const char *name, int threads, const A &a, unsigned blocks, size_t lds, hipStream_t stream,
                              std::atomic<unsigned long long> &done)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            set_error("%s: cannot raise the dynamic LDS limit on device %d: %s", name, dev, hipGetErrorString(e));
            return IQA_EHIP;
        }
        done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds, stream, a);
    return check_launch(name);
}

template <int KS, int DBG, bool ACC64>
static int ring_launch_one(const MfmaArgs &a, unsigned blocks, size_t lds, hipStream_t stream)
{
    static std::atomic<unsigned long long> done{0};
    return ring_launch_kernel(k_channelize_mfma_s16_ring<KS, DBG, ACC64>, "k_channelize_mfma_s16_ring", RingGeo<KS, false>::THREADS, a, blocks, lds, stream, done);
}

template <int KS>
static int ring_launch_half(const MfmaArgs &a, unsigned blocks, size_t lds, hipStream_t stream)
{
    static std::atomic<unsigned long long> done{0};
    return ring_launch_kernel(k_channelize_mfma_s16_ring_half<KS>, "k_channelize_mfma_s16_ring", RingGeo<KS, false>::THREADS, a, blocks, lds, stream, done);
}

template <int KS>
static int ring_launch_short(const MfmaArgs &a, unsigned blocks, size_t lds, hipStream_t stream)
{
    static std::atomic<unsigned long long> done{0};
    return ring_launch_kernel(k_channelize_mfma_s16_ring_short<KS>, "k_channelize_mfma_s16_ring_short", RingGeo<KS, false>::THREADS, a, blocks, lds, stream, done);
}

template <int KS>
static int ring_launch_rows(const MfmaArgs &a, unsigned blocks, size_t lds, hipStream_t stream)
{
    static std::atomic<unsigned long long> done{0};
    return ring_launch_kernel(k_channelize_mfma_s16_ring_rows<KS>, "k_channelize_mfma_s16_ring_rows", RingGeo<KS, true>::THREADS, a, blocks, lds, stream, done);
}

template <int KS>
static int ring_launch_rows_u8(const MfmaArgs &a, unsigned blocks, size_t lds, hipStream_t stream)
{
    static std::atomic<unsigned long long> done{0};
    return ring_launch_kernel(k_channelize_mfma_u8_ring_rows<KS>, "k_channelize_mfma_u8_ring_rows", RingGeo<KS, true, true>::THREADS, a, blocks, lds, stream, done);
}

template <int KS>
static int ring_launch_multi(const RingMultiArgs &m, unsigned blocks, size_t lds, hipStream_t stream, bool rows, bool u8, bool acc64, bool skipk)
{
    static std::atomic<unsigned long long> done[8] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};
    if (acc64) {
        if constexpr (KS < RG_MAX_KS) {  // (64-bit sums at 16 k steps do not fit the registers)
            if (!rows && !u8) {
                if (skipk)
                    return ring_launch_kernel(k_channelize_mfma_s16_ring_multi<KS, true, true>, "k_channelize_mfma_s16_ring_multi64", RingGeo<KS, false>::THREADS, m, blocks, lds, stream, done[4]);
                return ring_launch_kernel(k_channelize_mfma_s16_ring_multi<KS, true>, "k_channelize_mfma_s16_ring_multi64", RingGeo<KS, false>::THREADS, m, blocks, lds, stream, done[3]);
            }
        }
        set_error("multi-lane launches with 64-bit sums: contiguous int16 slots, at most %d k steps (got %d)", RG_MAX_KS - 1, KS);
        return IQA_EINVAL;
    }
    if constexpr (KS <= RG_ROWS_MAX_KS_C) {
        if (u8) return ring_launch_kernel(k_channelize_mfma_u8_ring_rows_multi<KS>, "k_channelize_mfma_u8_ring_rows_multi", RingGeo<KS, true, true>::THREADS, m, blocks, lds, stream, done[2]);
        if (rows && skipk) return ring_launch_kernel(k_channelize_mfma_s16_ring_rows_multi<KS, true>, "k_channelize_mfma_s16_ring_rows_multi", RingGeo<KS, true>::THREADS, m, blocks, lds, stream, done[5]);
        if (rows) return ring_launch_kernel(k_channelize_mfma_s16_ring_rows_multi<KS>, "k_channelize_mfma_s16_ring_rows_multi", RingGeo<KS, true>::THREADS, m, blocks, lds, stream, done[1]);
    } else {
        if (u8 || rows) {
            set_error("row-staged ring kernels take at most %d k steps per pass (got %d)", RG_ROWS_MAX_KS_C, KS);
            return IQA_EINVAL;
        }
    }
    if (skipk) return ring_launch_kernel(k_channelize_mfma_s16_ring_multi<KS, false, true>, "k_channelize_mfma_s16_ring_multi", RingGeo<KS, false>::THREADS, m, blocks, lds, stream, done[6]);
    if constexpr (KS <= 8 && IQA_RING_HALF_STEP != 0 && IQA_RING_SPLIT_STAGE != 0) {
        const int rem = (2 * m.c.D) & 31;  // values in the row's last k step
        if (rem != 0 && rem <= 16)
            return ring_launch_kernel(k_channelize_mfma_s16_ring_multi_half<KS>, "k_channelize_mfma_s16_ring_multi", RingGeo<KS, false>::THREADS, m, blocks, lds, stream, done[7]);
    }
    return ring_launch_kernel(k_channelize_mfma_s16_ring_multi<KS>, "k_channelize_mfma_s16_ring_multi", RingGeo<KS, false>::THREADS, m, blocks, lds, stream, done[0]);
}

constexpr int RG_PAIR_MIN_KS = 9;  // lane pairs from 9 k steps on (below, a range's single-lane workgroups share the capture through L2 by themselves)

template <int KS>
static int ring_launch_pairs(const RingMultiArgs &m, unsigned blocks, hipStream_t stream, bool acc64, bool skipk)
{
    static std::atomic<unsigned long long> done{0}, done64{0}, done_s{0}, done64_s{0};
    if constexpr (KS >= RG_PAIR_MIN_KS) {
        using G = RingGeo<KS, false, false, true>;
        if (acc64) {
            if constexpr (KS <= 14) {
                if (skipk)
                    return ring_launch_kernel(k_channelize_mfma_s16_ring_pairs<KS, true, true>, "k_channelize_mfma_s16_ring_pairs64", G::THREADS, m, blocks, G::LDS_BYTES, stream, done64_s);
                return ring_launch_kernel(k_channelize_mfma_s16_ring_pairs<KS, true>, "k_channelize_mfma_s16_ring_pairs64", G::THREADS, m, blocks, G::LDS_BYTES, stream, done64);
            }
            set_error("lane pairs with 64-bit sums: 9..14 k steps (got %d)", KS);
            return IQA_EINVAL;
        }
        if (skipk) return ring_launch_kernel(k_channelize_mfma_s16_ring_pairs<KS, false, true>, "k_channelize_mfma_s16_ring_pairs", G::THREADS, m, blocks, G::LDS_BYTES, stream, done_s);
        return ring_launch_kernel(k_channelize_mfma_s16_ring_pairs<KS>, "k_channelize_mfma_s16_ring_pairs", G::THREADS, m, blocks, G::LDS_BYTES, stream, done);
    } else {
        set_error("lane pairs need at least %d k steps (got %d)", RG_PAIR_MIN_KS, KS);
        return IQA_EINVAL;
    }
}

// The pacing words of the lane-pair launches: RG_PACE_WORDS x 4 bytes per device, library state (see the header's
// conventions), never freed.  Allocated and cleared -- synchronously -- at the first pair launch on a device that is
// NOT inside a stream capture (an allocation would invalidate the capture: such a launch runs unpaced, which is correct,
// only its workgroups may drift apart); every launch takes its own slice of the buffer, handed out round-robin, so two
// pair launches in flight on one device do not share words, and entries carry the launch's token, so nothing is reset
// between launches.
static unsigned int *ring_pace_buffer(unsigned int &token, int words, hipStream_t stream)
{
    static std::mutex mu;
    static unsigned int *buf[64] = {};
    static int next_off[64] = {};
    static unsigned int next_token = 1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    token = next_token = (next_token % 0xFFFFFu) + 1;
    unsigned int *&b = buf[dev & 63];
    if (b == nullptr) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return nullptr;
        }
        void *p = nullptr;
        if (hipMalloc(&p, RG_PACE_WORDS * sizeof(unsigned int)) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        if (hipMemset(p, 0, RG_PACE_WORDS * sizeof(unsigned int)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(p);
            return nullptr;
        }
        b = static_cast<unsigned int *>(p);
    }
    if (words > RG_PACE_WORDS) return nullptr;
    int &off = next_off[dev & 63];
    if (off + words > RG_PACE_WORDS) off = 0;
    unsigned int *mine = b + off;
    off += (words + 63) & ~63;
    return mine;
}

bool mfma_ring_pairs_supported(int decimation, int k_first, int k_count, bool u8, bool acc64)
{
    // (15 k steps: the pair kernel would need 20 registers more than a wave has -- a spill's scratch loads would join
    // the counted vmcnt sequence of the issuing waves; 64-bit sums cost ~20 registers more: up to 14 k steps)
    return mfma_ring_mode(decimation, k_first, k_count, acc64, u8) == 1 && k_count >= RG_PAIR_MIN_KS && k_count != 15 && (!acc64 || k_count <= 14);
}

constexpr int RG_ROWS_MAX_KS = RG_ROWS_MAX_KS_C;  // 8*KS tap registers + the rest must stay within 168 (three waves on two SIMDs)

// 0: the ring kernels do not apply; 1: contiguous slots; 2: row-staged slots (int32 sums only)
int mfma_ring_mode(int decimation, int k_first, int k_count, bool acc64, bool u8)
{
    const int ks_all = (2 * decimation + 31) / 32;
    // (64-bit sums at 16 k steps do not fit 256 registers: a spill's scratch loads would join the counted vmcnt sequence)
    if (!u8 && decimation >= 4 && (decimation & 3) == 0 && ks_all <= (acc64 ? RG_MAX_KS - 1 : RG_MAX_KS) && k_first == 0 && k_count == ks_all) return 1;
    if (!acc64 && decimation >= 1 && k_count >= 1 && k_count <= RG_ROWS_MAX_KS && k_first >= 0 && k_first + k_count <= ks_all) return 2;
    return 0;
}

bool mfma_ring_supported(int decimation)
{
    const int ks = (2 * decimation + 31) / 32;
    return decimation >= 4 && (decimation & 3) == 0 && ks <= RG_MAX_KS;
}

template <int KS>
static constexpr size_t ring_bytes_of(bool rows, bool u8)
{
    return u8     ? static_cast<size_t>(RingGeo<KS, true, true>::R) * 2 * RingGeo<KS, true, true>::SLOT + RG_ACC_BYTES
           : rows ? static_cast<size_t>(RingGeo<KS, true>::R) * 2 * RingGeo<KS, true>::SLOT + RG_ACC_BYTES
                  : static_cast<size_t>(RingGeo<KS, false>::R) * 2 * RingGeo<KS, false>::SLOT + RG_ACC_BYTES;
}

size_t mfma_ring_lds_bytes(int ksteps, bool rows, bool u8)
{
    switch (ksteps) {
#define RG_B(K) case K: return (rows && K > RG_ROWS_MAX_KS) ? 0 : ring_bytes_of<(K <= 16 ? K : 16)>(rows, u8)
        RG_B(1); RG_B(2); RG_B(3); RG_B(4); RG_B(5); RG_B(6); RG_B(7); RG_B(8);
        RG_B(9); RG_B(10); RG_B(11); RG_B(12); RG_B(13); RG_B(14); RG_B(15); RG_B(16);
#undef RG_B
        default: return 0;
    }
}

// debug bit 7 (128) selects the 32-bit sums (needs fragments from dsp_plan.plan_mfma(acc32=True))
int mfma_ring_launch(const MfmaArgs &a, unsigned blocks, size_t lds, hipStream_t stream, bool rows, bool u8)
{
    const int dbg = a.debug & (1 | 2 | 4 | 8 | 16 | 32);
    const bool acc64 = !(a.debug & 128);
    if (u8) {
        switch (a.ksteps) {
#define RG_ROWS_U8(K) case K: return ring_launch_rows_u8<K>(a, blocks, lds, stream)
            RG_ROWS_U8(1); RG_ROWS_U8(2); RG_ROWS_U8(3); RG_ROWS_U8(4); RG_ROWS_U8(5); RG_ROWS_U8(6); RG_ROWS_U8(7); RG_ROWS_U8(8);
            RG_ROWS_U8(9); RG_ROWS_U8(10); RG_ROWS_U8(11);
#undef RG_ROWS_U8
            default: break;
        }
        set_error("uint8 ring kernel: %d k steps per pass not instantiated (1..%d)", a.ksteps, RG_ROWS_MAX_KS);
        return IQA_EINVAL;
    }
    if (rows) {
        switch (a.ksteps) {
#define RG_ROWS(K) case K: return ring_launch_rows<K>(a, blocks, lds, stream)
            RG_ROWS(1); RG_ROWS(2); RG_ROWS(3); RG_ROWS(4); RG_ROWS(5); RG_ROWS(6); RG_ROWS(7); RG_ROWS(8);
            RG_ROWS(9); RG_ROWS(10); RG_ROWS(11);
#undef RG_ROWS
            default: break;
        }
        set_error("row-staged ring kernel: %d k steps per pass not instantiated (1..%d)", a.ksteps, RG_ROWS_MAX_KS);
        return IQA_EINVAL;
    }
    if (dbg && a.ksteps == 7 && !acc64) {  // diagnostic instantiations exist for the two benchmark shapes only
        switch (dbg) {
            case 1: return ring_launch_one<7, 1, false>(a, blocks, lds, stream);
            case 16: return ring_launch_one<7, 16, false>(a, blocks, lds, stream);
            case 17: return ring_launch_one<7, 17, false>(a, blocks, lds, stream);
            case 32: return ring_launch_one<7, 32, false>(a, blocks, lds, stream);
            case 33: return ring_launch_one<7, 33, false>(a, blocks, lds, stream);
            case 4: return ring_launch_one<7, 4, false>(a, blocks, lds, stream);
            case 8: return ring_launch_one<7, 8, false>(a, blocks, lds, stream);
            case 5: return ring_launch_one<7, 5, false>(a, blocks, lds, stream);
            case 13: return ring_launch_one<7, 13, false>(a, blocks, lds, stream);
            default: break;
        }
    }
    if (dbg && a.ksteps == 13 && !acc64) {  // (D = 208: BASELINE configs 3 and 4, the kernel without loader waves)
        switch (dbg) {
            case 1: return ring_launch_one<13, 1, false>(a, blocks, lds, stream);
            case 16: return ring_launch_one<13, 16, false>(a, blocks, lds, stream);
            case 17: return ring_launch_one<13, 17, false>(a, blocks, lds, stream);
            case 32: return ring_launch_one<13, 32, false>(a, blocks, lds, stream);
            case 33: return ring_launch_one<13, 33, false>(a, blocks, lds, stream);
            case 4: return ring_launch_one<13, 4, false>(a, blocks, lds, stream);
            case 8: return ring_launch_one<13, 8, false>(a, blocks, lds, stream);
            case 2: return ring_launch_one<13, 2, false>(a, blocks, lds, stream);    // per-wave barrier-wait / work cycles
            case 18: return ring_launch_one<13, 18, false>(a, blocks, lds, stream);  // the same without the DMA stream
            default: break;
        }
    }
    if (IQA_RING_HALF_STEP != 0 && IQA_RING_SPLIT_STAGE != 0 && !dbg && !acc64 && a.range >= 512 && a.ksteps <= 8 && !a.high_taps_only) {
        const int rem = (2 * a.D) & 31;  // values in the row's last k step
        if (rem != 0 && rem <= 16) {
            switch (a.ksteps) {
#define RG_HALF(K) case K: return ring_launch_half<K>(a, blocks, lds, stream)
                RG_HALF(1); RG_HALF(2); RG_HALF(3); RG_HALF(4); RG_HALF(5); RG_HALF(6); RG_HALF(7); RG_HALF(8);
#undef RG_HALF
                default: break;
            }
        }
    }
    if (!dbg && !acc64 && a.range < 512) {  // short launch (int32 sums): same code, its own kernel name
        switch (a.ksteps) {
#define RG_SHORT(K) case K: return ring_launch_short<K>(a, blocks, lds, stream)
            RG_SHORT(1); RG_SHORT(2); RG_SHORT(3); RG_SHORT(4); RG_SHORT(5); RG_SHORT(6); RG_SHORT(7); RG_SHORT(8);
            RG_SHORT(9); RG_SHORT(10); RG_SHORT(11); RG_SHORT(12); RG_SHORT(13); RG_SHORT(14); RG_SHORT(15); RG_SHORT(16);
#undef RG_SHORT
            default: break;
        }
    }
    switch (a.ksteps) {
#define RG_CASE(K) \
    case K: return acc64 ? ring_launch_one<K, 0, true>(a, blocks, lds, stream) : ring_launch_one<K, 0, false>(a, blocks, lds, stream)
        RG_CASE(1); RG_CASE(2); RG_CASE(3); RG_CASE(4); RG_CASE(5); RG_CASE(6); RG_CASE(7); RG_CASE(8);
        RG_CASE(9); RG_CASE(10); RG_CASE(11); RG_CASE(12); RG_CASE(13); RG_CASE(14); RG_CASE(15); RG_CASE(16);
#undef RG_CASE
        default: break;
    }
    set_error("ring kernel: %d k steps not instantiated (1..%d)", a.ksteps, RG_MAX_KS);
    return IQA_EINVAL;
}

// Several lanes (channels x tap-row groups) of one capture in one launch; int32 sums only.  `lanes` holds n_lanes
// entries whose fields mirror the per-lane part of MfmaArgs; `a` carries what they share.
int mfma_ring_launch_multi(const MfmaArgs &a, const MfmaLane *lanes, int n_lanes, size_t lds, hipStream_t stream, bool rows, bool u8,
                           unsigned *blocks_out, bool pairs, bool acc64)
{
    if (n_lanes < 1 || n_lanes > RG_MAX_LANES) {
        set_error("a multi-lane launch takes 1..%d lanes (got %d)", RG_MAX_LANES, n_lanes);
        return IQA_EINVAL;
    }
    if (pairs) {
        if (rows || u8 || (n_lanes & 1)) {
            set_error("lane pairs: contiguous int16 slots and an even number of lanes");
            return IQA_EINVAL;
        }
        for (int i = 0; i < n_lanes; i += 2)
            if (lanes[i].afrag == nullptr || (lanes[i + 1].afrag != nullptr && lanes[i].col_shift < lanes[i + 1].col_shift)) {
                set_error("lane pairs: lane %d must exist and have the larger (or the same) tap-row group of its pair", i);
                return IQA_EINVAL;
            }
    }
    RingMultiArgs m;
    m.c = a;
    m.n_lanes = n_lanes;
    for (int i = 0; i < n_lanes; ++i) {
        RingLane &l = m.lane[i];
        const MfmaLane &s = lanes[i];
        l.afrag = s.afrag;
        l.out = s.out;
        l.partial_in = s.partial_in;
        l.partial_out = s.partial_out;
        l.unit = s.unit;
        l.c_re = s.c_re;
        l.c_im = s.c_im;
        l.rot_step = s.rot_step;
        l.rot_base = s.rot_base;
        l.rot64_re = s.rot64_re;
        l.rot64_im = s.rot64_im;
        l.sc_re = s.sc_re;
        l.sc_im = s.sc_im;
        l.col_shift = s.col_shift;
        l.finalize = s.finalize;
        l.conj_sum = s.conj_sum;
        l.rotate = s.rotate;
        l.raw_partials = s.raw_partials;
        l.high_taps_only = s.high_taps_only;
    }
    for (int i = n_lanes; i < RG_MAX_LANES; ++i) m.lane[i] = m.lane[0];
    // Which lanes may skip the q2*hi product.  Pairs: the kernel variant skips it for every pair's FIRST lane at compile time,
    // so it is taken only when every first lane is high-byte-only and no second lane is (what the host's pairing of a
    // tap-row group's two lanes gives); otherwise nothing is skipped.  One lane per workgroup: the variant that asks the
    // lane's flag, when any lane has it.
    bool skipk = false;
    if (!u8) {
        if (pairs) {
            skipk = true;
            for (int i = 0; i < n_lanes; i += 2)
                if (!lanes[i].high_taps_only || (lanes[i + 1].afrag != nullptr && lanes[i + 1].high_taps_only)) skipk = false;
        } else {
            for (int i = 0; i < n_lanes; ++i) skipk = skipk || lanes[i].high_taps_only != 0;
        }
    }
    if (!skipk)
        for (int i = 0; i < RG_MAX_LANES; ++i) m.lane[i].high_taps_only = 0;
    const long long ranges = (a.n_out + a.range - 1) / a.range;
    const long long groups = (ranges + 7) / 8;  // ranges are dealt to the 8 XCD classes: workgroup b -> class b % 8
    const unsigned blocks = static_cast<unsigned>(groups * (pairs ? n_lanes / 2 : n_lanes) * 8);
    if (blocks_out) *blocks_out = blocks;
    if (pairs) {
        m.c.pace = nullptr;
        m.c.pace = ring_pace_buffer(m.c.pace_token, static_cast<int>(std::min<long long>(groups * 8 * (n_lanes / 2), RG_PACE_WORDS + 1)), stream);
        switch (a.ksteps) {
#define RG_PAIRS(K) case K: return ring_launch_pairs<K>(m, blocks, stream, acc64, skipk)
            RG_PAIRS(9); RG_PAIRS(10); RG_PAIRS(11); RG_PAIRS(12); RG_PAIRS(13); RG_PAIRS(14); RG_PAIRS(16);
#undef RG_PAIRS
            default: break;
        }
        set_error("lane pairs: %d k steps not instantiated (9..14, 16)", a.ksteps);
        return IQA_EINVAL;
    }
    switch (a.ksteps) {
#define RG_MULTI(K) case K: return ring_launch_multi<K>(m, blocks, lds, stream, rows, u8, acc64, skipk)
        RG_MULTI(1); RG_MULTI(2); RG_MULTI(3); RG_MULTI(4); RG_MULTI(5); RG_MULTI(6); RG_MULTI(7); RG_MULTI(8);
        RG_MULTI(9); RG_MULTI(10); RG_MULTI(11); RG_MULTI(12); RG_MULTI(13); RG_MULTI(14); RG_MULTI(15); RG_MULTI(16);
#undef RG_MULTI
        default: break;
    }
    set_error("ring kernel: %d k steps not instantiated (1..%d)", a.ksteps, RG_MAX_KS);
    return IQA_EINVAL;
}

}  // namespace iqa

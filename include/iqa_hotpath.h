/*
 * iqa_hotpath.h -- C ABI of the MI355X-native channelize -> demodulate hot path.
 *
 * This is the drop-in boundary for the DSP stages of rknightion/iq-to-audio
 * (src/iq_to_audio/processing.py and src/iq_to_audio/decoders/).  Every entry
 * point names the reference interface it replaces ("ref:" lines, paths relative to
 * the reference's src/iq_to_audio/).  The reference is pure Python, so its "FFI" is
 * a ctypes binding; INTEGRATION.md shows the stub a maintainer would add.
 *
 * Conventions
 *   - All `*_dev` pointers are DEVICE (HBM) pointers on the current HIP device;
 *     `stream` is a hipStream_t passed as void* (0 = default stream).  Calls only
 *     enqueue work: they never synchronise, allocate or free, so they can be
 *     captured into a hipGraph.
 *   - Complex samples are interleaved float pairs (re, im) == numpy complex64.
 *   - Raw capture frames are interleaved pairs in the capture's own sample format
 *     (iqa_fmt); one frame = one complex sample.
 *   - Return value: IQA_OK or an iqa_status error; iqa_last_error() gives the text
 *     (thread-local).  The Python shim maps IQA_EINVAL -> ValueError, everything
 *     else -> RuntimeError, as the reference raises them.
 *   - Streaming state (FIR history, discriminator previous sample, IIR states, peak)
 *     lives in small caller-owned device buffers whose layouts are given below, so
 *     the library keeps no per-stream or per-capture state and is re-entrant per stream and
 *     per host thread.  What it does keep, process-wide and thread-safe: one bit per
 *     (kernel, device id) "dynamic-LDS limit raised" (atomics), the spectrum entry point's
 *     LRU of rocFFT plans keyed by (device, nfft, batch) behind a mutex, and 64 KiB of pacing
 *     words per device for iqa_channelize_mfma_pairs (the ONE exception to "never allocate":
 *     hipMalloc + hipMemset at the first pair launch on a device -- make that call outside a
 *     stream capture; entries are tagged per launch, nothing is reset afterwards).
 */
#ifndef IQA_HOTPATH_H
#define IQA_HOTPATH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IQA_ABI_VERSION 1
/* Per-segment sums of squares are spread over this many sub-slots (sumsq_dev holds n_segs*IQA_SUMSQ_SLOTS
 * doubles, the caller adds the slots of a segment): float atomics on one address serialise at memory
 * latency, so 40 blocks adding into one word cost ~65 us; eight words cost ~8 us. */
#define IQA_SUMSQ_SLOTS 8

typedef enum {
    IQA_OK = 0,
    IQA_EINVAL = 1, /* bad argument (ValueError in the reference) */
    IQA_EHIP = 2,   /* HIP runtime / launch failure (RuntimeError) */
    IQA_ESTATE = 3  /* call sequence error, e.g. process before setup (RuntimeError) */
} iqa_status;

/* Sample format of a raw capture frame.  ref: input_formats.py:45-94 (_FORMAT_MAP)
 * and the ffmpeg conversion the reference relies on (processing.py:113-158):
 *   S16: x/32768   U8: (x-128)/128   F32: unchanged (also = complex64 stage data). */
typedef enum { IQA_FMT_S16 = 0, IQA_FMT_U8 = 1, IQA_FMT_F32 = 2 } iqa_fmt;

/* ref: IQReader._extract_iq, processing.py:268-279 */
typedef enum { IQA_ORDER_IQ = 0, IQA_ORDER_QI = 1, IQA_ORDER_IQ_INV = 2, IQA_ORDER_QI_INV = 3 } iqa_iq_order;

/* ref: create_decoder, decoders/__init__.py:9-24 */
typedef enum { IQA_DEMOD_NFM = 0, IQA_DEMOD_AM = 1, IQA_DEMOD_USB = 2, IQA_DEMOD_LSB = 3 } iqa_demod;

int iqa_abi_version(void);
const char *iqa_last_error(void);

/* ------------------------------------------------------------------------- *
 * Channelizer: ingest + NCO mix + channel FIR + decimate in ONE kernel.      *
 *                                                                            *
 * ref: the first half of the per-chunk body of ProcessingPipeline.run,       *
 *      processing.py:1088-1096, i.e.                                         *
 *        IQReader._extract_iq        processing.py:268-279                   *
 *        ComplexOscillator.mix       processing.py:289-297                   *
 *        OverlapSaveFIR.process      processing.py:325-346                   *
 *        Decimator.process           processing.py:354-360                   *
 *                                                                            *
 * Math: z[m] = e^{j*2pi*rot(m)} * sum_k g[k] * x[m*D - k],  m = m_first..    *
 * where g[k] = h[k]*e^{+j*s*w*k} are the channel taps pre-rotated on the     *
 * host (so no per-input-sample trig is needed), and rot() is a 64-bit        *
 * fixed-point phase in turns.  Equal to mix -> filter -> keep every D-th     *
 * sample of the reference, with zero initial state (causal convolution, group *
 * delay not compensated).                                                     *
 * ------------------------------------------------------------------------- */
typedef struct {
    int32_t fmt;          /* iqa_fmt of raw frames */
    int32_t ntaps;        /* L */
    int32_t decimation;   /* D >= 1 */
    int32_t conj_sum;     /* 1: conjugate the tap sum before rotation (host folds iq_order into taps + this flag:
                           *    x = c*conj(r)  =>  sum g*x = c*conj(sum conj(g)*r)) */
    int32_t rotate;       /* 0: no output rotation (plain FIR stage), 1: apply rot(m) */
    int32_t reserved;
    uint64_t rot_step;    /* phase advance per OUTPUT sample, turns * 2^64 (wraps) */
    uint64_t rot_base;    /* phase of output m = 0 (global), turns * 2^64 */
    float out_scale_re;   /* constant complex factor applied last (1, j, -j for iq_order) */
    float out_scale_im;
} iqa_chan_params;

/*
 * taps_dev    : float2[ntaps_padded] complex taps in WINDOW order (taps_dev[i] multiplies
 *               x[n0-(L-1)+i]), zero padded to iqa_taps_padded_len(L); already scaled by the
 *               ingest scale (1/32768 for S16, 1/128 for U8).
 * raw_dev     : frames [0, n_frames) of this block; frame 0 has GLOBAL index `consumed`.
 * hist_dev    : the L-1 frames preceding raw_dev (same fmt), or NULL for "all zeros"
 *               (start of capture).  ref: OverlapSaveFIR.state, processing.py:323,341-345.
 * m_first     : global index of the first output to produce; outputs m_first..m_first+n_out-1
 *               need frames up to (m_first+n_out-1)*D, all of which must be < consumed+n_frames.
 * z_out_dev   : float2[n_out].
 */
int64_t iqa_taps_padded_len(int32_t ntaps);
int iqa_channelize(const iqa_chan_params *p, const void *taps_dev, const void *raw_dev, int64_t n_frames,
                   int64_t consumed, const void *hist_dev, int64_t m_first, int64_t n_out, void *z_out_dev,
                   void *stream);

/*
 * int8-MFMA form of iqa_channelize for int16 captures -- and, ring variant only, uint8 ones -- (same reference lines, same result up to the
 * 16-bit fixed-point tap quantisation, ~1e-6 of full scale): the decimating FIR is evaluated as a
 * dense integer GEMM on the matrix cores with exact int32 accumulation (see channelize_mfma.hip).
 * It covers only outputs whose whole read range lies inside raw_dev[0, n_frames): columns
 * (m_first-64)*D+1 ... ; the caller runs iqa_channelize for the few outputs at the block's head
 * (history) and tail.
 * Here (unlike iqa_channelize) `consumed` may be negative: raw_dev then starts |consumed| frames before global
 * frame 0 and those frames must be zero (the filter's zero initial state, OverlapSaveFIR.state processing.py:323);
 * with such a lead-in and some readable slack behind the capture every output of a capture is an interior one.
 *   afrag_dev : tap fragments of THIS pass, k_count*8192 bytes, layout [kstep][rowtile 4][piece 2][lane 64][16 B]
 *               (host: dsp_plan.plan_mfma); unit/c_re/c_im from the same quantisation.
 * One call is one PASS over (q-group, k-step range).  A filter with ceil(L/D) <= 64 whose fragments fit
 * LDS needs a single pass (q_group 0, all k steps, finalize 1).  Longer filters are split into q-groups of
 * 64 tap rows (each pass reads the capture again, shifted by 64*q_group rows), larger decimations into k-step
 * ranges (each pass reads its own part of every row); passes chain their raw sums through
 * partial_out_dev -> partial_in_dev (double2[n_out]) and the last pass (finalize 1) rotates, scales, stores z.
 */
typedef struct {
    int32_t outputs_per_block; /* multiple of 32; LDS = afrag + 16*(outputs_per_block+160) bytes <= 160 KiB */
    int32_t reserved;          /* data-path variant + diagnostics flags.  0 = per-lane row loads; 64 = block-wide
                                * ring through LDS (byte planes staged by loader waves, or LDS-DMA), 64-bit sums (needs iqa_mfma_ring_mode(fmt, D, k_first, k_count, 0) != 0;
                                * its LDS does not depend on outputs_per_block); 64|128 = the ring with 256*S1 + S2
                                * in one int32, for fragments from a quantisation that bounds that sum
                                * (dsp_plan.plan_mfma(acc32=True); iqa_mfma_ring_mode(fmt, D, k_first, k_count, 1) != 0);
                                * bits 0..5 are timing diagnostics, never set in production; 256 = the low tap byte
                                * of these fragments is zero throughout (a hint: the multi-lane launches act on it, lane bit 1) */
    double unit;               /* value of one tap LSB (ingest scale folded in) */
    double c_re, c_im;         /* 128 * sum of quantised taps per output component (low-byte bias) */
    void *debug_stamps;        /* NULL in production; diagnostics builds write per-wave cycle stamps here
                                * (64 B per wave) when bit 1 of `reserved` is set */
    int32_t q_group;           /* tap rows 64*q_group+1 .. 64*q_group+64 */
    int32_t k_first;           /* first k step (32 int16 values each) of this pass */
    int32_t k_count;           /* k steps in this pass; 0 = all remaining */
    int32_t finalize;          /* 1 = last pass */
    const void *partial_in_dev;  /* double2[n_out] raw sums of the previous passes, or NULL */
    void *partial_out_dev;       /* double2[n_out], written when finalize == 0 */
} iqa_mfma_params;
int64_t iqa_mfma_afrag_bytes(int32_t decimation);
/* LDS bytes the ring variant (reserved = 64) spends on its data ring for this decimation; 0 = variant not
 * applicable (it needs D % 4 == 0 and D <= 256).  Every tile of 32 data rows is fetched as 2048*ceil(2D/32)
 * contiguous bytes, so the last output of a ring pass must satisfy
 * (m_last - 1)*D + 512*ceil(2D/32) < consumed + n_frames. */
int64_t iqa_mfma_ring_bytes(int32_t decimation);
/* LDS bytes of a workgroup of the ring kernel iqa_mfma_ring_mode selects for this pass (0: none): a launch may give a CU
 * as many workgroups as fit into its 160 KiB (two at <= 3 k steps -- short rows, e.g. D = 26 at 2.5 MS/s -- where a
 * second one hides the first one's per-round latency). */
int64_t iqa_mfma_ring_lds_bytes(int32_t fmt, int32_t decimation, int32_t k_first, int32_t k_count, int32_t acc32);
/* Which ring kernel covers a pass over k steps [k_first, k_first + k_count) at this decimation: 0 = none (int16: use
 * the per-lane kernel, reserved = 0; uint8: use iqa_channelize), 1 = contiguous slots (int16, all k steps in one
 * pass, D % 4 == 0, D <= 256), 2 = row-staged slots (int16 or uint8, any D, k_count <= 11, int32 sums only:
 * acc32 != 0).  Mode 2 reads exactly the frames the per-lane kernel reads; mode 1 needs the slack described above.
 * For uint8 captures (fmt = IQA_FMT_U8, reserved = 64|128) the data have a single byte piece: pass
 * unit = tap LSB / 256 and c_re = c_im = 0 (dsp_plan.plan_mfma does). */
int32_t iqa_mfma_ring_mode(int32_t fmt, int32_t decimation, int32_t k_first, int32_t k_count, int32_t acc32);
int iqa_channelize_mfma(const iqa_chan_params *p, const iqa_mfma_params *q, const void *afrag_dev,
                        const void *raw_dev, int64_t n_frames, int64_t consumed, int64_t m_first, int64_t n_out,
                        void *z_out_dev, void *stream);

/*
 * Several channels of ONE capture in one launch of the ring kernel (shared ingest).
 * ref: the reference CLI runs a whole pipeline per --ft target over the same file (cli.py:683-710, at most five
 * targets :514-521); here every (channel, tap-row group) is a LANE of one launch: the lanes share the capture, the
 * decimation, the k-step range and the output range, and differ in their tap fragments, scale, rotation and output.
 * Lanes of the same stretch of the capture run at the same time on the CUs of one XCD, so the stretch crosses the
 * fabric once (channelize_ring.hip).  int32 sums only: fragments from dsp_plan.plan_mfma(acc32=True);
 * iqa_mfma_ring_mode(fmt, D, k_first, k_count, 1) must be non-zero.  At most 16 lanes per call.
 * Two lanes may share a q_group (their data rows): a group's taps and the residue of their quantisation as a second lane
 * (the "fine" precision of the host pipeline, dsp_plan.plan_mfma(residual=True)).
 * A filter with several tap-row groups is several lanes with finalize = 0, each writing its raw sums to its own
 * partial_out_dev; iqa_mfma_combine adds them in group order and finishes z.  A decimation whose k steps need several
 * passes is several calls chained through partial_in_dev / partial_out_dev per lane, as with iqa_channelize_mfma.
 * outputs_per_block: multiple of 32; the launch has 8*ceil(ceil(n_out/outputs_per_block)/8)*n_lanes workgroups -- one
 * workgroup per CU (256) when ceil(n_out/outputs_per_block) = 8*floor(32/n_lanes).
 */
typedef struct {
    const void *afrag_dev;      /* this lane's tap fragments, first k step of the pass (as for iqa_channelize_mfma) */
    void *z_out_dev;            /* finalize != 0: float2[n_out] */
    const void *partial_in_dev; /* double2[n_out] raw sums of this lane's earlier k-step passes, or NULL */
    void *partial_out_dev;      /* finalize == 0: double2[n_out] */
    double unit, c_re, c_im;    /* as in iqa_mfma_params */
    uint64_t rot_step, rot_base;        /* as in iqa_chan_params */
    float out_scale_re, out_scale_im;
    int32_t q_group, finalize, conj_sum, rotate;
    int32_t raw_partials;       /* finalize == 0 and no partial_in: partial_out_dev is int32[2*n_out], the integer sums
                                 * (256*S1 + S2 per component) themselves; iqa_mfma_combine scales them (raw_scale) */
    int32_t reserved;           /* bit 0: 64-bit sums ((S1 << 32) + S2 per component: fragments WITHOUT the int32 bound, 16-bit
                                 * taps -- dsp_plan.plan_mfma(acc32=False)); the same for every lane of a launch; contiguous
                                 * slots only, no raw_partials.  See iqa_mfma_ring_lanes.
                                 * bit 1: this lane's LOW tap byte is zero throughout (piece 1 of every fragment: the first
                                 * lane of a tap-row group under dsp_plan.plan_mfma(residual=True)): the kernel skips the
                                 * q2*hi product of every k step (two matrix instructions per k step instead of three). */
} iqa_mfma_lane;
int iqa_channelize_mfma_multi(int32_t fmt, int32_t decimation, int32_t k_first, int32_t k_count,
                              int32_t outputs_per_block, const iqa_mfma_lane *lanes, int32_t n_lanes,
                              const void *raw_dev, int64_t n_frames, int64_t consumed, int64_t m_first,
                              int64_t n_out, void *stream);

/* The same launch with TWO lanes per workgroup: lanes[2i] and lanes[2i+1] (n_lanes even) share every staged tile of the
 * capture -- the workgroup's first four waves hold the tap rows of one, the other four those of the other, one tile
 * per round: half the L2 -> LDS traffic per lane and a ring twice as deep in rounds.  lanes[2i].q_group >=
 * lanes[2i+1].q_group (the first lane's stream is the one staged; the second works two rounds per group of difference
 * behind); lanes[2i+1].afrag_dev may be NULL: a pair without a second lane (that half of the workgroup idles).  Same
 * arguments otherwise, same results bit for bit as iqa_channelize_mfma_multi on the same lanes;
 * 8*ceil(ranges/8)*n_lanes/2 workgroups.  Available where iqa_mfma_ring_pairs(fmt, D, k_first, k_count) != 0 (int16 captures, contiguous slots,
 * 9..16 k steps: the decimations whose single-lane kernel runs without loader waves).
 * ref: the same CLI loop over --ft targets, cli.py:683-710. */
int iqa_channelize_mfma_pairs(int32_t fmt, int32_t decimation, int32_t k_first, int32_t k_count,
                              int32_t outputs_per_block, const iqa_mfma_lane *lanes, int32_t n_lanes,
                              const void *raw_dev, int64_t n_frames, int64_t consumed, int64_t m_first,
                              int64_t n_out, void *stream);
int32_t iqa_mfma_ring_pairs(int32_t fmt, int32_t decimation, int32_t k_first, int32_t k_count);
/* What the multi-lane launches offer for a pass over k steps [k_first, k_first + k_count) at this decimation and sum
 * width (acc32 != 0: one int32 per component; 0: one int64, iqa_mfma_lane.reserved bit 0): bit 0 = iqa_channelize_mfma_multi,
 * bit 1 = iqa_channelize_mfma_pairs.  0: no shared-ingest launch (64-bit sums need contiguous slots: D % 4 == 0, D <= 240;
 * their pairs 9..14 k steps).  The "full" precision of the host pipeline puts a filter's tap-row groups AND the residue of
 * their quantisation into such a launch as lanes (ref: the per---ft loop of the reference CLI, cli.py:683-710). */
int32_t iqa_mfma_ring_lanes(int32_t fmt, int32_t decimation, int32_t k_first, int32_t k_count, int32_t acc32);
/* z[m_first + i] = finish(sum_k partials_dev[k][i]): the float32 conversion, conjugation, rotation and scaling of the
 * kernels' own emission (p supplies conj_sum, rotate, rot_step, rot_base, out_scale).  1..16 buffers of double2[n_out]
 * (a filter's tap-row groups -- and, with residual quantisation, each group's second lane: dsp_plan.plan_mfma(residual=True)),
 * or -- raw_scale != NULL -- of int32[2*n_out] written by lanes with raw_partials = 1; raw_scale is a HOST array of
 * n_partials triples {unit, c_re, c_im} (the lane's own values), applied as (256*v + c)*unit before the sum. */
int iqa_mfma_combine(const iqa_chan_params *p, const void *const *partials_dev, int32_t n_partials,
                     const double *raw_scale, int64_t m_first, int64_t n_out, void *z_out_dev, void *stream);

/* Copy the last L-1 frames of (hist | raw) into hist (handles n_frames < L-1 by shifting).
 * ref: OverlapSaveFIR.process state update, processing.py:341-345.
 * hist_next_dev must not alias hist_dev. */
int iqa_history_update(int32_t fmt, int32_t ntaps, const void *hist_dev, const void *raw_dev, int64_t n_frames,
                       void *hist_next_dev, void *stream);

/* ------------------------------------------------------------------------- *
 * Stand-alone stages of the pluggable stage API                              *
 * ------------------------------------------------------------------------- */

/* ref: ComplexOscillator.mix, processing.py:289-297 (+ ingest conversion when fmt != F32).
 * out[i] = cvt(in[i]) * exp(j*(phase0 + step*i)), phase ramp evaluated in float64 exactly as
 * the reference does; the caller carries phase0 = (phase0 + step*n) mod 2pi on the host. */
int iqa_oscillator_mix(int32_t fmt, int32_t iq_order, const void *in_dev, int64_t n, double phase0, double step,
                       void *out_dev, void *stream);

/* ref: Decimator.process, processing.py:354-360.  out[i] = in[first + i*D]. */
int iqa_decimate(const void *in_dev, int64_t n, int64_t first, int32_t D, void *out_dev, int64_t n_out,
                 void *stream);

/* mean(|z|^2) over z[skip:n] accumulated in float64 into *power_dev (double[1], overwritten).
 * Up to 65536 samples (the mixer-sign probes) one workgroup WRITES the result -- no memset, no atomics -- so
 * power_dev may then be mapped pinned host memory (the probe's read-back without a copy).
 * ref: choose_mix_sign, processing.py:650-658; baseband_power, processing.py:1105. */
int iqa_mean_power(const void *z_dev, int64_t n, int64_t skip, void *power_dev, void *stream);

/* The same for `parts` stretches of n_each samples that lie back to back in z_dev: power_dev[p] = mean(|z|^2) over
 * z[p*n_each + skip : (p+1)*n_each].  One launch when a stretch has at most 65536 samples (then written, not
 * accumulated: power_dev may be mapped pinned host memory) -- the two mixer-sign probes of choose_mix_sign
 * (processing.py:623-663) side by side. */
int iqa_mean_power_batch(const void *z_dev, int64_t n_each, int32_t parts, int64_t skip, void *power_dev, void *stream);

/* Wideband level of raw capture frames: *mean_square_out (double[1], WRITTEN by one workgroup: device or mapped pinned
 * host memory) = mean of value^2 over up to 65536 values -- eight 16 KiB stretches spread evenly over raw_dev[0 : n_values] (int16 / uint8 - 128 /
 * float32 values, I and Q alike; raw_dev 16-byte aligned).  The caller scales: wideband RMS of the complex samples =
 * sqrt(2 * mean_square) * ingest scale.  It is the reference level of the precision guard of the fixed-point
 * channelizers (a channel far below the wideband level is re-run at a finer precision); the reference needs none -- its
 * filter runs in complex128 whatever the levels, processing.py:300-346 -- and the warm-up block it is measured on is the
 * one choose_mix_sign inspects, processing.py:1027-1043. */
int iqa_raw_level(int32_t fmt, const void *raw_dev, int64_t n_values, void *mean_square_out, void *stream);

/* ------------------------------------------------------------------------- *
 * Demodulators (channel rate)                                                 *
 * ------------------------------------------------------------------------- */

/* ref: QuadratureDemod.process, decoders/nfm.py:17-24.
 * out[i] = atan2 of z[i]*conj(z[i-1]); prev_dev = float2[1] state (init 1+0j), updated. */
int iqa_quadrature(const void *z_dev, int64_t n, void *prev_dev, void *out_dev, void *stream);

/* ref: AMDecoder.process envelope, decoders/am.py:28.  out[i] = |z[i]| (float32). */
int iqa_envelope(const void *z_dev, int64_t n, void *out_dev, void *stream);

/* ref: SSBDecoder.process, decoders/ssb.py:42-43.  out[i] = real(z[i]) for usb AND lsb. */
int iqa_real_part(const void *z_dev, int64_t n, void *out_dev, void *stream);

/*
 * First-order recurrences as parallel affine scans (float64 scan arithmetic).
 *
 * iqa_deemphasis : y[n] = (1-a)*x[n] + a*y[n-1]        ref: DeemphasisFilter.process, decoders/nfm.py:48-62
 *                  state_dev = double[1] holding y[last] (reference keeps a*y[last]); init 0.
 * iqa_dc_block   : y[n] = x[n] - x[n-1] + r*y[n-1]     ref: DCBlocker.process, decoders/common.py:16-30
 *                  state_dev = double[2] {x[last], y[last]}; init 0,0.
 * iqa_agc        : g[n] = g[n-1] + decay*(target/|x[n]| - g[n-1]) if |x[n]| > 1e-6 else g[n-1];
 *                  out[n] = x[n]*g[n]; g restarts at 1.0 at every multiple of `reset_period`
 *                  counted from element index `reset_phase` (the reference restarts it on every
 *                  process() call, i.e. at every chunk boundary).  ref: SSBDecoder._apply_agc,
 *                  decoders/ssb.py:65-80.  reset_starts_dev: optional sorted int64[n_resets] of
 *                  element indices where the gain restarts (index 0 always restarts).
 * work_dev: scratch, at least iqa_scan_workspace_bytes(n) bytes.
 */
int64_t iqa_scan_workspace_bytes(int64_t n);
int iqa_deemphasis(const void *x_dev, int64_t n, double alpha, void *state_dev, void *y_dev, void *work_dev,
                   void *stream);
int iqa_dc_block(const void *x_dev, int64_t n, double radius, void *state_dev, void *y_dev, void *work_dev,
                 void *stream);
int iqa_agc(const void *x_dev, int64_t n, double target, double decay, const void *reset_starts_dev,
            int64_t n_resets, void *y_dev, void *work_dev, void *stream);

/*
 * Whole demodulator + AudioWriter.write for a block of channel samples, in three launches.
 * ref: decoder.process (processing.py:1128) for nfm / am / usb / lsb as listed above, followed by
 *      AudioWriter.write (processing.py:1147 -> :440-456) and the per-chunk rms statistic.
 * The source stage (discriminator / |z| / real) and the sink (pre-clip peak, clip +-0.99, per-segment
 * sum of squares) are fused into the scan passes.  state_dev: 32 bytes {float2 prev (init 1+0j);
 * double y_last; double x_last, y_last}, carried across calls.  seg_starts_dev: sorted int64 chunk
 * starts within this block (seg_starts[0] == 0): AGC restarts + statistics segments.
 * scratch_dev: float[n], only used by SSB with AGC.  work_dev: iqa_scan_workspace_bytes(n).
 */
typedef struct {
    int32_t mode;        /* iqa_demod */
    int32_t agc_enabled; /* honoured for USB/LSB only, as in the reference */
    double deemph_alpha; /* exp(-1/(fs_ch*tau)) */
    double dc_radius;    /* 0.995 */
    double agc_target;   /* 10^(-12/20) */
    double agc_decay;    /* 0.001 */
} iqa_demod_params;
int iqa_demodulate(const iqa_demod_params *p, const void *z_dev, int64_t n, void *state_dev,
                   const void *seg_starts_dev, int64_t n_segs, void *peak_dev, void *sumsq_dev, void *audio_out_dev,
                   void *scratch_dev, void *work_dev, void *stream);
/* The same for the FIRST block of a stream: a decoder that has seen nothing (decoder setup / AudioWriter creation in
 * ProcessingPipeline.run, processing.py:1040-1066).  state_dev is not read (prev = 1+0j, filter states 0; it receives the
 * outgoing state as usual), peak_dev and sumsq_dev[0 .. n_segs*IQA_SUMSQ_SLOTS) are cleared by the call itself: no
 * reset copy in front of it (a node less in a captured step).  n > 0. */
int iqa_demodulate_from_reset(const iqa_demod_params *p, const void *z_dev, int64_t n, void *state_dev,
                              const void *seg_starts_dev, int64_t n_segs, void *peak_dev, void *sumsq_dev,
                              void *audio_out_dev, void *scratch_dev, void *work_dev, void *stream);

/* ref: AudioWriter.write, processing.py:440-456: peak = max(peak, max|a|) BEFORE the clip, then
 * clip to +-0.99.  peak_dev = float[1] (running, init 0).  In-place allowed (out_dev == a_dev).
 * Also accumulates sum(a^2) (pre-clip, float64) into sumsq_dev[seg*IQA_SUMSQ_SLOTS + slot] for the rms_dbfs statistic
 * (ref: decoders/nfm.py:88-89), where seg = index into seg_starts_dev (sorted int64[n_segs],
 * seg_starts[0] == 0); pass NULL/0 to skip.  peak_dev and out_dev may each be NULL (statistics only). */
int iqa_writer_clip(const void *a_dev, int64_t n, void *peak_dev, const void *seg_starts_dev, int64_t n_segs,
                    void *sumsq_dev, void *out_dev, void *stream);

/* ------------------------------------------------------------------------- *
 * 48 kHz resampler (replaces the `ffmpeg -ar 48000` leg, processing.py:399-418) *
 * BUILD-DEFINED SPEC (the reference's is libswresample: parity unpinned).      *
 *   y[j] = sum_t table[p][t] * x[q - (t - T)],  c = (j0+j)*down, q = c / up, p = c % up          *
 * table_dev: double[up][2T+1] polyphase rows; x zero outside [0, n_in).       *
 * y_dev: float32[n_out] and/or pcm16_dev: int16[n_out] = iqa_float_to_pcm16 of the float32 value (the writer's    *
 * `-acodec pcm_s16le` leg in the same pass); either may be NULL, not both.                                       *
 * ------------------------------------------------------------------------- */
int iqa_resample(const void *x_dev, int64_t n_in, const void *table_dev, int32_t up, int32_t down, int32_t T,
                 int64_t j0, int64_t n_out, void *y_dev, void *pcm16_dev, void *stream);

/* float32 -> PCM16 (round-half-even of y*32768, saturated).  Build-defined, see above. */
int iqa_float_to_pcm16(const void *y_dev, int64_t n, void *pcm_dev, void *stream);

/* Audio egress (the drain of AudioWriter, processing.py:433-438, without a host thread): copy nbytes from device
 * memory into MAPPED pinned host memory (hipHostMalloc / torch pin_memory) with `workgroups` small workgroups
 * (<= 0: 8), so that the copy can run beside a kernel that occupies every CU.  Both pointers 16-byte aligned. */
int iqa_trickle_copy(const void *src_dev, void *dst_mapped, int64_t nbytes, int32_t workgroups, void *stream);

/* float32 capture -> int16 copy when, and only when, every value is k / 32768 with k an integer in [-32768, 32767]
 * (what SDR software writes for int16 / 12-bit / int8 ADC samples): s16_out[i] = x[i] * 32768, and *flag_dev (int32,
 * zeroed by the caller) is OR-ed with 1 if any value is NOT of that form -- the copy is then not the capture.  Lets
 * cf32 captures that are integer captures in disguise take the matrix-core channelizers (ingest: IQReader._extract_iq,
 * processing.py:268-279, with ffmpeg's f32le -> float being the identity).  f32_dev 16-byte aligned. */
int iqa_f32_to_s16_exact(const void *f32_dev, int64_t n_values, void *s16_out_dev, void *flag_dev, void *stream);

/* A float32 capture as TWO int16 planes, x = 2^shift (hi + lo / 32768) / 32768 (hi = rint(2^(15-shift) x), |lo| <= 16384),
 * exact to 2^(shift-31) of full scale: the linear channel filter then gives z = 2^shift (z(hi) + 2^-15 z(lo)) from two
 * passes of the int16 matrix-core channelizers -- a float capture that is NOT on the 2^-15 grid (RTL-SDR's
 * (u - 127.5) / 127.5, k / 32767, resampled recordings) leaves the float32 VALU kernel too.  shift: headroom in bits (0:
 * values within [-1, 1 - 2^-16]; 1: within +-2; ...; the caller picks it from the warm-up block's largest value).
 * *flag_dev (int32, caller-zeroed) |= 1 when a value does not fit (or is a NaN: the planes are then not the capture),
 * |= 2 when some lo != 0 (otherwise hi alone IS the capture: iqa_f32_to_s16_exact's case at shift 0).  f32_dev 16-byte
 * aligned, outputs 8-byte aligned, n_values each.
 * ref: the float32 ingest of IQReader (ffmpeg hands the reference everything as f32le, processing.py:113-158, 268-279;
 * input_formats.py:62-64, 86-91). */
int iqa_f32_split_s16(const void *f32_dev, int64_t n_values, int32_t shift, void *hi_out_dev, void *lo_out_dev, void *flag_dev,
                      void *stream);

/* ------------------------------------------------------------------------- *
 * Spectrum / waterfall                                   *
 * ------------------------------------------------------------------------- */

/* ref: spectrum.py _SlidingFFT.psd :143-171, compute_psd :15-45, the frame loop of streaming_waterfall :58-93.
 * For f in [0, n_frames): frame = samples[first + f*hop : ... + use] (fmt / iq_order as in iqa_oscillator_mix),
 *   X = FFT_nfft(complex128(frame) * window[0:use], zero-padded to nfft)          (rocFFT, double complex)
 *   psd_db[f][k] = 10*log10(|X[(k - nfft/2) mod nfft]|^2 / scale + 1e-18)           (fftshift-ed, dB)
 * window_dev = double[use] (np.hanning(use)); scale = use*sample_rate*win_power + 1e-18 (spectrum.py:39,167).
 * work_dev = double2[n_frames*nfft] scratch.  Outputs, each optional (NULL): psd_db_dev double[n_frames][nfft],
 * psd_db_f32_dev float[n_frames][nfft] (the waterfall's slices, spectrum.py:181), sum_db_dev double[nfft]
 * += sum over the batch's frames (the averaged PSD accumulates dB values, spectrum.py:79-82).
 * FFT plans are cached inside the library per (nfft, n_frames). */
int iqa_psd_frames(int32_t fmt, int32_t iq_order, const void *samples_dev, int64_t n_samples, int64_t first,
                   int64_t hop, int32_t n_frames, int32_t nfft, int32_t use, const void *window_dev, double scale,
                   void *work_dev, void *psd_db_dev, void *psd_db_f32_dev, void *sum_db_dev, void *stream);

/* ref: _WaterfallAggregator._maybe_reduce, spectrum.py:190-208.  out[r] = float32((double(in[2r]) + double(in[2r+1]))/2),
 * an odd last row is copied; out_dev (ceil(n_rows/2) rows) must not alias rows_dev. */
int iqa_pair_average_rows(const void *rows_dev, int32_t n_rows, int32_t n_cols, void *out_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* IQA_HOTPATH_H */

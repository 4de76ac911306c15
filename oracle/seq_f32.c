/*
 * oracle/seq_f32.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Scalar C restatement of the reference's two per-sample Python loops, with the
 * exact NumPy-2 "weak scalar" float32 rounding sequence they execute:
 *
 *   dc_block_f32 : DCBlocker.process      reference decoders/common.py:16-30
 *   agc_f32      : SSBDecoder._apply_agc  reference decoders/ssb.py:65-80
 *
 * Why C: the reference runs these as Python `for` loops over np.float32 scalars;
 * restating them in NumPy vector form would change the rounding order, and a
 * pure-Python loop is too slow for the full-size parity cases.  oracle/cpu_ref.py
 * also carries the pure-Python statement (dc_block_py / agc_py); the unit tests
 * check that both agree bit-for-bit and that both agree with the imported
 * reference on the golden fixtures.
 *
 * Build: `make -C oracle` (gcc -O2 -ffp-contract=off, so no FMA contraction
 * changes the rounding sequence).
 */
#include <math.h>
#include <stddef.h>

/*
 * y[n] = x[n] - x[n-1] + r*y[n-1]
 *
 * Rounding sequence (reference decoders/common.py:23-27):
 *   - inside a call, after the first sample, x_prev / y_prev are np.float32 and
 *     the Python float r is a weak scalar: r*y_prev is float32(r)*y_prev in f32;
 *   - on the FIRST sample of every call x_prev / y_prev are Python floats
 *     (`float(x_prev)` at :28-29): r*y_prev is a double product that is rounded
 *     to float32 when added to the np.float32 difference.
 * State is handed across calls as doubles holding float32 values.
 */
void dc_block_f32(const float *x, float *y, size_t n, double r, double *x_prev_io,
                  double *y_prev_io)
{
    if (n == 0)
        return;
    const float rf = (float)r;
    /* first sample: sample(np.float32) - x_prev(py float) -> f32;
       r(py float) * y_prev(py float) -> double, then weak-cast to f32 on the add */
    float xp = (float)(*x_prev_io);
    float d = x[0] - xp;
    float ry = (float)(r * (*y_prev_io));
    float yp = d + ry;
    y[0] = yp;
    xp = x[0];
    for (size_t i = 1; i < n; ++i) {
        float s = x[i];
        float diff = s - xp;
        float fb = rf * yp;
        float out = diff + fb;
        y[i] = out;
        xp = s;
        yp = out;
    }
    *x_prev_io = (double)xp;
    *y_prev_io = (double)yp;
}

/*
 * gain restarts at 1.0 on every call (reference decoders/ssb.py:72);
 * per sample: if |s| > 1e-6: gain += decay*(target/|s| - gain); out = s*gain.
 * All arithmetic is float32 (np.float32 sample with weak Python-float scalars);
 * the comparison against 1e-6 is done in float32 as NumPy 2 does.
 */
void agc_f32(const float *x, float *y, size_t n, double target, double decay)
{
    const float tf = (float)target;
    const float df = (float)decay;
    const float thr = (float)1e-6;
    float gain = 1.0f;
    for (size_t i = 0; i < n; ++i) {
        float s = x[i];
        float mag = fabsf(s);
        if (mag > thr) {
            float desired = tf / mag;
            float delta = desired - gain;
            float step = df * delta;
            gain = gain + step;
        }
        y[i] = s * gain;
    }
}

/*
 * The same two recurrences in float64 -- NOT the reference's arithmetic.  Used by the SSB+AGC parity tests to
 * separate LOGIC (which sample restarts the gain, which samples pass the 1e-6 threshold, how the DC blocker's state
 * crosses block edges: must agree with the GPU to ~1e-6) from ROUNDING (float32 sequential in the reference, float64
 * scans on the GPU: amplified by the AGC's 1/|s|, see tests/test_gpu_configs.py::ssb_agc_evidence).
 * Interfaces mirror the float32 functions: float32 samples in and out, the DC-blocked sample is rounded to float32
 * before the AGC sees it (as both implementations store it), the threshold test and target/|s| are the float32 ones.
 */
void dc_block_f64(const float *x, float *y, size_t n, double r, double *x_prev_io, double *y_prev_io)
{
    /* the input difference x[n] - x[n-1] is formed in float32, exactly as the reference forms it (decoders/common.py:24,
       two np.float32 operands); only the recurrence on y runs in float64 */
    float xp = (float)(*x_prev_io);
    double yp = *y_prev_io;
    const double rr = (double)(float)r;
    for (size_t i = 0; i < n; ++i) {
        const float diff = x[i] - xp;
        const double out = (double)diff + rr * yp;
        y[i] = (float)out;
        xp = x[i];
        yp = out;
    }
    *x_prev_io = (double)xp;
    *y_prev_io = yp;
}

void agc_f64(const float *x, float *y, size_t n, double target, double decay)
{
    const double tf = (double)(float)target, df = (double)(float)decay;
    const float thr = (float)1e-6;
    double gain = 1.0;
    for (size_t i = 0; i < n; ++i) {
        const float s = x[i];
        const float mag = fabsf(s);
        if (mag > thr)
            gain += df * ((double)((float)tf / mag) - gain);
        y[i] = (float)((double)s * gain);
    }
}

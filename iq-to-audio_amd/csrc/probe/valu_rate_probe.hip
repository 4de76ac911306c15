// valu_rate_probe.hip -- issue cost of the float64 instructions the resampler's inner loop is made of, per wave64:
// v_fma_f64, v_add_f64, v_cvt_f64_f32, v_cvt_f32_f64, and v_fma_f32 for scale.  One workgroup per CU, WAVES waves per
// SIMD, each wave a run of independent chains long enough to hide the pipeline's latency; cycles by s_memtime.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate_probe.hip -o valu_rate_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int OP>
__global__ __launch_bounds__(1024) void k_rate(float *sink, int iters, unsigned long long *cycles, float seed)
{
    double d[8];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        d[i] = seed + i + threadIdx.x;
        f[i] = seed * 2 + i + threadIdx.x;
    }
    const double m = 1.0 + seed * 1e-9, c = seed * 1e-12;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(m), "v"(c));
                if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c));
                if (OP == 2) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
                if (OP == 3) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
                if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                if (OP == 5) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(m));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += static_cast<float>(d[i]) + f[i];
    if (acc == 123.456f) sink[threadIdx.x] = acc;
    if (threadIdx.x == 0) {
        cycles[blockIdx.x * 2] = t1 - t0;
        cycles[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

template <int OP>
void run(const char *name, int waves_per_simd, float *sink, unsigned long long *cyc)
{
    // up to 4 waves per SIMD in one workgroup per CU; 8 = two workgroups of 16 waves per CU
    const int iters = 2000, blocks = waves_per_simd > 4 ? 512 : 256, threads = 256 * (waves_per_simd > 4 ? waves_per_simd / 2 : waves_per_simd);
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(threads), 0, 0, sink, iters, cyc, 1.0f);
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(threads), 0, 0, sink, iters, cyc, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(512 * 2);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0, real = 0;
    for (int b = 0; b < blocks; ++b) mean += static_cast<double>(h[b * 2]), real += static_cast<double>(h[b * 2 + 1]);
    mean /= blocks, real /= blocks;
    // s_memrealtime counts at 100 MHz: ns = 10 * ticks
    const double ns = 10.0 * real / (iters * 32.0);
    printf("%-14s %d wave(s)/SIMD: %.3f ns per wave-instruction, %.3f ns per instruction slot of the SIMD (memtime/realtime = %.2f)\n", name,
           waves_per_simd, ns, ns / waves_per_simd, mean / real);
}

int main()
{
    float *sink;
    unsigned long long *cyc;
    hipMalloc(&sink, 4096);
    hipMalloc(&cyc, 512 * 2 * 8);
    for (int w : {1, 2, 4, 8}) {
        run<4>("v_fma_f32", w, sink, cyc);
        run<0>("v_fma_f64", w, sink, cyc);
        run<1>("v_add_f64", w, sink, cyc);
        run<5>("v_mul_f64", w, sink, cyc);
        run<2>("v_cvt_f64_f32", w, sink, cyc);
        run<3>("v_cvt_f32_f64", w, sink, cyc);
    }
    return 0;
}

// kstep_probe.hip -- how busy can the matrix pipe get with the ring kernel's k-step body, two waves per SIMD?
// Isolates the steady state of channelize_ring.hip's inner loop from its round structure (barriers, DMA, scatter):
//   mode 0: 3 x v_mfma_i32_32x32x32_i8 per step, operands constant                       (the pipe's own rate)
//   mode 1: + the byte split (8 v_perm_b32 + 4 v_xor) of a data fragment held in registers
//   mode 2: + the two ds_read_b128 per step from an LDS tile at the padded row pitch (prefetched two steps ahead)
//   mode 3: mode 2 with a workgroup barrier every KS steps (one "tile") -- the round structure without DMA and scatter
//   mode 4: mode 3 + the scatter (16 ds_add_u32 of 256*acc1 + acc2 per tile)
// 256 workgroups x 8 waves (two per SIMD), tap fragments in registers as in the kernel.  Prints MFMAs per SIMD-cycle
// against the pipe's 1 / 32.   Build: hipcc --offload-arch=gfx950 -O3 kstep_probe.hip -o kstep_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <type_traits>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int KS = 13;
constexpr int PITCH = (4 * KS + 1) * 16;  // bytes: 52 units of data + 1 of padding (D = 208)

template <int MODE, int VAR = 0>
__global__ __launch_bounds__(512, 2) void k_probe(const v4i *taps, int *sink, int tiles, unsigned long long *cycles)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    v4i fq[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        fq[ks][0] = taps[(ks * 2 + 0) * 64 + lane];
        fq[ks][1] = taps[(ks * 2 + 1) * 64 + lane];
    }
    for (int i = tid; i < 32 * PITCH / 4 + 1024; i += 512) reinterpret_cast<int *>(smem)[i] = i * 2654435761u;
    int *s_acc = reinterpret_cast<int *>(smem + 32 * PITCH + 1024);
    for (int i = tid; i < 1152; i += 512) s_acc[i] = 0;
    __syncthreads();
    const char *la = smem + (lane & 31) * PITCH + 32 * (lane >> 5);
    const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v16i acc1 = zero16, acc2 = zero16;
    v4i d0 = taps[lane], d1 = taps[64 + lane];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
        if (MODE >= 3) asm volatile("s_barrier" ::: "memory");
        v4i dd[KS][2];
        constexpr int PD = (VAR & 1) ? 4 : 2;
        auto rd = [&](int ks) {
            if constexpr (VAR & 16) {
                typedef int v2i __attribute__((ext_vector_type(2)));
                const v2i a0 = *reinterpret_cast<const v2i *>(la + 64 * ks), a1 = *reinterpret_cast<const v2i *>(la + 64 * ks + 8);
                const v2i b0 = *reinterpret_cast<const v2i *>(la + 64 * ks + 16), b1 = *reinterpret_cast<const v2i *>(la + 64 * ks + 24);
                dd[ks][0] = v4i{a0.x, a0.y, a1.x, a1.y};
                dd[ks][1] = v4i{b0.x, b0.y, b1.x, b1.y};
            } else {
                dd[ks][0] = *reinterpret_cast<const v4i *>(la + 64 * ks);
                dd[ks][1] = (VAR & 8) ? dd[ks][0] : *reinterpret_cast<const v4i *>(la + 64 * ks + 16);
            }
        };
        if (MODE >= 2) {
#pragma unroll
            for (int ks = 0; ks < PD; ++ks) rd(ks);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            v4i hi = d0, lo = d1;
            if (MODE >= 1) {
                const v4i e0 = (MODE >= 2) ? dd[ks][0] : d0, e1 = (MODE >= 2) ? dd[ks][1] : d1;
                hi.x = __builtin_amdgcn_perm(e0.y, e0.x, 0x07050301);
                hi.y = __builtin_amdgcn_perm(e0.w, e0.z, 0x07050301);
                hi.z = __builtin_amdgcn_perm(e1.y, e1.x, 0x07050301);
                hi.w = __builtin_amdgcn_perm(e1.w, e1.z, 0x07050301);
                lo.x = __builtin_amdgcn_perm(e0.y, e0.x, 0x06040200) ^ 0x80808080;
                lo.y = __builtin_amdgcn_perm(e0.w, e0.z, 0x06040200) ^ 0x80808080;
                lo.z = __builtin_amdgcn_perm(e1.y, e1.x, 0x06040200) ^ 0x80808080;
                lo.w = __builtin_amdgcn_perm(e1.w, e1.z, 0x06040200) ^ 0x80808080;
            }
            if (MODE >= 2 && !(VAR & 2) && ks + PD < KS) rd(ks + PD);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][0], hi, (MODE >= 3 && ks == 0) ? zero16 : acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][0], lo, (MODE >= 3 && ks == 0) ? zero16 : acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][1], hi, acc2, 0, 0, 0);
            if (MODE >= 2 && (VAR & 2) && ks + PD < KS) rd(ks + PD);
            if (!(VAR & 4)) __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE >= 4) {
            const unsigned p = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) const void *)(
                s_acc + ((t * 32 + (lane & 31) + 4 * (lane >> 5) + 1) & 511) + (wave & 1) * 32)));
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int comb = (acc1[q] << 8) + acc2[q];
                asm volatile("ds_add_u32 %0, %1 offset:%2" ::"v"(p), "v"(comb), "n"(4 * ((q & 3) + 8 * (q >> 2))));
            }
        } else if (MODE >= 3) {
            asm volatile("" ::"v"(acc1), "v"(acc2));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += acc1[q] + acc2[q];
    if (s == 0x12345678) sink[0] = s;  // keep the accumulators alive
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE, int VAR = 0>
static void run(const v4i *taps, int *sink, unsigned long long *cyc, int tiles)
{
    const size_t lds = 32 * PITCH + 1024 + 1152 * 4 + 4096;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_probe<MODE, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_probe<MODE, VAR>), dim3(256), dim3(512), lds, 0, taps, sink, tiles, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    // 2 waves per SIMD, each 3*KS MFMAs per tile
    const double mfma_per_simd = 2.0 * 3 * KS * tiles;
    const double us = ms * 1e3;
    printf("mode %d var %2d: %8.3f ms for %d tiles per wave; %.1f ns per tile-pair per SIMD; at 32 cycles per MFMA the pipe alone needs %.0f cycles per "
           "tile-pair -> busy %.1f %% if the clock were 2.0 GHz, %.1f %% at 2.4 GHz (s_memtime ticks per block: %llu)\n",
           MODE, VAR, ms, tiles, us * 1e3 / tiles, 2.0 * 3 * KS * 32, 100.0 * mfma_per_simd * 32 / (us * 2000.0), 100.0 * mfma_per_simd * 32 / (us * 2400.0),
           h[0]);
}


// ---- more tap rows per wave: RT row tiles (32 rows each) share every data fragment a wave reads and splits ----------
// WAVES = 8: two waves per SIMD (<= 256 registers each); WAVES = 4: one wave per SIMD (<= 512).
// SYNC = 1: the round structure -- a workgroup barrier in front of every tile and the scatter of the tile's sums (16
// ds_add_u32 per row tile) behind it.  BLK workgroups per CU: BLK = 2 with WAVES = 4 puts two INDEPENDENT barrier domains
// on a CU (each SIMD holds one wave of either), against one domain of 8 waves.
template <int KSX, int RT, int WAVES, int SYNC = 0, int BLK = 1>
__global__ __launch_bounds__(WAVES * 64, BLK) void k_probe_rt(const v4i *taps, int *sink, int tiles, unsigned long long *clk)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PITCHX = (4 * KSX + 1) * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    v4i fq[KSX][RT][2];
#pragma unroll
    for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            fq[ks][rt][0] = taps[((ks * 2 + 0) * 2 + rt) * 64 + lane];
            fq[ks][rt][1] = taps[((ks * 2 + 1) * 2 + rt) * 64 + lane];
        }
    for (int i = tid; i < 32 * PITCHX / 4 + 1024; i += WAVES * 64) reinterpret_cast<int *>(smem)[i] = i * 2654435761u;
    __syncthreads();
    const char *la = smem + (lane & 31) * PITCHX + 32 * (lane >> 5);
    const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v16i acc1[RT], acc2[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc1[rt] = acc2[rt] = zero16;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int *s_win = reinterpret_cast<int *>(smem + 32 * PITCHX + 1024);
    const unsigned win_at = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) const void *)(s_win + (lane & 31) + 4 * (lane >> 5))));
    if (SYNC == 2 && (blockIdx.x & 256)) {  // the second workgroup of a CU starts half a tile late: out of phase for good
        for (int ks = 0; ks < (KSX * RT * 3) / 2; ++ks) acc1[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[0][0][0], fq[0][0][1], acc1[0], 0, 0, 0);
    }
    for (int t = 0; t < tiles; ++t) {
        if (SYNC) asm volatile("s_barrier" ::: "memory");
        v4i dd[KSX][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            dd[ks][0] = *reinterpret_cast<const v4i *>(la + 64 * ks);
            dd[ks][1] = *reinterpret_cast<const v4i *>(la + 64 * ks + 16);
        }
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks) {
            const v4i e0 = dd[ks][0], e1 = dd[ks][1];
            v4i hi, lo;
            hi.x = __builtin_amdgcn_perm(e0.y, e0.x, 0x07050301);
            hi.y = __builtin_amdgcn_perm(e0.w, e0.z, 0x07050301);
            hi.z = __builtin_amdgcn_perm(e1.y, e1.x, 0x07050301);
            hi.w = __builtin_amdgcn_perm(e1.w, e1.z, 0x07050301);
            lo.x = __builtin_amdgcn_perm(e0.y, e0.x, 0x06040200) ^ 0x80808080;
            lo.y = __builtin_amdgcn_perm(e0.w, e0.z, 0x06040200) ^ 0x80808080;
            lo.z = __builtin_amdgcn_perm(e1.y, e1.x, 0x06040200) ^ 0x80808080;
            lo.w = __builtin_amdgcn_perm(e1.w, e1.z, 0x06040200) ^ 0x80808080;
            if (ks + 2 < KSX) {
                dd[ks + 2][0] = *reinterpret_cast<const v4i *>(la + 64 * (ks + 2));
                dd[ks + 2][1] = *reinterpret_cast<const v4i *>(la + 64 * (ks + 2) + 16);
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                acc1[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][rt][0], hi, acc1[rt], 0, 0, 0);
                acc2[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][rt][0], lo, acc2[rt], 0, 0, 0);
                acc2[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][rt][1], hi, acc2[rt], 0, 0, 0);
            }
        }
        if (SYNC) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int comb = (acc1[rt][q] << 8) + acc2[rt][q];
                    asm volatile("ds_add_u32 %0, %1 offset:%2" ::"v"(win_at), "v"(comb), "n"(4 * ((q & 3) + 8 * (q >> 2)) + 1024 * (rt & 1)));
                }
                acc1[rt] = acc2[rt] = zero16;
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += acc1[rt][q] + acc2[rt][q];
    if (s == 0x12345678) sink[0] = s;
    if (tid == 0 && blockIdx.x == 100) {
        clk[0] = c1 - c0;
        clk[1] = r1 - r0;
    }
}

// ---- the round structure without its bubble ----------------------------------------------------------------------
// One barrier per tile as before, but in the MIDDLE of the tile (it announces the NEXT tile's data and releases the one
// before this one), the fragment prefetch running on across the tile boundary, and two sets of sums: the finished tile's
// 16 adds go out one per k step between the next tile's MFMAs instead of draining the pipe in front of a barrier.
template <int KSX, int WAVES, int OFF = 0>  // OFF bits: 1 no barrier, 2 no scatter, 4 scatter by plain stores, 8 prefetch within the tile only
__global__ __launch_bounds__(WAVES * 64, 1) void k_probe_cont(const v4i *taps, int *sink, int tiles, unsigned long long *clk)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PITCHX = (4 * KSX + 1) * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    v4i fq[KSX][2];
#pragma unroll
    for (int ks = 0; ks < KSX; ++ks) {
        fq[ks][0] = taps[((ks * 2 + 0) * 2) * 64 + lane];
        fq[ks][1] = taps[((ks * 2 + 1) * 2) * 64 + lane];
    }
    for (int i = tid; i < 32 * PITCHX / 4 + 1024; i += WAVES * 64) reinterpret_cast<int *>(smem)[i] = i * 2654435761u;
    __syncthreads();
    const char *la = smem + (lane & 31) * PITCHX + 32 * (lane >> 5);
    int *s_win = reinterpret_cast<int *>(smem + 32 * PITCHX + 1024);
    const unsigned win_at = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) const void *)(s_win + (lane & 31) + 4 * (lane >> 5))));
    const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v16i a1[2] = {zero16, zero16}, a2[2] = {zero16, zero16};
    v4i dd[KSX + 2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        dd[ks][0] = *reinterpret_cast<const v4i *>(la + 64 * ks);
        dd[ks][1] = *reinterpret_cast<const v4i *>(la + 64 * ks + 16);
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    auto tile = [&](auto cur_c, bool have_prev) {
        constexpr int CUR = decltype(cur_c)::value, PRV = 1 - CUR;
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks) {
            if (ks == KSX / 2 && !(OFF & 1)) asm volatile("s_barrier" ::: "memory");
            const v4i e0 = dd[ks][0], e1 = dd[ks][1];
            v4i hi, lo;
            hi.x = __builtin_amdgcn_perm(e0.y, e0.x, 0x07050301);
            hi.y = __builtin_amdgcn_perm(e0.w, e0.z, 0x07050301);
            hi.z = __builtin_amdgcn_perm(e1.y, e1.x, 0x07050301);
            hi.w = __builtin_amdgcn_perm(e1.w, e1.z, 0x07050301);
            lo.x = __builtin_amdgcn_perm(e0.y, e0.x, 0x06040200) ^ 0x80808080;
            lo.y = __builtin_amdgcn_perm(e0.w, e0.z, 0x06040200) ^ 0x80808080;
            lo.z = __builtin_amdgcn_perm(e1.y, e1.x, 0x06040200) ^ 0x80808080;
            lo.w = __builtin_amdgcn_perm(e1.w, e1.z, 0x06040200) ^ 0x80808080;
            {  // two steps ahead, across the tile boundary (the barrier of this tile has announced the next one)
                const int nk = (ks + 2) % KSX;
                dd[ks + 2][0] = *reinterpret_cast<const v4i *>(la + 64 * nk);
                dd[ks + 2][1] = *reinterpret_cast<const v4i *>(la + 64 * nk + 16);
            }
            a1[CUR] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][0], hi, a1[CUR], 0, 0, 0);
            a2[CUR] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][0], lo, a2[CUR], 0, 0, 0);
            a2[CUR] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[ks][1], hi, a2[CUR], 0, 0, 0);
            // the previous tile's sums leave between this tile's MFMAs: ceil(16 / KSX) adds per k step
            constexpr int PER = (16 + KSX - 1) / KSX;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int q = ks * PER + j;
                if (q < 16 && have_prev && !(OFF & 2)) {
                    const int comb = (a1[PRV][q] << 8) + a2[PRV][q];
                    if (OFF & 4) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(win_at), "v"(comb), "n"(4 * ((q & 3) + 8 * (q >> 2))));
                    else asm volatile("ds_add_u32 %0, %1 offset:%2" ::"v"(win_at), "v"(comb), "n"(4 * ((q & 3) + 8 * (q >> 2))));
                }
            }
        }
        // hand the two prefetched fragments to the next tile's steps 0 and 1, and clear the set the next-but-one tile sums into
        dd[0][0] = dd[KSX][0], dd[0][1] = dd[KSX][1], dd[1][0] = dd[KSX + 1][0], dd[1][1] = dd[KSX + 1][1];
        a1[PRV] = zero16, a2[PRV] = zero16;
    };
    for (int t = 0; t < tiles; t += 2) {
        tile(std::integral_constant<int, 0>{}, t > 0);
        tile(std::integral_constant<int, 1>{}, true);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int sacc = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) sacc += a1[0][q] + a2[0][q] + a1[1][q] + a2[1][q];
    if (sacc == 0x12345678) sink[0] = sacc;
    if (tid == 0 && blockIdx.x == 100) {
        clk[0] = c1 - c0;
        clk[1] = r1 - r0;
    }
}

template <int KSX, int WAVES, int OFF = 0>
static void run_cont(const v4i *taps, int *sink, int tiles, unsigned long long *clk)
{
    const size_t lds = 32 * ((4 * KSX + 1) * 16) + 4096 + 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_probe_cont<KSX, WAVES, OFF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_probe_cont<KSX, WAVES, OFF>), dim3(256), dim3(WAVES * 64), lds, 0, taps, sink, tiles, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (WAVES / 4.0) * 3 * KSX * tiles;
    unsigned long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = double(h[0]) / double(h[1]) * 0.1;
    printf("k steps %2d, %d wave(s) per SIMD, mid-tile barrier + prefetch across tiles + interleaved scatter (off bits %d): %8.3f ms; in-kernel clock %.3f GHz -> "
           "matrix pipe busy %.1f %%; %.1f ns per 128 tap rows x 32 data rows per CU\n",
           KSX, WAVES / 4, OFF, ms, ghz, 100.0 * mfma_per_simd * 32 / (ms * 1e6 * ghz), ms * 1e6 / tiles / (WAVES / 4.0));
}

template <int KSX, int RT, int WAVES, int SYNC = 0, int BLK = 1>
static void run_rt(const v4i *taps, int *sink, int tiles, unsigned long long *clk)
{
    const size_t lds = 32 * ((4 * KSX + 1) * 16) + 4096 + 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_probe_rt<KSX, RT, WAVES, SYNC, BLK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_probe_rt<KSX, RT, WAVES, SYNC, BLK>), dim3(256 * BLK), dim3(WAVES * 64), lds, 0, taps, sink, tiles, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = BLK * (WAVES / 4.0) * 3 * KSX * RT * tiles;
    unsigned long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = double(h[0]) / double(h[1]) * 0.1;  // in-kernel clock: s_memtime ticks per 100 MHz s_memrealtime tick
    printf("k steps %2d, %d row tile(s) per wave, %d workgroup(s) per CU x %d wave(s) per SIMD%s: %8.3f ms; in-kernel clock %.3f GHz -> matrix pipe busy %.1f %%; "
           "%.1f ns per 128 tap rows x 32 data rows per CU\n",
           KSX, RT, BLK, WAVES / 4, SYNC == 2 ? ", barrier + scatter per tile, second workgroup half a tile late" : SYNC ? ", barrier + scatter per tile" : "", ms, ghz, 100.0 * mfma_per_simd * 32 / (ms * 1e6 * ghz),
           ms * 1e6 / tiles / (BLK * WAVES * RT / 4.0));
}

int main(int argc, char **argv)
{
    const int tiles = argc > 1 ? atoi(argv[1]) : 20000;
    v4i *taps;
    int *sink;
    unsigned long long *cyc;
    hipMalloc(&taps, 16 * 2 * 2 * 64 * 16 + 4096);
    hipMemset(taps, 1, 16 * 2 * 2 * 64 * 16 + 4096);
    hipMalloc(&sink, 64);
    hipMalloc(&cyc, 256 * 8);
    const bool rt_only = argc > 2;
    if (!rt_only) {
    run<0>(taps, sink, cyc, tiles);
    run<1>(taps, sink, cyc, tiles);
    run<2>(taps, sink, cyc, tiles);
    run<3>(taps, sink, cyc, tiles);
    run<4>(taps, sink, cyc, tiles);
    // mode 2 variants: 1 = prefetch 4 steps ahead, 2 = reads behind the step's MFMAs, 4 = compiler's own schedule,
    // 8 = one read per step, 16 = four ds_read_b64 per step
    run<2, 1>(taps, sink, cyc, tiles);
    run<2, 2>(taps, sink, cyc, tiles);
    run<2, 4>(taps, sink, cyc, tiles);
    run<2, 8>(taps, sink, cyc, tiles);
    run<2, 16>(taps, sink, cyc, tiles);
    run<2, 3>(taps, sink, cyc, tiles);
    run<2, 5>(taps, sink, cyc, tiles);
    run<3, 4>(taps, sink, cyc, tiles);  // barrier per tile, compiler's schedule
    run<4, 4>(taps, sink, cyc, tiles);  // + scatter
    }
    run_rt<13, 1, 8>(taps, sink, tiles, cyc);
    run_rt<13, 2, 4>(taps, sink, tiles, cyc);
    run_rt<13, 1, 4>(taps, sink, tiles, cyc);
    run_rt<7, 1, 8>(taps, sink, tiles, cyc);
    run_rt<7, 2, 8>(taps, sink, tiles, cyc);
    run_rt<7, 2, 4>(taps, sink, tiles, cyc);
    run_rt<7, 4, 4>(taps, sink, tiles, cyc);
    // the round structure: one barrier domain of 8 waves against two of 4 on the same CU, and one wave per SIMD with 64 rows
    run_rt<13, 1, 8, 1, 1>(taps, sink, tiles, cyc);
    run_rt<13, 1, 4, 1, 2>(taps, sink, tiles, cyc);
    run_rt<13, 2, 4, 1, 1>(taps, sink, tiles, cyc);
    run_rt<13, 2, 4, 1, 2>(taps, sink, tiles, cyc);
    run_rt<7, 1, 8, 1, 1>(taps, sink, tiles, cyc);
    run_rt<7, 1, 4, 1, 2>(taps, sink, tiles, cyc);
    run_rt<7, 2, 4, 1, 1>(taps, sink, tiles, cyc);
    run_rt<7, 2, 4, 1, 2>(taps, sink, tiles, cyc);
    run_cont<13, 8>(taps, sink, tiles, cyc);
    run_cont<13, 8, 1>(taps, sink, tiles, cyc);
    run_cont<13, 8, 2>(taps, sink, tiles, cyc);
    run_cont<13, 8, 3>(taps, sink, tiles, cyc);
    run_cont<13, 8, 4>(taps, sink, tiles, cyc);
    run_cont<7, 8>(taps, sink, tiles, cyc);
    run_cont<7, 8, 3>(taps, sink, tiles, cyc);
    if (argc > 3) return 0;
    run_rt<13, 1, 4, 2, 2>(taps, sink, tiles, cyc);
    run_rt<13, 2, 4, 2, 2>(taps, sink, tiles, cyc);
    run_rt<7, 1, 4, 2, 2>(taps, sink, tiles, cyc);
    run_rt<7, 2, 4, 2, 2>(taps, sink, tiles, cyc);
    return 0;
}

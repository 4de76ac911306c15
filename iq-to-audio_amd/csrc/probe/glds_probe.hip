// glds_probe.hip -- does global_load_lds_dwordx4 accept 4-byte-aligned (not 16-byte-aligned) per-lane
// global addresses, and is the LDS image base + lane*16?  (needed by the LDS-staged MFMA channelizer)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

__global__ void k(const int *src, int *out, int misalign_dwords)
{
    __shared__ __attribute__((aligned(16))) int lds[64 * 4 * 2];
    const int lane = threadIdx.x;
    // lane l fetches 16 bytes from an arbitrary (permuted, 4-byte aligned) place
    const int *g = src + ((lane * 7) % 64) * 4 + misalign_dwords;
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(g + 512, (__attribute__((address_space(3))) void *)(lds + 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = 0; j < 4; ++j) {
        out[lane * 4 + j] = lds[lane * 4 + j];
        out[256 + lane * 4 + j] = lds[256 + lane * 4 + j];
    }
}

int main()
{
    std::vector<int> h(2048);
    for (int i = 0; i < 2048; ++i) h[i] = i * 3 + 1;
    int *d, *o;
    hipMalloc(&d, 2048 * 4);
    hipMalloc(&o, 512 * 4);
    hipMemcpy(d, h.data(), 2048 * 4, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int mis = 0; mis < 4; ++mis) {
        hipMemset(o, 0, 512 * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, mis);
        std::vector<int> r(512);
        hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                int want0 = h[((l * 7) % 64) * 4 + mis + j], want1 = h[((l * 7) % 64) * 4 + mis + j + 512];
                if (r[l * 4 + j] != want0 || r[256 + l * 4 + j] != want1) ++bad;
            }
        printf("misalign %d dwords: %s (%d bad)\n", mis, bad ? "FAIL" : "ok", bad);
        bad_total += bad;
    }
    return bad_total ? 1 : 0;
}

// glds_probe2.hip -- does global_load_lds_dwordx4 accept 1/2/3-byte misaligned per-lane global addresses (uint8 I/Q
// captures: frames are 2 bytes, rows start on odd frames)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

__global__ void k(const unsigned char *src, unsigned char *out, int mis)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[1024];
    const int lane = threadIdx.x;
    const unsigned char *g = src + lane * 16 + mis;
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = 0; j < 16; ++j) out[lane * 16 + j] = lds[lane * 16 + j];
}

int main()
{
    std::vector<unsigned char> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (unsigned char)((i * 7 + 3) & 255);
    unsigned char *d, *o;
    hipMalloc(&d, 4096);
    hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int mis = 0; mis < 8; ++mis) {
        hipMemset(o, 0, 1024);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, mis);
        std::vector<unsigned char> r(1024);
        hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 1024; ++i) if (r[i] != h[i + mis]) ++bad;
        printf("misalign %d bytes: %s (%d bad)\n", mis, bad ? "FAIL" : "ok", bad);
        bad_total += bad;
    }
    return bad_total ? 1 : 0;
}

// mfma_i8_probe.hip -- empirical lane maps of the gfx950 int8 MFMA forms used by the channelizer.
// The guide gives the bf16 maps and says "other dtypes: check the map with exact integer data".
// Build: hipcc --offload-arch=gfx950 -O2 mfma_i8_probe.hip -o mfma_i8_probe ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// per-lane operands supplied verbatim by the host: a[lane][16 bytes], b[lane][16 bytes]
__global__ void k32(const v4i *a, const v4i *b, int *c)
{
    const int l = threadIdx.x;
    v16i acc = {0};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) c[l * 16 + r] = acc[r];
}
__global__ void k16(const v4i *a, const v4i *b, int *c)
{
    const int l = threadIdx.x;
    v4i acc = {0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) c[l * 4 + r] = acc[r];
}

template <int M, int K>
static int run(bool big)
{
    // hypothesis: lane l holds A[row = l % M][k = KL*(l / M) + j], B[k = KL*(l / M) + j][col = l % M], j = byte 0..15,
    // KL = 16 bytes per lane;  C: 32x32 -> col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5);  16x16 -> col = l&15, row = 4*(l>>4)+r
    const int KL = 16, NREG = big ? 16 : 4;
    std::vector<int8_t> A(M * K), B(K * M);
    for (int i = 0; i < M; ++i)
        for (int k = 0; k < K; ++k) A[i * K + k] = (int8_t)(((i * 7 + k * 3) % 11) - 5);
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < M; ++j) B[k * M + j] = (int8_t)(((k * 5 + j * 13) % 9) - 4);  // asymmetric
    std::vector<int8_t> ha(64 * 16), hb(64 * 16);
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < KL; ++j) {
            const int k = KL * (l / M) + j;
            ha[l * 16 + j] = A[(l % M) * K + k];
            hb[l * 16 + j] = B[k * M + (l % M)];
        }
    void *da, *db;
    int *dc;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dc, 64 * NREG * 4);
    hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice);
    if (big) hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, (const v4i *)da, (const v4i *)db, dc);
    else hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, (const v4i *)da, (const v4i *)db, dc);
    std::vector<int> hc(64 * NREG);
    hipMemcpy(hc.data(), dc, hc.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < NREG; ++r) {
            const int col = big ? (l & 31) : (l & 15);
            const int row = big ? ((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) : (4 * (l >> 4) + r);
            int want = 0;
            for (int k = 0; k < K; ++k) want += (int)A[row * K + k] * (int)B[k * M + col];
            if (want != hc[l * NREG + r]) {
                if (bad < 8) printf("  mismatch lane %d reg %d: got %d want %d\n", l, r, hc[l * NREG + r], want);
                ++bad;
            }
        }
    printf("%s: %s (%d mismatches)\n", big ? "mfma_i32_32x32x32_i8" : "mfma_i32_16x16x64_i8", bad ? "LAYOUT HYPOTHESIS FAILS" : "layout hypothesis OK", bad);
    hipFree(da); hipFree(db); hipFree(dc);
    return bad;
}

int main()
{
    int bad = run<32, 32>(true) + run<16, 64>(false);
    return bad ? 1 : 0;
}

// semantics + rate of the 64-bit DPP move with row_newbcast (gfx90a+ "DP ALU DPP"), the one-instruction
// "lane 15 of the row -> the lanes of one bank" used by the sweeps' bottom-row collector
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(long long* o)
{
    const int lane = threadIdx.x;
    long long src = 1000 + lane + ((long long) (2000 + lane) << 32), dst = -1;
    asm volatile("s_nop 4\n v_mov_b64_dpp %0, %1 row_newbcast:15 row_mask:0xf bank_mask:0x4\n s_nop 4" : "+v"(dst) : "v"(src));
    o[lane] = dst;
    long long d2 = -1;
    int s2 = 3000 + lane, r2 = -1;
    asm volatile("s_nop 4\n v_mov_b32_dpp %0, %1 row_newbcast:15 row_mask:0xf bank_mask:0x8\n s_nop 4" : "+v"(r2) : "v"(s2));
    o[64 + lane] = r2;
}
__global__ void __launch_bounds__(256) rate(long long* o, int iters)
{
    long long a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:15 row_mask:0xf bank_mask:0x1\n"
                         "v_mov_b64_dpp %1, %2 row_newbcast:15 row_mask:0xf bank_mask:0x2\n"
                         "v_mov_b64_dpp %2, %3 row_newbcast:15 row_mask:0xf bank_mask:0x4\n"
                         "v_mov_b64_dpp %3, %0 row_newbcast:15 row_mask:0xf bank_mask:0x8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    }
    o[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}
int main()
{
    long long* o; (void) hipMalloc(&o, 8 * 256 * 256 * 8);
    probe<<<1, 64>>>(o);
    long long h[128]; (void) hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    printf("# v_mov_b64_dpp row_newbcast:15 bank_mask:0x4 (src lo = 1000 + lane, hi = 2000 + lane; dst preset -1): lo / hi per lane\n");
    for (int l = 0; l < 64; ++l) printf("%d/%d%s", (int) (h[l] & 0xffffffff), (int) (h[l] >> 32), (l & 15) == 15 ? "\n" : " ");
    printf("# v_mov_b32_dpp row_newbcast:15 bank_mask:0x8 (src = 3000 + lane)\n");
    for (int l = 0; l < 64; ++l) printf("%d%s", (int) h[64 + l], (l & 15) == 15 ? "\n" : " ");
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    for (int wps = 1; wps <= 8; wps *= 2) {
        rate<<<256 * wps, 256>>>(o, 20); (void) hipDeviceSynchronize();
        (void) hipEventRecord(e0); rate<<<256 * wps, 256>>>(o, 4000); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
        float ms; (void) hipEventElapsedTime(&ms, e0, e1);
        printf("v_mov_b64_dpp newbcast  w%d: %.2f G wave-instr/s/SIMD\n", wps, 4000.0 * 64 * wps / (ms * 1e-3) / 1e9);
    }
    return 0;
}

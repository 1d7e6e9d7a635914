// Peak issue rate of the integer VALU mix spdp_sweep uses (v_add_u32, v_max_i32, v_cmp + v_cndmask,
// DPP moves), in wave-instructions per cycle per SIMD, for 1 .. 8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(256) spin(int* out, int iters, int seed)
{
    int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 ^ 11, a5 = a0 ^ 13, a6 = a0 + 17, a7 = a0 - 19;
    const int c1 = seed | 1, c2 = seed | 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {            // 8 independent chains of add
                a0 += c1; a1 += c2; a2 += c1; a3 += c2; a4 += c1; a5 += c2; a6 += c1; a7 += c2;
            } else if (MODE == 1) {     // add + max (clamp) pairs
                a0 = max(a0 + c1, -32768); a1 = max(a1 + c2, -32768); a2 = max(a2 + c1, -32768); a3 = max(a3 + c2, -32768);
                a4 = max(a4 + c1, -32768); a5 = max(a5 + c2, -32768); a6 = max(a6 + c1, -32768); a7 = max(a7 + c2, -32768);
            } else if (MODE == 2) {     // compare + select
                a0 = a1 > a2 ? a3 : a0; a1 = a2 > a3 ? a4 : a1; a2 = a3 > a4 ? a5 : a2; a3 = a4 > a5 ? a6 : a3;
                a4 = a5 > a6 ? a7 : a4; a5 = a6 > a7 ? a0 : a5; a6 = a7 > a0 ? a1 : a6; a7 = a0 > a1 ? a2 : a7;
            } else {                    // DPP row_shr moves feeding adds
                a0 += __builtin_amdgcn_mov_dpp(a1, 0x111, 0xf, 0xf, false); a1 += __builtin_amdgcn_mov_dpp(a2, 0x111, 0xf, 0xf, false);
                a2 += __builtin_amdgcn_mov_dpp(a3, 0x111, 0xf, 0xf, false); a3 += __builtin_amdgcn_mov_dpp(a4, 0x111, 0xf, 0xf, false);
                a4 += __builtin_amdgcn_mov_dpp(a5, 0x111, 0xf, 0xf, false); a5 += __builtin_amdgcn_mov_dpp(a6, 0x111, 0xf, 0xf, false);
                a6 += __builtin_amdgcn_mov_dpp(a7, 0x111, 0xf, 0xf, false); a7 += __builtin_amdgcn_mov_dpp(a0, 0x111, 0xf, 0xf, false);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
static void run(const char* name, int ops_per_iter)
{
    int dev = 0; hipDeviceProp_t prop; hipGetDeviceProperties(&prop, dev);
    const int cus = prop.multiProcessorCount;
    int* out; hipMalloc(&out, sizeof(int) * 256 * cus * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {            // waves per SIMD = blocks per CU (256 threads = 4 waves = 1 per SIMD)
        const int grid = cus * wps;
        spin<MODE><<<grid, 256>>>(out, 100, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0); spin<MODE><<<grid, 256>>>(out, iters, 1); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double winst = (double) grid * 4 * iters * ops_per_iter;      // wave-instructions
        const double per_simd_per_s = winst / (cus * 4) / (ms * 1e-3);
        printf("%-14s waves/SIMD %d: %.3f ms, %.2f G wave-instr/s/SIMD (= %.2f per cycle at 2.4 GHz), chip %.1f T lane-ops/s\n",
               name, wps, ms, per_simd_per_s / 1e9, per_simd_per_s / 2.4e9, winst * 64 / (ms * 1e-3) / 1e12);
    }
    hipFree(out);
}

int main()
{
    run<0>("add", 64);
    run<1>("add+max", 128);
    run<2>("cmp+cndmask", 128);
    run<3>("dpp_mov+add", 128);
    return 0;
}

// What does SQ_ACTIVE_INST_VALU count per VALU instruction on gfx950?  (is it 4 cycles per instruction -- then the uint8
// raster kernel keeps its VALU pipes ~89 % busy -- or 2?)  A kernel that is nothing but a dependent chain of v_add_u32, at a chosen occupancy
// (dynamic LDS holds the number of workgroups per CU down), all 64 lanes or only `active` of them:
//   tools/_build/valu_calibrate <workgroups per CU: 3 | 6> <active lanes: 64 | 16>
// run under  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
// (tools/valu_calibrate.sh).  Per instruction:  SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256) add_chain(unsigned *out, int rounds, int active) {
    extern __shared__ unsigned lds[];
    unsigned v = threadIdx.x, w = blockIdx.x;
    if ((int)(threadIdx.x & 63) < active) {
        for (int r = 0; r < rounds; ++r) {
#pragma unroll
            for (int i = 0; i < 256; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(w));
        }
    }
    if (v == 0xdeadbeefu) lds[threadIdx.x] = v;
    out[blockIdx.x * 256 + threadIdx.x] = v;
}

int main(int argc, char **argv) {
    const int per_cu = argc > 1 ? atoi(argv[1]) : 3, active = argc > 2 ? atoi(argv[2]) : 64;
    int cus = 0;
    hipSetDevice(0);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int lds = (160 * 1024 / per_cu) & ~2047;          // so many workgroups of 4 waves per CU = per_cu waves per SIMD
    hipFuncSetAttribute((const void *)add_chain, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    unsigned *out = nullptr;
    hipMalloc((void **)&out, (size_t)cus * per_cu * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int rounds = 2000;
    hipLaunchKernelGGL(add_chain, dim3(cus * per_cu), dim3(256), lds - 64, 0, out, 10, active);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(add_chain, dim3(cus * per_cu), dim3(256), lds - 64, 0, out, rounds, active);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double insts = (double)cus * per_cu * 4 * rounds * 256;          // wave instructions
    printf("%d workgroups per CU (= waves per SIMD), %d active lanes: %.3f ms, %.3e wave instructions of v_add_u32 -> %.2f cycles per instruction and SIMD at 2.4 GHz\n",
           per_cu, active, ms, insts, ms * 1e-3 * 2.4e9 / (insts / (cus * 4)));
    return 0;
}

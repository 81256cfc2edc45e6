// Attainable fp32-MFMA rate on this chip under sustained load (clock included): the ceiling the
// shared-MLP kernels are measured against.  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters)
{
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f + 1.0f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpc : {1, 2, 3}) {                       // workgroups (of 4 waves) per CU
        const int blocks = 256 * wpc, iters = 20000;
        mfma_loop<<<blocks, 256>>>(out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        mfma_loop<<<blocks, 256>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * 4 * (2.0 * 32 * 32 * 2);
        printf("%d workgroup(s)/CU: %.1f TFLOP/s fp32 MFMA (%.2f ms) -> implied clock %.2f GHz\n", wpc,
               flop / ms / 1e9, ms, flop / ms / 1e9 / 157.3 * 2.4);
    }
    return 0;
}

// probe: hand-built buffer descriptor + inline-asm buffer_load_dword vs plain loads
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_srd(const void* p, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)p;
    i32x4 r;
    r.x = (int)(unsigned)a; r.y = (int)((unsigned)(a >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
}
__global__ void k(const float* x, float* y, int P, int K)
{
    const i32x4 s = make_srd(x + (long long)blockIdx.x * K * P, (unsigned)K * P * 4u);
    const int xoff = threadIdx.x * 4;
    float acc = 0.f;
    for (int kk = 0; kk < K; kk += 8) {
        float r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int soff = (kk + i) * P * 4;
            asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(r[i]) : "v"(xoff), "s"(s), "s"(soff));
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += r[i];
    }
    y[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main()
{
    const int P = 256, K = 64, B = 4;
    float *x, *y;
    hipMalloc(&x, sizeof(float) * B * K * P); hipMalloc(&y, sizeof(float) * B * P);
    float* h = (float*)malloc(sizeof(float) * B * K * P);
    for (int i = 0; i < B * K * P; ++i) h[i] = (float)(i % 97);
    hipMemcpy(x, h, sizeof(float) * B * K * P, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(B), dim3(P), 0, 0, x, y, P, K);
    hipError_t e = hipDeviceSynchronize();
    float out[4];
    hipMemcpy(out, y, sizeof(out), hipMemcpyDeviceToHost);
    double want = 0; for (int kk = 0; kk < K; ++kk) want += h[kk * P];
    printf("err=%d got %f want %f\n", (int)e, out[0], want);
    return 0;
}

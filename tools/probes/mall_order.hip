// Does the 256 MB Infinity Cache reward walking a big tensor in the OPPOSITE direction to the kernel before?
// Pass A touches a buffer of `mb` MB front to back (reads it, or writes it with plain / nt stores); pass B reads it front to
// back again or back to front.  If reads/writes allocate in the memory-side cache, the reversed B finds the buffer's tail
// (the last ~256 MB A touched) there.  Times of pass B, median of 15.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mall_order.hip -o tools/probes/mall_order.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));

// chunk c of the grid-stride walk: blocks take chunks of 256 x 4 float4 in ascending (rev = 0) or descending order
template <int MODE>   // 0 read (plain), 1 write plain, 2 write nt, 3 read nt
__global__ __launch_bounds__(256) void pass_kernel(float4* buf, long long n4, int rev, float* sink)
{
    const long long chunks = n4 / 1024;
    float acc = 0.f;
    for (long long c = blockIdx.x; c < chunks; c += gridDim.x) {
        const long long cc = rev ? chunks - 1 - c : c;
        float4* p = buf + cc * 1024 + threadIdx.x;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) { const float4 v = p[u * 256]; acc += v.x + v.y + v.z + v.w; }
            else if (MODE == 3) { const f4v v = __builtin_nontemporal_load((const f4v*)(p + u * 256)); acc += v.x + v.y + v.z + v.w; }
            else if (MODE == 1) p[u * 256] = make_float4(1.f, 2.f, 3.f, (float)u);
            else { f4v v = {1.f, 2.f, 3.f, (float)u}; __builtin_nontemporal_store(v, (f4v*)(p + u * 256)); }
        }
    }
    if (MODE == 0 || MODE == 3) if (acc == 12345.678f) sink[0] = acc;
}

template <int MA, int MB>
static int run(const char* label, float4* buf, long long n4, float* sink, int revB, int grid)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t;
    for (int i = 0; i < 15; ++i) {
        pass_kernel<MA><<<grid, 256>>>(buf, n4, 0, sink);
        CK(hipEventRecord(e0));
        pass_kernel<MB><<<grid, 256>>>(buf, n4, revB, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    const double mb = n4 * 16 / 1e6;
    printf("%-44s B %s: %7.1f us  %6.2f TB/s\n", label, revB ? "back to front" : "front to back", t[7], mb / t[7]);
    return 0;
}

int main()
{
    float* sink; CK(hipMalloc(&sink, 64));
    for (long long mb : {134LL, 268LL, 537LL, 1074LL}) {
        const long long n4 = mb * 1000000 / 16 / 1024 * 1024;
        float4* buf; CK(hipMalloc(&buf, n4 * 16)); CK(hipMemset(buf, 0, n4 * 16));
        printf("-- buffer %lld MB, 2048 workgroups\n", mb);
        for (int rev = 0; rev < 2; ++rev) {
            run<0, 0>("A reads, B reads", buf, n4, sink, rev, 2048);
            run<3, 0>("A reads nt, B reads", buf, n4, sink, rev, 2048);
            run<1, 0>("A writes, B reads", buf, n4, sink, rev, 2048);
            run<2, 0>("A writes nt, B reads", buf, n4, sink, rev, 2048);
            run<0, 3>("A reads, B reads nt", buf, n4, sink, rev, 2048);
        }
        CK(hipFree(buf));
    }
    return 0;
}

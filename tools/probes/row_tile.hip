// How much of the mixed-stream rate (tools/probes/stream_mix.hip: ~5.8 TB/s) survives the ACCESS PATTERN of the
// channel-major layout: an activation is [rows = clouds x channels][P positions]; a workgroup's tile is BP consecutive
// positions of ALL C channel rows of a cloud, i.e. C pieces of BP x 4 bytes that lie P x 4 bytes (0.13-2 MB) apart.
// The kernel reads R such tensors and writes W, tile by tile (persistent workgroups, grid-stride over tiles), with one
// load instruction per lane and row piece: BP = 32 -> a half-wave reads 128 B of a row (two rows per wave instruction),
// 64 -> 256 B (dword per lane), 128 -> 512 B (dwordx2), 256 -> 1 KiB (dwordx4).  U row pieces in flight per stream.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/row_tile.hip -o tools/probes/row_tile.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Ptrs { const float* r[4]; float* w[2]; };

template <int VEC> struct V;
template <> struct V<1> { typedef float T; };
template <> struct V<2> { typedef float2 T; };
template <> struct V<4> { typedef float4 T; };
__device__ __forceinline__ float sum(float v) { return v; }
__device__ __forceinline__ float sum(float2 v) { return v.x + v.y; }
__device__ __forceinline__ float sum(float4 v) { return (v.x + v.y) + (v.z + v.w); }
__device__ __forceinline__ void add(float& a, float b) { a += b; }
__device__ __forceinline__ void add(float2& a, float2 b) { a.x += b.x; a.y += b.y; }
__device__ __forceinline__ void add(float4& a, float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// 256 threads = 4 waves.  BP = 32: lane -> (row parity = lane >> 5, position lane & 31), a wave covers 2 rows per instruction,
// the workgroup 8 rows; BP >= 64: a wave covers 1 row (BP = 64 x VEC positions), the workgroup 4 rows per instruction.
template <int R, int W, int BP, int U>
__global__ __launch_bounds__(256) void tile_kernel(Ptrs p, int C, int P, int nb)
{
    constexpr int VEC = BP <= 64 ? 1 : BP / 64;
    typedef typename V<VEC>::T T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rows_per_instr = BP == 32 ? 8 : 4;
    const int row_in = BP == 32 ? wave * 2 + (lane >> 5) : wave;
    const int pos = BP == 32 ? (lane & 31) : lane * VEC;
    const int tpc = P / BP, total = nb * tpc;
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int b = tile / tpc, p0 = (tile - b * tpc) * BP;
        const long long base = ((long long)b * C + row_in) * P + p0 + pos;
        for (int c0 = 0; c0 < C; c0 += rows_per_instr * U) {
            T v[R][U];
#pragma unroll
            for (int s = 0; s < R; ++s)
#pragma unroll
                for (int u = 0; u < U; ++u)
                    v[s][u] = *reinterpret_cast<const T*>(p.r[s] + base + (long long)(c0 + u * rows_per_instr) * P);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                T a = v[0][u];
#pragma unroll
                for (int s = 1; s < R; ++s) add(a, v[s][u]);
#pragma unroll
                for (int t = 0; t < W; ++t)
                    *reinterpret_cast<T*>(p.w[t] + base + (long long)(c0 + u * rows_per_instr) * P) = a;
                if (W == 0 && sum(a) == 12345.678f) p.w[0][0] = 1.f;
            }
        }
    }
}

template <int R, int W, int BP, int U>
static int run(std::vector<float*>& bufs, int C, int P, int nb, int grid)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Ptrs p[2];
    for (int h = 0; h < 2; ++h) {
        for (int s = 0; s < 4; ++s) p[h].r[s] = bufs[h * 6 + s];
        for (int t = 0; t < 2; ++t) p[h].w[t] = bufs[h * 6 + 4 + t];
    }
    const int reps = 20;
    for (int i = 0; i < 3; ++i) tile_kernel<R, W, BP, U><<<grid, 256>>>(p[i & 1], C, P, nb);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) tile_kernel<R, W, BP, U><<<grid, 256>>>(p[i & 1], C, P, nb);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, mb = (double)(R + W) * nb * C * (double)P * 4 / 1e6;
    printf("R=%d W=%d  C=%3d  BP=%3d (%4d B per row piece)  U=%d  grid=%5d  %7.1f us  %6.2f TB/s\n", R, W, C, BP, BP * 4, U, grid, us, mb / us);
    return 0;
}

int main()
{
    const int nb = 16, P = 32768;                              // the Ball front end: 16 clouds x 512 nodes x 64 neighbours
    std::vector<float*> bufs(12);
    for (auto& b : bufs) { CK(hipMalloc(&b, (size_t)nb * 128 * P * 4)); CK(hipMemset(b, 0, (size_t)nb * 128 * P * 4)); }
#define ROWS(R, W, C) \
    run<R, W, 32, 2>(bufs, C, P, nb, 512); run<R, W, 32, 4>(bufs, C, P, nb, 512); run<R, W, 32, 4>(bufs, C, P, nb, 1024); run<R, W, 32, 8>(bufs, C, P, nb, 512); \
    run<R, W, 64, 4>(bufs, C, P, nb, 512); run<R, W, 64, 8>(bufs, C, P, nb, 512); run<R, W, 64, 8>(bufs, C, P, nb, 1024); \
    run<R, W, 128, 4>(bufs, C, P, nb, 512); run<R, W, 128, 8>(bufs, C, P, nb, 512); \
    run<R, W, 256, 4>(bufs, C, P, nb, 512); run<R, W, 256, 4>(bufs, C, P, nb, 1024); run<R, W, 256, 8>(bufs, C, P, nb, 512); printf("\n");
    ROWS(3, 1, 64)      // layer_bwd_x2<64,64>: dZ, Y, X in, dX out
    ROWS(2, 1, 128)     // the (dZ, Y) pair of a 128-wide layer + 128 rows out (layer_bwd_x2<128,128>: 2 R + dX)
    ROWS(1, 1, 64)      // narrow_fwd 64 -> 64
    ROWS(2, 0, 128)     // pure reads
    return 0;
}

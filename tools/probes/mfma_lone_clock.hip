// Round 6 go / no-go probe for a ONE-wave-per-SIMD form of the f32x2 GEMM stage (csrc/gemm_x2d.hip runs two waves per SIMD,
// 8 accumulator tiles each): the skeleton of a 16-k stage -- NT x 3 MFMAs, R weight-fragment reads (ds_read_b128), V vector
// instructions, L operand loads of LW dwords per lane (two stages ahead), D LDS-DMA pieces of 1 KB, one counted-wait barrier --
// with random operands, every CU busy.  WPS = waves per SIMD (1: 512 registers per wave, 16 accumulator tiles = 64 positions x 256
// channels per wave; 2: today's 8 tiles = 32 positions).  Wall time (HIP events) gives the MFMA rate the chip sustains for that mix.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_lone_clock.hip -o /tmp/mfma_lone_clock && /tmp/mfma_lone_clock
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int v4i32 __attribute__((ext_vector_type(4)));

constexpr int STAGE = 16384, SLOTS = 4;

template <int WPS, int NT, int R, int V, int L, int LW, int D, bool MEM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPS, WPS)))
void lone_loop(unsigned long long* ticks, float* sink, const float* X, unsigned xmask, const uint4* W, int n)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[SLOTS * STAGE];
    constexpr int NM = NT * 3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned seed = (blockIdx.x * 256 + tid) * 2654435761u + 12345u;
    for (int i = tid; i < SLOTS * STAGE / 4; i += 256) {
        seed = seed * 1664525u + 1013904223u;
        reinterpret_cast<unsigned*>(lds)[i] = (seed & 0xbfffbfffu) | 0x30003000u;          // fp16, both signs, |x| in [0.125, 2)
    }
    __syncthreads();
    f16x8 xh, xl;
    for (int i = 0; i < 8; ++i) {
        seed = seed * 1664525u + 1013904223u;
        xh[i] = (_Float16)(((seed >> 8) & 1023) * (1.0f / 512.0f) - 1.0f);
        xl[i] = (_Float16)(((seed >> 18) & 1023) * (1.0f / 512.0f) - 1.0f);
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f16x8 f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = *reinterpret_cast<const f16x8*>(lds + lane * 16 + j * 1024);

    // streamed operand: pairs of waves read the same lines (the second reader of the real kernel hits L2)
    const unsigned gw = (blockIdx.x * 4 + wave) >> 1;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, xmask + 1u, 0x00020000);
    const unsigned long long pw = (unsigned long long)reinterpret_cast<uintptr_t>(W);
    const v4i32 rW = {(int)__builtin_amdgcn_readfirstlane((unsigned)pw), (int)__builtin_amdgcn_readfirstlane((unsigned)(pw >> 32) & 0xffffu),
                      (int)(32 * STAGE), 0x00020000};
    const unsigned ring = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds);
    constexpr int XB = 64 * 4 * LW;                                  // bytes of one load instruction
    float x0[L > 0 ? L * LW : 1], x1[L > 0 ? L * LW : 1];
    auto load1 = [&](int st, float (&S)[L > 0 ? L * LW : 1], int q) {
        if constexpr (MEM && L > 0) {
            const unsigned off = (((gw * (unsigned)n + (unsigned)st) * L + q) * XB) & xmask & ~(unsigned)(XB - 1);
            if constexpr (LW == 1) {
                S[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, lane * 4, off, 0));
            } else {
                const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rX, lane * 8, off, 0));
                S[2 * q] = v[0]; S[2 * q + 1] = v[1];
            }
        }
    };
    auto dma1 = [&](int st, int j) {
        if constexpr (MEM && D > 0) {
            const unsigned dst = ring + (unsigned)((st & (SLOTS - 1)) * STAGE) + (unsigned)((wave * D + j) * 1024);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(dst), "v"(lane * 16), "s"(rW), "s"((st & 31) * STAGE + (wave * D + j) * 1024) : "memory");
        }
    };
    for (int q = 0; q < L; ++q) { x0[q * LW] = 0.f; x1[q * LW] = 0.f; if (LW == 2) { x0[q * LW + 1] = 0.f; x1[q * LW + 1] = 0.f; } }
    for (int q = 0; q < L; ++q) { load1(0, x0, q); load1(1, x1, q); }
    float v0 = 0.5f + lane, v1 = 1.5f, v2 = 0.25f, v3 = 3.0f;
    // loads of this stage issued before the barrier slot + DMA + the previous stage's loads issued after its DMA slots
    constexpr int BSLOT = NM * 3 / 4;
    constexpr int NWAIT = (L * 3 / 4 + D + L * 3 / 4) > 63 ? 63 : (L * 3 / 4 + D + L * 3 / 4);

    auto stage = [&](int st, float (&S)[L > 0 ? L * LW : 1]) {
#pragma unroll
        for (int s = 0; s < NM; ++s) {
            const int t = (s / 6) * 2 + (s & 1);                     // pair-major order of the real kernel: dependent MFMAs two apart
            const f16x8& xx = ((s % 6) / 2 == 1) ? xl : xh;
            // the fragment used was read four reads (>= 6 slots) ago: no LDS latency in front of an MFMA, as in the real schedule
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[t % NT]) : "v"(xx), "v"(f[(R > 0 ? (s * R) / NM + 4 : s) & 7]));
            if ((s * R) / NM != ((s + 1) * R) / NM) {
                const int fr = (s * R) / NM;
                f[fr & 7] = *reinterpret_cast<const f16x8*>(lds + (st & (SLOTS - 1)) * STAGE + lane * 16 + (fr & 15) * 1024);
            }
            if (D > 0 && s >= 4 && s < 4 + D) dma1(st + 2, s - 4);
#pragma unroll
            for (int k = 0; k < (V * (s + 1)) / NM - (V * s) / NM; ++k) {
                const int id = (V * s) / NM + k;
                if ((id & 3) == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v0) : "v"(v1), "v"(v2));
                else if ((id & 3) == 1) asm volatile("v_max_f32 %0, %1, %0" : "+v"(v1) : "v"(v2));
                else if ((id & 3) == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v3) : "v"(v1), "v"(v2));
                else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(v2) : "v"(v0), "v"(v3));
            }
            if (L > 0 && (s * L) / NM != ((s + 1) * L) / NM) {
                const int q = (s * L) / NM;
#pragma unroll
                for (int w = 0; w < LW; ++w) v3 += S[q * LW + w];    // its value is used, then the register is reloaded
                load1(st + 2, S, q);
            }
            if (s == BSLOT) {
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(MEM ? NWAIT : 0) : "memory");
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    dma1(0, 0); for (int j = 1; j < D; ++j) dma1(0, j);
    for (int j = 0; j < D; ++j) dma1(1, j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += 2) {
        stage(i, x0);
        stage(i + 1, x1);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = v0 + v1 + v2 + v3;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        asm volatile("s_nop 7\n\ts_nop 7" : "+a"(acc[k]));
        for (int r = 0; r < 16; ++r) s += acc[k][r];
    }
    for (int j = 0; j < 8; ++j) s += (float)f[j][0];
    if (s == 12345.f) sink[0] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int WPS, int NT, int R, int V, int L, int LW, int D, bool MEM>
void run(const char* what, unsigned long long* d, float* s, const float* X, unsigned xmask, const uint4* W, hipEvent_t e0, hipEvent_t e1)
{
    const int n = 6000 / (NT / 8), blocks = 256 * WPS;
    float ms = 0, best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        lone_loop<WPS, NT, R, V, L, LW, D, MEM><<<blocks, 256>>>(d, s, X, xmask, W, n);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    ms = best;
    unsigned long long h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    const double mf = (double)blocks * 4 * n * NT * 3;
    const double tf = mf * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    // positions per stage and CU: WPS * 4 waves * (NT / 8) * 32
    printf("%-34s waves/SIMD %d  tiles %2d | per stage: %2d MFMA %2d ds_read %3d VALU %2d loads x%d dw %d DMA | %8.1f us  clock %.2f GHz  %7.1f cycles/stage/wave  %5.1f cyc/MFMA/wave  %7.1f TFLOP/s f16 = %.0f fp32-eq | %.3f us per 64 positions of a SIMD\n",
           what, WPS, NT, NT * 3, R, V, L, LW, D, ms * 1e3, h / (ms * 1e6), (double)h / n, (double)h / n / (NT * 3), tf, tf / 3,
           ms * 1e3 / n / (WPS * NT / 16.0));
    fflush(stdout);
}

int main()
{
    unsigned long long* d; float* s; float* X; uint4* W;
    const unsigned xbytes = 1u << 28;
    (void)hipMalloc(&d, 8 * 8192); (void)hipMalloc(&s, 4); (void)hipMalloc(&X, xbytes); (void)hipMalloc(&W, 32 * STAGE);
    (void)hipMemset(X, 0x3c, xbytes); (void)hipMemset(W, 0x3c, 32 * STAGE);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned m = xbytes - 1;
    for (int round = 0; round < 2; ++round) {
        //   WPS NT  R   V   L LW D  MEM
        run<2,  8, 16, 31,  8, 1, 2, false>("today, no memory instructions", d, s, X, m, W, e0, e1);
        run<2,  8, 16, 31,  8, 1, 2, true >("today (x2d forward stage)", d, s, X, m, W, e0, e1);
        run<2,  8, 16, 62, 16, 1, 2, true >("today (x2d data-gradient stage)", d, s, X, m, W, e0, e1);
        run<1, 16, 16, 62,  8, 2, 4, false>("lone 64x256, no memory instr.", d, s, X, m, W, e0, e1);
        run<1, 16, 16, 62,  8, 2, 4, true >("lone 64x256, dwordx2 loads", d, s, X, m, W, e0, e1);
        run<1, 16, 16, 62, 16, 1, 4, true >("lone 64x256, dword loads", d, s, X, m, W, e0, e1);
        run<1, 16, 16, 124, 16, 2, 4, true>("lone 64x256 data gradient, x2", d, s, X, m, W, e0, e1);
        run<1,  8, 16, 31,  8, 1, 4, true >("lone 32x256 (ping-pong form)", d, s, X, m, W, e0, e1);
        run<1, 16,  0,  0,  0, 1, 0, false>("lone, bare MFMAs", d, s, X, m, W, e0, e1);
        run<2,  8,  0,  0,  0, 1, 0, false>("two waves, bare MFMAs", d, s, X, m, W, e0, e1);
    }
    return 0;
}

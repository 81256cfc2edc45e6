// A one-wave monitor kernel that samples (s_memtime = shader cycles, s_memrealtime = 100 MHz) every ~1.5 us while other
// kernels run: the shader clock the chip holds under a given kernel = d cycles / d realtime.  Built as a small shared
// library and driven from Python (tools/clock_under_kernel.py); measurement only, not part of libusip_hip.so.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/probes/clockmon.hip -o tools/probes/libclockmon.so
#include <hip/hip_runtime.h>

__global__ void clockmon_kernel(volatile int* stop, unsigned long long* out, int max_samples, int* count)
{
    int i = 0;
    while (i < max_samples) {
        const unsigned long long c = __builtin_amdgcn_s_memtime();
        const unsigned long long r = __builtin_amdgcn_s_memrealtime();
        out[2 * i] = c;
        out[2 * i + 1] = r;
        ++i;
        if (*stop) break;
        __builtin_amdgcn_s_sleep(32);
        __builtin_amdgcn_s_sleep(32);
    }
    *count = i;
}

static hipStream_t g_stream = nullptr;
static int* g_stop = nullptr;          // host-mapped
static int* g_count = nullptr;         // device

extern "C" int clockmon_start(void* out_device, int max_samples)
{
    if (!g_stream && hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 1;
    if (!g_stop && hipHostMalloc((void**)&g_stop, sizeof(int), hipHostMallocMapped) != hipSuccess) return 2;
    if (!g_count && hipMalloc((void**)&g_count, sizeof(int)) != hipSuccess) return 3;
    *g_stop = 0;
    int* dstop = nullptr;
    if (hipHostGetDevicePointer((void**)&dstop, g_stop, 0) != hipSuccess) return 4;
    hipLaunchKernelGGL(clockmon_kernel, dim3(1), dim3(64), 0, g_stream, dstop, (unsigned long long*)out_device, max_samples, g_count);
    return (int)hipGetLastError();
}

extern "C" int clockmon_stop()
{
    if (!g_stop) return -1;
    *g_stop = 1;
    if (hipStreamSynchronize(g_stream) != hipSuccess) return -2;
    int n = 0;
    if (hipMemcpy(&n, g_count, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -3;
    return n;
}

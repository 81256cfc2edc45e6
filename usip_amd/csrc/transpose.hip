// usip_amd/csrc/transpose.hip -- K-major copies of all shared-MLP weight matrices in ONE launch.
//
// The forward GEMMs take the matrix operand K-major (W^T, [Cin][Cout]).  The reference keeps Conv weights as
// [Cout][Cin][1(,1)] (models/layers.py:186-205, :268-287); transposing each layer's weight where it is used is
// 13 tiny launches per step.  With the parameters laid out in one flat buffer (usip_amd/step.py) a table of
// (source offset, rows, cols, destination offset, first tile) describes all of them and one kernel transposes
// every 32 x 32 tile.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void multi_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                              const int* __restrict__ table, int ntensors)
{
    __shared__ float tile[32][33];
    // table[5*t .. 5*t+4] = (src offset, rows, cols, dst offset, first tile); tiles are numbered tensor by tensor
    int t = 0;
    while (t + 1 < ntensors && (int)blockIdx.x >= table[5 * (t + 1) + 4]) ++t;
    const int so = table[5 * t], rows = table[5 * t + 1], cols = table[5 * t + 2], dof = table[5 * t + 3];
    const int local = blockIdx.x - table[5 * t + 4];
    const int tc = (cols + 31) / 32;
    const int r0 = (local / tc) * 32, c0 = (local % tc) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8 threads
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = src[so + (long long)(r0 + i) * cols + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) dst[dof + (long long)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

}  // namespace

extern "C" int usip_multi_transpose_f32(const float* src, float* dst, const int32_t* table, int ntensors,
                                        int total_tiles, void* stream)
{
    if (ntensors < 0 || total_tiles < 0) return USIP_EINVAL;
    if (ntensors == 0 || total_tiles == 0) return USIP_OK;
    if (!src || !dst || !table) return USIP_EINVAL;
    USIP_LAUNCH(multi_transpose_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, src, dst,
                table, ntensors);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

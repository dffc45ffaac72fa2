// spc_tools.hip -- measurement instruments of the MI355X coupling step (NOT part of the product ABI include/spc.h):
// streaming copies (the measured-bandwidth yardstick bench.py prints beside the 8 TB/s peak; PMC calibration on a known
// byte count) and the many-stream bandwidth probe of tools/bwprobe.py.  Built to tools/libspc_tools.so by
// __graft_entry__.build(); loaded by tools/spc_tools.py.  Round 2 exported these from libspc_hip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "spc_tools.h"

namespace {

constexpr int BLOCK = 256;
thread_local char g_err[256] = "";

int fail(int code, const char *fmt, const char *a = "")
{
    snprintf(g_err, sizeof(g_err), fmt, a);
    return code;
}

int launch_status(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return -3;
}

// 16 B/lane streaming copy: the measured-bandwidth yardstick reported beside the roofline (1:1 read/write;
// a 4x-unrolled non-temporal variant measured no better: 4.4-5.1 vs 5.0-5.1 TB/s read+write).
__global__ __launch_bounds__(BLOCK) void k_copy16(uint4 *dst, const uint4 *src, int64_t n16)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n16; i += (int64_t)gridDim.x * BLOCK)
        dst[i] = src[i];
}

// Bandwidth probes (tools/bwprobe.py): what this box's HBM delivers for a pure read, a pure write and a
// read:write mix in MANY concurrent streams like the coupling kernels' (NS separate arrays advancing together),
// so that the kernels' achieved GB/s can be set against the ceiling of their own access pattern.
// mode 0: copy, 1: read only (sum), 2: write only; NR read streams + NW write streams of `n16` uint4 each.
template <int NR, int NW, int PB = BLOCK>
__global__ __launch_bounds__(PB) void k_probe(uint4 *dst, const uint4 *src, int64_t n16, int64_t stride16, unsigned *sink)
{
    unsigned acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * PB + threadIdx.x; i < n16; i += (int64_t)gridDim.x * PB) {
        uint4 v[NR > 0 ? NR : 1];
#pragma unroll
        for (int r = 0; r < NR; ++r) v[r] = src[r * stride16 + i];
        uint4 w = {1u, 2u, 3u, (unsigned)i};
#pragma unroll
        for (int r = 0; r < NR; ++r) { w.x ^= v[r].x; w.y += v[r].y; w.z ^= v[r].z; w.w += v[r].w; }
#pragma unroll
        for (int q = 0; q < NW; ++q) dst[q * stride16 + i] = w;
        if (NW == 0) acc += w.x + w.y + w.z + w.w;
    }
    if (NW == 0 && acc == 0x12345678u) *sink = acc;      // keeps the loads alive; practically never true
}

// 8 B/lane streaming copy: the access width of the coupling kernels (PMC calibration, tools/pmc_summary.py)
__global__ __launch_bounds__(BLOCK) void k_copy8(double *dst, const double *src, int64_t n8)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n8; i += (int64_t)gridDim.x * BLOCK)
        dst[i] = src[i];
}

// Latency of a DEPENDENT fp64 add chain (what bounds k_vnudge_std: numpy's sequential qt.std sums): one wave, `n` adds
// x = x + c[i & 15] with the addends in registers, result stored so the chain is live.  tools/fp64_chain.py times it.
__global__ __launch_bounds__(64) void k_add_chain(double *out, const double *in, int n)
{
    double c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = in[i];
    double x = in[16 + (threadIdx.x & 15)];
    for (int i = 0; i < n; i += 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = x + c[j];
    }
    out[threadIdx.x] = x;
}

}  // namespace

extern "C" {

int spc_probe_add_chain(void *out, const void *in, int n, void *stream)
{
    if (!out || !in || n <= 0) return fail(-1, "%sadd_chain: bad arguments");
    hipLaunchKernelGGL(k_add_chain, dim3(1), dim3(64), 0, (hipStream_t)stream, (double *)out, (const double *)in, n);
    return launch_status("k_add_chain");
}


const char *spc_tools_last_error(void) { return g_err; }

static unsigned copy_grid(void)
{
    static const unsigned g = [] { const char *e = getenv("SPC_COPY_GRID"); return e ? (unsigned)atoi(e) : 2048u; }();
    return g ? g : 2048u;
}

int spc_stream_copy(void *dst, const void *src, int64_t bytes, void *stream)
{
    if (bytes < 0 || (bytes & 15) || !dst || !src) return fail(-1, "%sstream_copy: bytes must be a multiple of 16, pointers non-NULL");
    if (bytes == 0) return 0;
    hipLaunchKernelGGL(k_copy16, dim3(copy_grid()), dim3(BLOCK), 0, (hipStream_t)stream, (uint4 *)dst, (const uint4 *)src, bytes / 16);
    return launch_status("k_copy16");
}

int spc_stream_probe(int n_read, int n_write, void *dst, const void *src, int64_t bytes_per_stream, int grid, void *stream)
{
    if (bytes_per_stream <= 0 || (bytes_per_stream & 15) || !dst || !src || grid <= 0)
        return fail(-1, "%sstream_probe: bad arguments");
    const int64_t n16 = bytes_per_stream / 16;
    unsigned *sink = (unsigned *)dst;
#define PROBE(NR_, NW_) \
    if (n_read == NR_ && n_write == NW_) { \
        hipLaunchKernelGGL((k_probe<NR_, NW_>), dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, (uint4 *)dst, (const uint4 *)src, n16, n16, sink); \
        return launch_status("k_probe"); \
    }
    PROBE(1, 1) PROBE(1, 0) PROBE(0, 1) PROBE(2, 1) PROBE(14, 7) PROBE(16, 7) PROBE(8, 0) PROBE(0, 7) PROBE(4, 2)
#undef PROBE
    // n_read = 114 / 214: the 14 R + 7 W mix with 512- / 1024-thread workgroups, i.e. 8 KiB / 16 KiB contiguous per
    // stream per workgroup iteration instead of 4 KiB (does the mix ceiling move with the burst length?)
    if (n_read == 114 && n_write == 7) {
        hipLaunchKernelGGL((k_probe<14, 7, 512>), dim3(grid), dim3(512), 0, (hipStream_t)stream, (uint4 *)dst, (const uint4 *)src, n16, n16, sink);
        return launch_status("k_probe");
    }
    if (n_read == 214 && n_write == 7) {
        hipLaunchKernelGGL((k_probe<14, 7, 1024>), dim3(grid), dim3(1024), 0, (hipStream_t)stream, (uint4 *)dst, (const uint4 *)src, n16, n16, sink);
        return launch_status("k_probe");
    }
    return fail(-2, "%sstream_probe: stream mix not instantiated");
}

int spc_stream_copy_f64(void *dst, const void *src, int64_t bytes, void *stream)
{
    if (bytes < 0 || (bytes & 7) || !dst || !src) return fail(-1, "%sstream_copy_f64: bytes must be a multiple of 8, pointers non-NULL");
    if (bytes == 0) return 0;
    hipLaunchKernelGGL(k_copy8, dim3(2048), dim3(BLOCK), 0, (hipStream_t)stream, (double *)dst, (const double *)src, bytes / 8);
    return launch_status("k_copy8");
}

}  // extern "C"

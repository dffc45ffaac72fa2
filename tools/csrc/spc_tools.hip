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

// 4 B/lane streaming copy: the access width of the fp32 arithmetic variant's kernels (is the width itself what they lose to?)
__global__ __launch_bounds__(BLOCK) void k_copy4(float *dst, const float *src, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * BLOCK)
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


// Issue cost of ONE vector instruction (tools/issue_rate.py): 256 threads = one wave per SIMD of one CU, each wave running
// `n` rounds of 8 INDEPENDENT copies of the instruction (8 register sets, so no result is needed before its next use 8
// instructions later); time / (8 n) is what the instruction occupies its SIMD for.  The fp32 arithmetic variant forms its
// quotients through fp64 (spc_hip.hip: div_f32_via_f64): what matters there is what cvt / mul / rcp cost next to v_fma_f32.
#define ISSUE8(ASM, ...)                                                                                   \
    for (int i = 0; i < n; ++i) {                                                                          \
        asm volatile(ASM "\n" ASM##1 "\n" ASM##2 "\n" ASM##3 "\n" ASM##4 "\n" ASM##5 "\n" ASM##6 "\n" ASM##7 : __VA_ARGS__); \
    }
template <int OP> __global__ __launch_bounds__(256) void k_issue(double *out, const double *in, int n)
{
    double d0 = in[threadIdx.x & 15], d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    const double c = in[16], e = in[17];
    float f0 = (float)d0, f1 = (float)d1, f2 = (float)d2, f3 = (float)d3, f4 = (float)d4, f5 = (float)d5, f6 = (float)d6, f7 = (float)d7;
    const float cf = (float)c, ef = (float)e;
    int i0 = threadIdx.x;
#define D8 "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
#define F8 "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
#define DD(op) for (int i = 0; i < n; ++i) asm volatile(op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" \
        op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9" : D8 : "v"(c), "v"(e))
#define DD2(op) for (int i = 0; i < n; ++i) asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" \
        op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8" : D8 : "v"(c))
#define DD1(op) for (int i = 0; i < n; ++i) asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" \
        op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7" : D8)
#define FF(op) for (int i = 0; i < n; ++i) asm volatile(op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" \
        op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9" : F8 : "v"(cf), "v"(ef))
#define FF1(op) for (int i = 0; i < n; ++i) asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" \
        op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7" : F8)
    if constexpr (OP == 0) DD("v_fma_f64");
    if constexpr (OP == 1) DD2("v_mul_f64");
    if constexpr (OP == 2) DD2("v_add_f64");
    if constexpr (OP == 3) DD1("v_rcp_f64");
    if constexpr (OP == 4) DD1("v_rndne_f64");
    if constexpr (OP == 5) DD1("v_frexp_mant_f64");
    if constexpr (OP == 6)
        for (int i = 0; i < n; ++i) asm volatile("v_ldexp_f64 %0, %0, %8\nv_ldexp_f64 %1, %1, %8\nv_ldexp_f64 %2, %2, %8\nv_ldexp_f64 %3, %3, %8\n"
                                                 "v_ldexp_f64 %4, %4, %8\nv_ldexp_f64 %5, %5, %8\nv_ldexp_f64 %6, %6, %8\nv_ldexp_f64 %7, %7, %8" : D8 : "v"(i0 & 1));
    if constexpr (OP == 7) DD("v_div_fixup_f64");
    if constexpr (OP == 8) FF("v_fma_f32");
    if constexpr (OP == 9) FF1("v_rcp_f32");
    if constexpr (OP == 10) FF("v_div_fixup_f32");
    if constexpr (OP == 11)       // v_cvt_f64_f32: f -> d
        for (int i = 0; i < n; ++i) asm volatile("v_cvt_f64_f32 %0, %8\nv_cvt_f64_f32 %1, %9\nv_cvt_f64_f32 %2, %10\nv_cvt_f64_f32 %3, %11\n"
                                                 "v_cvt_f64_f32 %4, %12\nv_cvt_f64_f32 %5, %13\nv_cvt_f64_f32 %6, %14\nv_cvt_f64_f32 %7, %15"
                                                 : D8 : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7));
    if constexpr (OP == 12)       // v_cvt_f32_f64: d -> f
        for (int i = 0; i < n; ++i) asm volatile("v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\n"
                                                 "v_cvt_f32_f64 %4, %12\nv_cvt_f32_f64 %5, %13\nv_cvt_f32_f64 %6, %14\nv_cvt_f32_f64 %7, %15"
                                                 : F8 : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));
    if constexpr (OP == 13)       // v_div_scale_f64 (writes vcc)
        for (int i = 0; i < n; ++i) asm volatile("v_div_scale_f64 %0, vcc, %0, %8, %0\nv_div_scale_f64 %1, vcc, %1, %8, %1\nv_div_scale_f64 %2, vcc, %2, %8, %2\n"
                                                 "v_div_scale_f64 %3, vcc, %3, %8, %3\nv_div_scale_f64 %4, vcc, %4, %8, %4\nv_div_scale_f64 %5, vcc, %5, %8, %5\n"
                                                 "v_div_scale_f64 %6, vcc, %6, %8, %6\nv_div_scale_f64 %7, vcc, %7, %8, %7" : D8 : "v"(c) : "vcc");
    if constexpr (OP == 14)       // v_div_scale_f32
        for (int i = 0; i < n; ++i) asm volatile("v_div_scale_f32 %0, vcc, %0, %8, %0\nv_div_scale_f32 %1, vcc, %1, %8, %1\nv_div_scale_f32 %2, vcc, %2, %8, %2\n"
                                                 "v_div_scale_f32 %3, vcc, %3, %8, %3\nv_div_scale_f32 %4, vcc, %4, %8, %4\nv_div_scale_f32 %5, vcc, %5, %8, %5\n"
                                                 "v_div_scale_f32 %6, vcc, %6, %8, %6\nv_div_scale_f32 %7, vcc, %7, %8, %7" : F8 : "v"(cf) : "vcc");
    if constexpr (OP == 15) FF1("v_exp_f32");
    if constexpr (OP == 16) FF1("v_log_f32");
    if constexpr (OP == 17)       // ds_read_b64 at a lane-own address (LDS issue)
        for (int i = 0; i < n; ++i) asm volatile("v_cvt_f64_i32 %0, %8\nv_cvt_f64_i32 %1, %8\nv_cvt_f64_i32 %2, %8\nv_cvt_f64_i32 %3, %8\n"
                                                 "v_cvt_f64_i32 %4, %8\nv_cvt_f64_i32 %5, %8\nv_cvt_f64_i32 %6, %8\nv_cvt_f64_i32 %7, %8" : D8 : "v"(i0));
    out[threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + (double)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
#undef D8
#undef F8
#undef DD
#undef DD2
#undef DD1
#undef FF
#undef FF1
}
#undef ISSUE8

}  // namespace

extern "C" {

int spc_probe_issue(int op, void *out, const void *in, int n, void *stream)
{
    if (!out || !in || n <= 0) return fail(-1, "%sprobe_issue: bad arguments");
#define ISS(OP_) if (op == OP_) { hipLaunchKernelGGL(k_issue<OP_>, dim3(1), dim3(256), 0, (hipStream_t)stream, (double *)out, (const double *)in, n); return launch_status("k_issue"); }
    ISS(0) ISS(1) ISS(2) ISS(3) ISS(4) ISS(5) ISS(6) ISS(7) ISS(8) ISS(9) ISS(10) ISS(11) ISS(12) ISS(13) ISS(14) ISS(15) ISS(16) ISS(17)
#undef ISS
    return fail(-2, "%sprobe_issue: op not instantiated");
}

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

int spc_stream_copy_f32(void *dst, const void *src, int64_t bytes, int grid, void *stream)
{
    if (bytes < 0 || (bytes & 3) || !dst || !src || grid <= 0) return fail(-1, "%sstream_copy_f32: bytes must be a multiple of 4, pointers non-NULL");
    if (bytes == 0) return 0;
    hipLaunchKernelGGL(k_copy4, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, (float *)dst, (const float *)src, bytes / 4);
    return launch_status("k_copy4");
}

int spc_stream_copy_f64(void *dst, const void *src, int64_t bytes, void *stream)
{
    if (bytes < 0 || (bytes & 7) || !dst || !src) return fail(-1, "%sstream_copy_f64: bytes must be a multiple of 8, pointers non-NULL");
    if (bytes == 0) return 0;
    hipLaunchKernelGGL(k_copy8, dim3(2048), dim3(BLOCK), 0, (hipStream_t)stream, (double *)dst, (const double *)src, bytes / 8);
    return launch_status("k_copy8");
}

}  // extern "C"

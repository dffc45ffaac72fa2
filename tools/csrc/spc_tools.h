/* spc_tools.h -- measurement instruments (tools/libspc_tools.so); not part of the product ABI (include/spc.h).
 * Every function enqueues one kernel on `stream` (hipStream_t as void*) and returns 0 or a negative code
 * (-1 invalid argument, -2 not instantiated, -3 launch error; text: spc_tools_last_error()). */
#ifndef SPC_TOOLS_H
#define SPC_TOOLS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
const char *spc_tools_last_error(void);
/* Measured device-to-device copy rate helper for roofline reporting: copies `bytes` from src to dst with a
 * 16 B/lane streaming kernel. */
int spc_stream_copy(void *dst, const void *src, int64_t bytes, void *stream);
/* The same with 8 B/lane accesses (the access width of the coupling kernels): calibrates the HBM PMC counters on
 * a known byte count in this path's own access pattern. */
int spc_stream_copy_f64(void *dst, const void *src, int64_t bytes, void *stream);
/* The same with 4 B/lane accesses and a caller-chosen grid (tools/copy_width.py). */
int spc_stream_copy_f32(void *dst, const void *src, int64_t bytes, int grid, void *stream);
/* Bandwidth probe (tools/bwprobe.py): n_read read streams and n_write write streams of bytes_per_stream bytes each
 * (stream r at src + r*bytes_per_stream, w at dst + w*bytes_per_stream), 16 B/lane, `grid` workgroups of 256
 * threads.  Instantiated mixes: 1:1, 1:0, 0:1, 2:1, 4:2, 8:0, 0:7, 14:7, 16:7 (114:7 / 214:7 = 14:7 with 512- /
 * 1024-thread workgroups). */
int spc_stream_probe(int n_read, int n_write, void *dst, const void *src, int64_t bytes_per_stream, int grid, void *stream);
/* one wave running `n` dependent fp64 adds (x = x + c); in: >= 32 doubles, out: 64 doubles.  tools/fp64_chain.py */
int spc_probe_add_chain(void *out, const void *in, int n, void *stream);
/* one wave per SIMD of one CU running `n` rounds of 8 independent copies of vector instruction `op` (tools/issue_rate.py
 * holds the table of ops); in: >= 32 doubles, out: 256 doubles */
int spc_probe_issue(int op, void *out, const void *in, int n, void *stream);
#ifdef __cplusplus
}
#endif
#endif

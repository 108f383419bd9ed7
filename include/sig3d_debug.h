/* sig3d_debug.h -- measurement entry points of libsig3d_hip.so (gfx950).  NOT part of the drop-in boundary
 * (include/sig3d_hip.h): nothing of the reference binds these, the training / serving path calls none of them unless a
 * tool asks (situation3d_amd/timeline.py, streams.py with an explicit CU mask, tools/probes/).  Same conventions as
 * sig3d_hip.h: plain C, int status, an explicit hipStream_t as void *. */
#ifndef SIG3D_DEBUG_H
#define SIG3D_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* Store the GPU wall clock into *slot from `stream` (a graph node when captured) -- the concurrent timeline of a
 * replayed hipGraph, which per-node events and profilers cannot give.  sig3d_timestamp_rate: ticks per second. */
int sig3d_timestamp(unsigned long long *slot, void *stream);
int sig3d_timestamp_rate(int device, long long *hz);

/* A stream confined to the CUs of `mask` (bit k = XCD k % 8, CU slot k / 8; `words` 32-bit words), and its release.
 * sig3d_whereami: blocks x threads workgroups report (XCC id, CU id) into slots and hold their CU for hold_us --
 * verifies the mask map and tells whether two streams share a hardware queue. */
int sig3d_stream_create_with_cu_mask(int words, const unsigned int *mask, void **stream);
int sig3d_stream_destroy(void *stream);
int sig3d_whereami(unsigned int *slots, int blocks, int threads, int hold_us, void *stream);

/* blocks x threads workgroups that keep ~vgprs (0 / 100 / 220) registers per lane and lds_bytes of LDS while they
 * sleep for hold_us: the cost of a resident footprint to another stream's kernels.  The
 * 220-register variant needs threads <= 256 (a lane of a larger workgroup has 128). */
int sig3d_hold(float *sink, int blocks, int threads, int hold_us, int vgprs, int lds_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif

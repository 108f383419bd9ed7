/*
 * include/sig3d_hip.h -- C ABI of libsig3d_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary for the SIG3D hot path.  Every entry point is extern "C",
 * takes plain device pointers + sizes + an explicit HIP stream (void* == hipStream_t) and
 * returns an int status (0 == hipSuccess, otherwise the hipError_t value; the message is
 * available from sig3d_last_error()).  No torch / ATen types cross this boundary.
 *
 * Each function replaces one host->kernel wrapper of the reference
 * (YunzeMan/Situation3D, paths relative to the reference root).  Differences from the
 * reference wrappers, all deliberate:
 *   - explicit stream argument (the reference takes at::cuda::getCurrentCUDAStream()
 *     implicitly, e.g. lib/pointnet2/_ext_src/src/ball_query_gpu.cu:49);
 *   - int status return instead of fprintf+exit(-1) (include/cuda_utils.h:30-39);
 *   - outputs that the reference host code zero-initialises (torch::zeros) are
 *     zero-initialised INSIDE these calls (hipMemsetAsync on the same stream), so callers
 *     may pass uninitialised buffers.
 * All pointers are device pointers; float == IEEE binary32, int == int32.
 * All calls are asynchronous on `stream`, re-entrant, and keep no global state.
 */
#ifndef SIG3D_HIP_H
#define SIG3D_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library info ------------------------------------------------------------------ */
const char *sig3d_version(void);          /* "sig3d-hip <semver> gfx950" */
const char *sig3d_last_error(void);       /* thread-local message of the last failing call */

/* Device-side handshake between two streams of one device (no reference counterpart: the reference runs one stream).
 * sig3d_ticket_signal: *ticket += 1 (release, agent scope) as a one-lane kernel on `stream`.
 * sig3d_ticket_wait: a one-lane kernel on `stream` that spins until *ticket has passed *consumed, then sets
 * *consumed += 1; after `timeout_us` it sets *error = 1 and returns anyway.  ticket / consumed / error are device
 * words the caller zeroes once.  Replaces hipStreamWaitEvent where the waiting stream would otherwise hold a blocked
 * barrier packet for milliseconds (see csrc/capi.hip).  A waiter that gave up runs on inputs that may not be staged:
 * the caller must treat a raised error word as fatal (geometry.GeometryPipeline.advance raises).
 * (The measurement entry points -- timestamps, CU-masked streams, resident-footprint probes -- are declared in
 * include/sig3d_debug.h: they are not part of the drop-in boundary.) */
int sig3d_ticket_signal(unsigned int *ticket, void *stream);
int sig3d_ticket_wait(const unsigned int *ticket, unsigned int *consumed, long long timeout_us, int *error,
                      void *stream);
/* one wave that sleeps for hold_us on `stream`: two streams served by ONE hardware queue run two of these in a row,
 * streams on different queues side by side (situation3d_amd/streams.py: run_concurrently, stream_beside). */
int sig3d_queue_hold(int hold_us, void *stream);

/* ---- PointNet++ ops: lib/pointnet2/_ext_src ------------------------------------------ */

/* replaces furthest_point_sampling_kernel_wrapper(b,n,m,dataset,temp,idxs)
 *   lib/pointnet2/_ext_src/src/sampling.cpp:11-13, sampling_gpu.cu:175-229
 * dataset (b,n,3) -> idxs (b,m).  temp (b,n) is scratch; it is (re)initialised here
 * (the reference fills it with 1e10 in sampling.cpp:74-76).  Bit-exact winner selection,
 * including the reference's tie-break order and its `mag <= 1e-3` skip rule. */
int sig3d_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                  int *idxs, void *stream);

/* The same wrapper with a caller-provided workspace instead of `temp` (the reference's wrapper has no way to ask
 * for more than its (b,n) floats: sampling.cpp:74-76): `work` = sig3d_fps_blocks_workspace_bytes(b, n) bytes,
 * 16-byte aligned, contents irrelevant.  Scenes of 8193 .. 196 608 points then run as ONE workgroup per scene over
 * a Morton-ordered copy held in L2 -- only the blocks a new sample can reach are swept -- instead of eight
 * register-resident workgroups exchanging candidates through memory.  Same indices for any input
 * (sampling_gpu.cu:69-173: same distance arithmetic, skip rule and tie order).
 * waves (scenes of up to 65 536 points): waves of that workgroup -- 16: lowest latency (3.7 ms for 2047 rounds over
 * 40 000 points; the cooperative kernel: 3.5), for a call nothing runs beside; 4 (= 0, the default): 6.1 ms alone, and a
 * chain that runs BESIDE other kernels then takes nothing from them (16 waves: +0.27 ms on the training step) -- but
 * every round fetches its operands from L2 / memory, so beside a bandwidth-heavy step the call takes 1.5-2 x as long
 * as alone (9.5-13 ms), where the register-resident cooperative kernel takes 7: DESIGN.md section 4j, why
 * pointnet2._ext and geometry.GeometryPlan keep sig3d_furthest_point_sampling unless SIG3D_FPS_BLOCKS=1. */
long sig3d_fps_blocks_workspace_bytes(int b, int n);
int sig3d_furthest_point_sampling_blocks(int b, int n, int m, const float *dataset, void *work, long work_bytes,
                                         int waves, int *idxs, void *stream);

/* Same wrapper, same results for ANY input, for the call sites whose `dataset` is itself the
 * output of an earlier FPS stored in pick order (SA level l+1 sampling the centres of level l,
 * pointnet2_modules.py:233-240 stacked four times in models/.../pointnet2 backbone).  FPS over
 * an FPS-ordered cloud picks 0,1,..,m-1 unless candidates tie exactly; two parallel kernels
 * PROVE that per scene with the reference's own arithmetic and tie order, and only the scenes
 * that fail the proof run the m dependent rounds.  flags: b ints, 1 = proven, 0 = computed. */
int sig3d_furthest_point_sampling_nested(int b, int n, int m, const float *dataset, float *temp,
                                         int *idxs, int *flags, void *stream);
/* A chain of nested samplings with ONE proof: level l draws m[l] points from the output of level l-1 (level 0 from the
 * n0 points of `dataset`), exactly as sig3d_furthest_point_sampling_nested + sig3d_gather_xyz called level by level
 * (same indices, same centres, for ANY input), but one radius launch and one check launch cover all levels.
 * m: nlevels (1..4) host ints with m[l] <= n[l] <= 8192, m[l] <= 4096; idxs[l] (b, m[l]) int32 and new_xyz[l]
 * (b, m[l], 3) f32: HOST arrays of device pointers; temp (b, n0) f32 scratch; flags (nlevels, b) int32: 1 where
 * level l of a scene was proven (and every level above it). */
int sig3d_fps_nested_chain(int b, int n0, int nlevels, const int *m, const float *dataset, float *temp,
                           int *const *idxs, float *const *new_xyz, int *flags, void *stream);

/* replaces gather_points_kernel_wrapper(b,c,n,npoints,points,idx,out)
 *   sampling.cpp:4-6, sampling_gpu.cu:8-31.   points (b,c,n), idx (b,npoints) -> out (b,c,npoints) */
int sig3d_gather_points(int b, int c, int n, int npoints, const float *points,
                        const int *idx, float *out, void *stream);

/* replaces gather_points_grad_kernel_wrapper(b,c,n,npoints,grad_out,idx,grad_points)
 *   sampling.cpp:7-9, sampling_gpu.cu:34-57.  grad_points (b,c,n) is zeroed here. */
int sig3d_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                             const int *idx, float *grad_points, void *stream);

/* Health of the cooperative FPS (n > 8192: one scene is sampled by 8 workgroups that exchange their
 * per-round candidates through memory and therefore must be co-resident).  A workgroup that waits
 * for a peer longer than 2^22 polls gives up, fills the rest of that scene's indices with -1 and
 * counts the event; the gathers clamp such indices, so nothing reads out of bounds, but the batch is
 * garbage.  The launch itself returned 0 long before, so the host asks here -- SYNCHRONOUSLY (a
 * blocking device read; call it between steps, never inside a capture): *count = events since the
 * library was loaded (or since the last call with reset != 0). */
int sig3d_fps_timeout_count(unsigned *count, int reset);

/* Fuses the centre gather of _PointnetSAModuleBase.forward / PointnetSAModuleVotes.forward
 *   lib/pointnet2/pointnet2_modules.py:52-56, 233-240
 * (xyz.transpose(1,2).contiguous() -> gather_operation -> .transpose(1,2).contiguous()):
 * xyz (b,n,3), idx (b,m) -> out (b,m,3) with out[b,j,:] = xyz[b,idx[b,j],:]. */
int sig3d_gather_xyz(int b, int n, int m, const float *xyz, const int *idx, float *out,
                     void *stream);

/* replaces query_ball_point_kernel_wrapper(b,n,m,radius,nsample,new_xyz,xyz,idx)
 *   ball_query.cpp:4-6, ball_query_gpu.cu:9-54.
 * new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample): first `nsample` points (in index
 * order) with d2 < radius*radius, padded with the first hit; all-zero row when no hit. */
int sig3d_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                     const float *xyz, int *idx, void *stream);

/* Same result, bit for bit, for several problems in ONE launch pair (the levels of a set-abstraction stack).
 * Scenes of more than 4096 points: the CENTRES of a scene are binned into hashed cells of edge 2.02*radius
 * inside LDS and the scene's points are streamed past them once, in index order; every point appends itself to
 * the centres it hits (at most eight buckets to look at), and a second kernel puts each centre's hit list back
 * into index order by rank counting, truncates to nsample, pads with the first hit / writes the zero row;
 * centres with more than 256 hits fall back to the ordered scan.  ~5 distance tests per POINT instead of n per
 * centre.  Scenes of at most 4096 points (deeper levels: a ball holds a tenth of the scene) run the ordered scan
 * of sig3d_ball_query from an LDS copy of the scene, in the same launch.
 * levels: HOST array; radius > 0; m of any size (blocks of 4096 centres, 16 blocks per launch at most).
 * workspace: sig3d_ball_query_levels_workspace_bytes(b, nlevels, levels) bytes (-1: does not fit one launch);
 * at most 4 * (1 + 256) bytes per centre.  Assumes |coordinate| / radius < ~4e4 (cell indexing in f32). */
typedef struct {
  int n, m, nsample;
  float radius;
  const float *xyz;      /* (b, n, 3) */
  const float *new_xyz;  /* (b, m, 3) */
  int *idx;              /* (b, m, nsample) */
} sig3d_bq_level;
long sig3d_ball_query_levels_workspace_bytes(int b, int nlevels, const sig3d_bq_level *levels);
int sig3d_ball_query_levels(int b, int nlevels, const sig3d_bq_level *levels, void *workspace,
                            long workspace_bytes, void *stream);
/* flags: SIG3D_BQ_CLEAN = the caller vouches that `workspace` has been through a completed call with the SAME
 * (b, levels sizes) since it was last written by anyone else: the per-centre counters are then zero (the rank kernel
 * zeroes each one it reads) and the memset in front of the launch pair is skipped.  flags = 0 is always safe. */
#define SIG3D_BQ_CLEAN 1
int sig3d_ball_query_levels_ex(int b, int nlevels, const sig3d_bq_level *levels, void *workspace,
                               long workspace_bytes, int flags, void *stream);
/* The same call, counting its work: *stats (a device word the caller zeroes) += the centre-point distance tests the
 * scatter kernel performed (16 bytes of LDS read and 8 flop each) -- what bench.py's roofline_ball_query prices the
 * kernel against; the neighbour search streams 10 MB and is bound by LDS round trips, not by HBM.  stats may be NULL. */
int sig3d_ball_query_levels_stats(int b, int nlevels, const sig3d_bq_level *levels, void *workspace, long workspace_bytes,
                                  int flags, unsigned long long *stats, void *stream);

/* One problem through sig3d_ball_query_levels (the round-1 name: it was a hashed grid over the points).
 * workspace: 1028 * b * m bytes.  n < 256, radius <= 0 or more than 65536 centres are forwarded to
 * sig3d_ball_query. */
int sig3d_ball_query_grid(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                          const float *xyz, int *idx, void *workspace, long workspace_bytes,
                          void *stream);

/* ---- scene voxelisation / de-duplication (the DataLoader-side numpy path) ----
 * replaces, for a whole ragged batch at once,
 *   lib/sepdataset.py:286-300 (augmentation rotations p <- p.R^T, p <- p - p.min(0)),
 *   lib/openscene/voxelizer_dev.py:35-48 (Voxelizer.voxelize: floor([p,1] @ diag(1/voxel)^T), gathers),
 *   lib/openscene/voxelization_utils.py:9-24,108-140 (fnv_hash_vec, sparse_quantize's np.unique).
 * offsets (b+1) i32 DEVICE array: scene s owns points [offsets[s], offsets[s+1]) of the flat arrays;
 * max_n >= the largest scene, total = offsets[b] (both host values, no device sync).
 * coords (total,3) f32 (coords_f64 == 0) or f64; rot (b,n_rot,9) f64 row-major matrices applied in
 * order (n_rot <= 8, NULL when 0); flips (b) i32 device array or NULL, bit a set = negate axis a first
 * (the mirror augmentation, sepdataset.py:243-262); shift_min: subtract the per-scene minimum (in float32 when the
 * scene is still float32, as numpy would); cell = floor(v * quant[a]) (divide == 0, the voxeliser's
 * matmul) or floor(v / quant[a]) (divide == 1, sparse_quantize), all in float64.
 * -> inds (total) i32: per scene, scene-local index of the FIRST point of every distinct cell in
 *    ascending FNV-key order (np.unique's return_index), first num_unique[s] entries valid;
 *    inverse (total) i32: rank of every point's cell (return_inverse); num_unique (b) i32;
 *    mins (b,3) f64: per-scene minimum of the (rotated) coordinates (always written);
 *    optional (NULL to skip) vox (total,3) i32 cells of the kept points, feats_out (total,c_feat) f32
 *    = feats[inds], labels_out (total) i32 = labels[inds].
 * Bit-identical to the numpy path (stable 8-pass LSD radix sort of the 64-bit keys). */
long sig3d_voxelize_workspace_bytes(int b, long total, int max_n);
int sig3d_voxelize(int b, int max_n, const int *offsets, const void *coords, int coords_f64, int n_rot,
                   const double *rot, const int *flips, int shift_min, int divide, const double *quant, int c_feat,
                   const float *feats, const int *labels, int *inds, int *inverse, int *num_unique,
                   int *vox, float *feats_out, int *labels_out, double *mins, void *workspace,
                   long workspace_bytes, long total, void *stream);
/* voxelization_utils.py:9-24 alone: arr (n,d) i64 cells -> out (n) u64 keys */
int sig3d_fnv_hash_vec(long n, int d, const long *arr, unsigned long long *out, void *stream);

/* replaces group_points_kernel_wrapper(b,c,n,npoints,nsample,points,idx,out)
 *   group_points.cpp:4-6, group_points_gpu.cu:8-39.
 * points (b,c,n), idx (b,npoints,nsample) -> out (b,c,npoints,nsample) */
int sig3d_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                       const int *idx, float *out, void *stream);

/* replaces group_points_grad_kernel_wrapper(b,c,n,npoints,nsample,grad_out,idx,grad_points)
 *   group_points.cpp:8-10, group_points_gpu.cu:43-75.  grad_points (b,c,n) is zeroed here. */
int sig3d_group_points_grad(int b, int c, int n, int npoints, int nsample,
                            const float *grad_out, const int *idx, float *grad_points,
                            void *stream);

/* replaces three_nn_kernel_wrapper(b,n,m,unknown,known,dist2,idx)
 *   interpolate.cpp:4-5, interpolate_gpu.cu:9-68.
 * unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) squared distances ascending, idx (b,n,3) */
int sig3d_three_nn(int b, int n, int m, const float *unknown, const float *known,
                   float *dist2, int *idx, void *stream);

/* replaces three_interpolate_kernel_wrapper(b,c,m,n,points,idx,weight,out)
 *   interpolate.cpp:6-8, interpolate_gpu.cu:72-111.
 * points (b,c,m), idx (b,n,3), weight (b,n,3) -> out (b,c,n) */
int sig3d_three_interpolate(int b, int c, int m, int n, const float *points,
                            const int *idx, const float *weight, float *out, void *stream);

/* replaces three_interpolate_grad_kernel_wrapper(b,c,n,m,grad_out,idx,weight,grad_points)
 *   interpolate.cpp:9-12, interpolate_gpu.cu:116-154.  grad_points (b,c,m) is zeroed here. */
int sig3d_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                 const int *idx, const float *weight, float *grad_points,
                                 void *stream);

/* ---- fused grouping (the path QueryAndGroup.forward takes) ---------------------------- */

/* Fuses lib/pointnet2/pointnet2_utils.py:348-359 (QueryAndGroup.forward after ball_query):
 *   grouped_xyz = group(xyz^T, idx) - new_xyz^T[..., None]  [ / radius if normalize ]
 *   new_features = cat([grouped_xyz, group(features, idx)], dim=1)
 * xyz (b,n,3), new_xyz (b,m,3), features (b,c,n) or NULL (c == 0), idx (b,m,nsample)
 * -> out (b, (use_xyz?3:0)+c, m, nsample).  inv_radius = 1/radius when normalize_xyz else 1
 * (division is performed as `/ radius`, not a multiply, to stay bit-identical: pass radius). */
int sig3d_query_group_fused(int b, int n, int m, int c, int nsample, int use_xyz,
                            int normalize_xyz, float radius, const float *xyz,
                            const float *new_xyz, const float *features, const int *idx,
                            float *out, void *stream);

/* QueryAndGroup's grouping of SEVERAL set-abstraction levels as ONE launch (pointnet2_utils.py:348-359 per level;
 * group_points_gpu.cu:8-28): the neighbour lists of all levels exist once sig3d_ball_query_levels has run, so a stack's
 * four bandwidth-sized launches -- each with a ramp and a tail of its own -- become one grid (largest level first).
 * Each level is exactly sig3d_query_group_fused (point_major == 0: features (b,c,n), nsample % 4 == 0) or
 * sig3d_query_group_fused_pm (point_major != 0: features (b,n,ld)) with the same arguments; same bits out. */
typedef struct sig3d_group_level {
  int n, m, c, ld, nsample;
  int point_major, use_xyz, normalize_xyz;
  float radius;
  const float *xyz, *new_xyz, *features;
  const int *idx;
  float *out;
} sig3d_group_level;
int sig3d_query_group_levels(int b, int nlevels, const sig3d_group_level *levels, void *stream);

/* Same result, bit for bit, from POINT-MAJOR features: features_pm (b,n,ld) holds the c channels of
 * point k at features_pm[(b*n + k)*ld .. +c) (c % 4 == 0, ld % 4 == 0, 16-byte aligned), so a neighbour is
 * one contiguous row instead of c strided 4-byte gathers; tiles of 64 grouped elements x 128 channels are
 * turned through LDS.  sig3d_transpose_cn makes the point-major copy of a (b,c,n) tensor. */
int sig3d_query_group_fused_pm(int b, int n, int m, int c, int ld, int nsample, int use_xyz,
                               int normalize_xyz, float radius, const float *xyz, const float *new_xyz,
                               const float *features_pm, const int *idx, float *out, void *stream);
int sig3d_transpose_cn(int b, int c, int n, const float *in, float *out, void *stream);

/* Backward of the feature half into a POINT-MAJOR gradient: grad_out (b,c_total,m,nsample), window
 * [c_off, c_off+c) -> grad_features_pm (b,n,ld) (zeroed here); sig3d_transpose_cn(b, n, c, ...) turns it
 * back into (b,c,n) when ld == c.  Float atomics: sums agree with sig3d_query_group_fused_grad up to
 * rounding order. */
int sig3d_query_group_fused_grad_pm(int b, int n, int m, int c, int ld, int nsample, int c_total,
                                    int c_off, const float *grad_out, const int *idx,
                                    float *grad_features_pm, void *stream);
/* _z: grad_features_pm arrives ZEROED (a slice of the region a training step clears with one fill,
 * situation3d_amd/scratch.py) and is not cleared here. */
int sig3d_query_group_fused_grad_pm_z(int b, int n, int m, int c, int ld, int nsample, int c_total,
                                      int c_off, const float *grad_out, const int *idx,
                                      float *grad_features_pm, void *stream);

/* Backward of the feature half of sig3d_query_group_fused: grad_out (b,c_total,m,nsample)
 * with channel offset c_off -> grad_features (b,c,n) (zeroed here). */
int sig3d_query_group_fused_grad(int b, int n, int m, int c, int nsample, int c_total,
                                 int c_off, const float *grad_out, const int *idx,
                                 float *grad_features, void *stream);

/* ---- fused SharedMLP (1x1 conv + BatchNorm(train) + ReLU) + neighbourhood max-pool -------- */

/* One layer of pt_utils.SharedMLP (lib/pointnet2/pytorch_utils.py:11-36: Conv2d 1x1 bias=False,
 * then BatchNorm2d, then ReLU) as ONE kernel: y (b,cout,e) = w (cout,cin) . a, where
 * a = x (b,cin,e) when pscale == NULL, else a = relu(x * pscale[ci] + pshift[ci]) -- i.e. the
 * PREVIOUS layer's BatchNorm+ReLU applied on load to its raw conv output.  stat_sum / stat_sq
 * (cout doubles each) receive sum_e y and sum_e y^2 per output channel; both NULL: no statistics
 * (input-gradient use).  `accumulate` (here and in sig3d_channel_stats / sig3d_bn_relu_bwd /
 * sig3d_mlp_layer_dw): 0 = the accumulators (stat_sum/stat_sq, s1/s2, dW) are zeroed by the call;
 * != 0 = the caller zeroed them (e.g. every layer of a stack with one fill) and the call only adds. */
int sig3d_mlp_layer_fwd(int b, int cin, int cout, long e, const float *x, const float *w,
                        const float *pscale, const float *pshift, float *y, double *stat_sum,
                        double *stat_sq, int accumulate, void *stream);

/* Turns the sums into the layer's training-mode BatchNorm affine: scale = gamma/sqrt(var+eps),
 * shift = beta - mean*scale (biased var), saves mean / invstd for the backward pass and updates
 * running_mean / running_var (unbiased) / num_batches_tracked like nn.BatchNorm2d; the three
 * running_* pointers may be NULL. */
int sig3d_bn_finalize(int c, double count, float eps, float momentum, const double *stat_sum,
                      const double *stat_sq, const float *gamma, const float *beta, float *scale,
                      float *shift, float *save_mean, float *save_invstd, float *running_mean,
                      float *running_var, long long *num_batches_tracked, void *stream);

/* Last layer's BatchNorm + ReLU fused with F.max_pool2d(x, [1, nsample])
 * (lib/pointnet2/pointnet2_modules.py:259-262): y (b,c,p,s) raw -> out (b,c,p), arg (b,c,p) =
 * index of the first maximum along s. */
int sig3d_bn_relu_maxpool(int b, int c, int p, int s, const float *y, const float *scale,
                          const float *shift, float *out, int *arg, void *stream);

/* Stand-alone pieces for layers whose 1x1 convolution runs as a library GEMM (small levels, where
 * a few thousand positions do not fill the MFMA kernel): per-channel sum(y), sum(y^2) of y (b,c,e)
 * as doubles (zeroed here; feed sig3d_bn_finalize), and out = relu(y*scale + shift). */
int sig3d_channel_stats(int b, int c, long e, const float *y, double *stat_sum, double *stat_sq,
                        int accumulate, void *stream);
int sig3d_bn_relu_apply(int b, int c, long e, const float *y, const float *scale, const float *shift,
                        float *out, void *stream);

/* Backward of BatchNorm(train)+ReLU of one layer: y (b,c,e) raw conv output, upstream gradient
 * either dense dA (b,c,e) or -- for the last layer -- the max-pool gradient given as
 * dOut (b,c,e/s) + arg (b,c,e/s) (pass dA = NULL).  Produces s1 = sum dZ (= d beta) and
 * s2 = sum dZ*xhat (= d gamma) as doubles (zeroed here) and dY (b,c,e), the gradient w.r.t. the
 * raw conv output.  scale/shift/mean/invstd are the outputs of sig3d_bn_finalize. */
int sig3d_bn_relu_bwd(int b, int c, long e, int s, const float *dA, const float *dOut,
                      const int *arg, const float *y, const float *scale, const float *shift,
                      const float *mean, const float *invstd, double *s1, double *s2, float *dY,
                      int accumulate, void *stream);

/* Weight gradient of one layer: dW (cout,cin) = sum_{b,e} dY[b,co,e] * a[b,ci,e], with a = x or
 * relu(x*pscale + pshift) exactly as in sig3d_mlp_layer_fwd.  dW is zeroed here.  (The input
 * gradient dA = W^T dY is sig3d_mlp_layer_fwd with x = dY and w = W^T.) */
int sig3d_mlp_layer_dw(int b, int cin, int cout, long e, const float *dY, const float *x,
                       const float *pscale, const float *pshift, float *dW, int accumulate, void *stream);

/* ---- situational pose re-encode -------------------------------------------------------- */

/* replaces situation3d/utils/temp.py:42-97 (batch_matrix_function + homogeneous bmm):
 * pose (b,7) = [tx,ty,tz, qx,qy,qz,qw]; points (b,n,3) -> out (b,n,3) = R(q) p + t with
 * R exactly as written in temp.py:63-73 (x2-y2-z2+w2 form, no normalisation).
 * inverse != 0 computes the agent-frame map R^T (p - t) instead (not in the reference). */
int sig3d_situational_transform(int b, int n, const float *pose, const float *points,
                                float *out, int inverse, void *stream);

/* Backward: grad_out (b,n,3) -> grad_points (b,n,3) and grad_pose (b,7) (zeroed here). */
int sig3d_situational_transform_grad(int b, int n, const float *pose, const float *points,
                                     const float *grad_out, float *grad_points,
                                     float *grad_pose, int inverse, void *stream);

/* ---- Blip2T5 pre-stage ------------------------------------------------------------------ */

/* replaces the host-side position-embedding build + add of Blip2T5.forward
 *   3DLLM_BLIP2-base/lavis/models/blip2_models/blip2_t5.py:106-118
 * feat (b,n,c), pc (b,n,3) integer-valued floats, table (trows,tw) -> out (b,n,c):
 * out[...,ch] = feat[...,ch] + scale * table[(long)pc[...,ch/tw]][ch%tw] for ch < 3*tw, else feat.
 * Reference values: c = 1408, tw = 1408/3 = 469, trows = 256, scale = 0.01.  Indices are clamped
 * to [0, trows) (the reference would raise on an out-of-range index). */
int sig3d_pos_embed_add(int b, int n, int c, int tw, int trows, float scale, const float *feat,
                        const float *pc, const float *table, float *out, void *stream);

/* ---- Q-Former dense-layer helpers ------------------------------------------------------- */

/* Bias gradient of an nn.Linear (backward of Qformer.py:242,311,324 `self.dense(...)` and of the
 * query/key/value projections :164-178): out[p][c] = sum_r x[p][r][c], x (parts, rows, cols)
 * row-major, out (parts, cols); parts > 1 serves a batched (strided) pair of layers in one launch.
 * Deterministic (no atomics). */
int sig3d_column_sum(int parts, int rows, int cols, const float *x, float *out, void *stream);
/* Up to SIG3D_COLUMN_SUM_MAX_JOBS such sums in ONE launch (the weight-gradient flush of the Q-Former: the bias gradients
 * torch takes as `grad.sum(0)` per nn.Linear, Qformer.py:116-118, :238, :305, :320, and the folds of the LayerNorm
 * tails' partial rows): job j sums x (parts, rows, cols) over its rows into out (parts, cols). */
#define SIG3D_COLUMN_SUM_MAX_JOBS 8
typedef struct sig3d_column_sum_job {
  const float *x;
  float *out;
  int parts, rows, cols, pad;
} sig3d_column_sum_job;
int sig3d_column_sum_multi(int njobs, const sig3d_column_sum_job *jobs, void *stream);

/* BertIntermediate's bias + erf-GELU (Qformer.py:311-313) on the GEMM output WITHOUT bias:
 *   gy == NULL: out = gelu(x + bias)            gy != NULL: out = gy * gelu'(x + bias)
 * x, gy, out (rows, cols), cols % 4 == 0; bias (parts, cols): rows [p*part_rows, (p+1)*part_rows) use
 * set p (part_rows <= 0: one set). */
int sig3d_bias_gelu(int rows, int cols, int part_rows, const float *x, const float *bias,
                    const float *gy, float *out, void *stream);

/* Tail of BertSelfOutput / BertOutput (Qformer.py:241-246, 323-328) as one row kernel:
 *   out = LayerNorm(dropout(x + bias) + res) * gamma + beta,   x = dense(...) WITHOUT its bias.
 * x, res, out, v (rows, cols) with cols <= 1024 and rows*cols < 2^32; v receives the
 * pre-LayerNorm sum, mean / rstd (rows) the row statistics, mask (rows*64 uint16, required when
 * p_drop > 0) the keep mask: word [r*64 + l] bit i = "column l + 64*i of row r was kept".
 * Dropout bits = hash(*rng_counter, call_id, element index): advance the device counter once per
 * forward pass (sig3d_counter_increment) so that hipGraph replays draw fresh masks.
 * part_rows > 0: bias / gamma / beta are (rows/part_rows, cols) and rows [p*part_rows, (p+1)*part_rows)
 * use parameter set p (the query and text feed-forward tails of a layer in one launch); <= 0: one set.
 * live_rows in (0, rows): rows [live_rows, rows) are PADDING of the two-segment layout -- x, v (and dx in the
 * backward) hold live_rows rows only, out rows (dres rows in the backward) beyond are written as zeros, so
 * that the projections around the tail run on the live rows alone; 0: all rows are live.
 * live_rows < 0: -live_rows rows are live and the others PASS THROUGH: out row = res row (dres row = dy row in
 * the backward) -- the text rows under a cross-attention block (Qformer.py:375-402) without split / cat. */
int sig3d_dropout_add_ln_fwd(int rows, int cols, int part_rows, int live_rows, float p_drop, unsigned call_id,
                             const unsigned *rng_counter, const float *x, const float *bias,
                             const float *res, const float *gamma, const float *beta, float eps,
                             float *out, float *v, float *mean, float *rstd, unsigned short *mask,
                             void *stream);

/* Backward of the above: dy (rows, cols) -> dx (gradient of x, feeds the dense layer's GEMMs),
 * dres (gradient of the residual input) and dparams = [d gamma | d beta | d bias] (3*cols floats,
 * fully written here).  workspace: 3*cols*ceil(rows/4) floats of scratch (per-workgroup partial
 * column sums, folded without atomics: deterministic).  part_rows as in the forward: dparams is then
 * (parts, 3, cols) and part_rows must be a multiple of 4.
 * dparams == NULL: the fold is left to the caller -- the workspace then holds `blocks` partial rows of 3*cols
 * floats, blocks = ceil(ceil(rows / r) / 4) with r = 8 / 2 / 1 rows per wave for rows >= 4096 / >= 2048 / below
 * (halved until part_rows % (4 r) == 0), blocks / parts consecutive rows per part; sig3d_column_sum over
 * the workspaces of many tails folds them in one launch. */
int sig3d_dropout_add_ln_bwd(int rows, int cols, int part_rows, int live_rows, float p_drop, const float *dy, const float *v,
                             const float *mean, const float *rstd, const float *gamma,
                             const unsigned short *mask, float *dx, float *dres, float *dparams,
                             float *workspace, void *stream);

/* The same tails behind a SPLIT dense layer (sig3d_gemm16 with splits > 1): x (dy in the backward) is the sum of
 * slab 0 -- the x / dy argument -- and extra_slabs further (rows, cols) matrices at x_slabs + z*slab_stride, added
 * while they are loaded (no fold launch, no atomics in the GEMM).  Backward: only rows below slab_rows have slabs
 * (slab_rows <= 0: all rows), the others take dy alone -- the padding / pass-through rows a GEMM over the live rows
 * never wrote. */
int sig3d_dropout_add_ln_fwd_slabs(int rows, int cols, int part_rows, int live_rows, float p_drop, unsigned call_id,
                                   const unsigned *rng_counter, const float *x, const float *x_slabs, int extra_slabs,
                                   long slab_stride, const float *bias, const float *res, const float *gamma,
                                   const float *beta, float eps, float *out, float *v, float *mean, float *rstd,
                                   unsigned short *mask, void *stream);
int sig3d_dropout_add_ln_bwd_slabs(int rows, int cols, int part_rows, int live_rows, float p_drop, const float *dy,
                                   const float *dy_slabs, int extra_slabs, long slab_stride, int slab_rows,
                                   const float *v, const float *mean, const float *rstd, const float *gamma,
                                   const unsigned short *mask, float *dx, float *dres, float *dparams,
                                   float *workspace, void *stream);

/* The same fused tails with the MCAN blocks' own normalisation (situation3d/models/mcan_sqa_module.py:57-69:
 * a_2 * (x - mean) / (std + eps) + b_2 with the UNBIASED standard deviation and eps added to the std):
 * SA / SGA compute norm(x + dropout(sublayer(x))) (mcan_sqa_module.py:216-224, 249-261).  rstd receives
 * 1/(std + eps); the backward needs eps again. */
int sig3d_dropout_add_mcan_norm_fwd(int rows, int cols, int part_rows, int live_rows, float p_drop, unsigned call_id,
                                    const unsigned *rng_counter, const float *x, const float *bias,
                                    const float *res, const float *gamma, const float *beta, float eps,
                                    float *out, float *v, float *mean, float *rstd, unsigned short *mask,
                                    void *stream);
int sig3d_dropout_add_mcan_norm_bwd(int rows, int cols, int part_rows, int live_rows, float p_drop, float eps, const float *dy,
                                    const float *v, const float *mean, const float *rstd, const float *gamma,
                                    const unsigned short *mask, float *dx, float *dres, float *dparams,
                                    float *workspace, void *stream);

/* BertEmbeddings.forward (Qformer.py:70-98) as one row kernel each way (csrc/qformer_embed.hip):
 *   x = cat(query_embeds, word[ids] + pos[pos_off + t]) ; out = dropout(LayerNorm(x) * gamma + beta)
 * b scenes, q query tokens, t text tokens, cols <= 1024.  query (.., q, cols) with batch stride query_bstride
 * ELEMENTS (0: one (q, cols) block shared by the batch, i.e. query_tokens.expand(b, -1, -1)); ids (b, t) int64
 * (clamped into the table); word (vocab, cols); pos (pos_rows, cols).
 * seg_rows == 0: out rows in (b, q + t) order.  seg_rows = P > 0: the two-segment row layout, 2P rows
 * [b*q query rows, zero padding | b*t text rows, zero padding].  out / v are (rows, cols), mean / rstd (rows),
 * mask (rows*64 uint16, p_drop > 0) as in sig3d_dropout_add_ln_fwd; dropout bits from (*rng_counter, call_id,
 * element index).
 * Backward: dy (rows, cols) -> dquery ((q, cols) summed over the batch when query_shared, else (b, q, cols)),
 * dpos (pos_rows, cols: every row written), the word rows either ADDED to dword (vocab, cols; zeroed by the caller;
 * float atomics; row pad_id gets nothing, like nn.Embedding(padding_idx)) or, rows_out != NULL, stored as (b*t, cols)
 * rows in (b, t) order for a row exchange; dgamma_dbeta (2*cols); workspace (q + t) * 2 * cols floats. */
int sig3d_qformer_embed_fwd(int b, int q, int t, int cols, int seg_rows, const float *query, long query_bstride,
                            const long long *ids, const float *word, int vocab, const float *pos, int pos_rows,
                            int pos_off, const float *gamma, const float *beta, float eps, float p_drop,
                            unsigned call_id, const unsigned *rng_counter, float *out, float *v, float *mean,
                            float *rstd, unsigned short *mask, void *stream);
int sig3d_qformer_embed_bwd(int b, int q, int t, int cols, int seg_rows, int query_shared, const long long *ids,
                            int vocab, int pos_rows, int pos_off, int pad_id, const float *dy, const float *v,
                            const float *mean, const float *rstd, const float *gamma, const unsigned short *mask,
                            float p_drop, float *dquery, float *dpos, float *dword, float *rows_out,
                            float *dgamma_dbeta, float *workspace, void *stream);
/* out[i] = (1 - mask[i]) * -10000 (Qformer.py:729-731, invert_attention_mask) for a 0/1 mask of
 * kind 0: f32, 1: i64, 2: i32, 3: u8 / bool. */
int sig3d_additive_mask(long n, const void *mask, int kind, float *out, void *stream);

/* *counter += 1 (uint32) on the stream: the per-forward seed of the dropout hash. */
int sig3d_counter_increment(unsigned *counter, void *stream);

/* ---- training loss and localisation target -------------------------------------------------- */

/* The SQA3D loss of lib/loss_helper.py:195-227, 286-300 in one launch:
 *   answer = BCE-with-logits(answer_scores, answer_targets, 'sum') / b          (soft multi-hot targets)
 *   pos / rot = MSE (l1 == 0) or L1 (l1 != 0) of aux_scores vs aux_targets on columns [0,3) / [3,aux_dim), 'mean'
 *   loss = amplify * (situation_w * (pos_w * pos + rot_w * rot) + qa_w * answer)     (amplify = 10, :300)
 * losses[5] = {loss, answer, pos, rot, aux}; d_answer (b, num_answers) and d_aux (b, aux_dim) receive
 * d loss / d scores.  sig3d_sqa_loss_scale multiplies them by the incoming gradient (*upstream) -- the whole
 * backward of the loss. */
int sig3d_sqa_loss(int b, int num_answers, int aux_dim, int l1, const float *answer_scores,
                   const float *answer_targets, const float *aux_scores, const float *aux_targets, float qa_w,
                   float situation_w, float pos_w, float rot_w, float amplify, float *losses, float *d_answer,
                   float *d_aux, void *stream);
int sig3d_sqa_loss_scale(int n_answer, int n_aux, const float *upstream, const float *d_answer, const float *d_aux,
                         float *g_answer, float *g_aux, void *stream);

/* Gaussian localisation target of SIG3D.forward (situation3d/models/sqa_module.py:328-338):
 *   out[b][i] = exp(-|positions[b][i][:2] - pose[b][:2]|^2 / (2 sigma^2)) / sum_i(...)
 * positions (b, t, pdim >= 2), pose (b, pose_dim >= 2), out (b, t). */
int sig3d_gaussian_target(int b, int t, int pdim, float sigma, const float *positions, const float *pose,
                          int pose_dim, float *out, void *stream);

/* ---- optimizer step ---------------------------------------------------------------------- */

/* clip_grad_value_ + AdamW.step() (+ the next zero_grad) of lib/solver.py:618-627 /
 * situation3d/train/train.py:226-238 over FLAT storage: p, g, m, v are n-element arrays (one set
 * per parameter group), `step` a device scalar holding the 1-based step count t (advance it with
 * sig3d_step_increment BEFORE the update of a step).  clip_value <= 0 disables the clamp;
 * zero_grad != 0 writes zeros back to g.  Same update rule as torch.optim.AdamW.
 * lr_device (may be NULL): a device scalar that REPLACES `lr` when given -- the learning rate is
 * then read when the kernel executes, so a captured launch follows the reference's StepLR /
 * MultiStepLR schedule (lib/solver.py:239-247) across hipGraph replays. */
int sig3d_step_increment(float *step, void *stream);
int sig3d_adamw_flat(long n, float *p, float *g, float *m, float *v, const float *step, float lr,
                     const float *lr_device, float beta1, float beta2, float eps, float weight_decay,
                     float clip_value, int zero_grad, void *stream);

/* Same update driven by a device-resident table of `nchunks` records
 *   struct { float *p, *g, *m, *v; long long n; float weight_decay; float pad; }   (48 bytes)
 * one workgroup per record (n <= 65536 recommended): gradients may live wherever autograd
 * allocated them.  A record with n <= 0 is skipped entirely (a parameter that received no gradient
 * is neither decayed nor stepped, like torch.optim.AdamW).  sig3d_gather_table copies g -> m for
 * every record (used to gather scattered gradients into flat storage before a data-parallel
 * all-reduce). */
int sig3d_adamw_table(int nchunks, const void *table, const float *step, float lr,
                      const float *lr_device, float beta1, float beta2, float eps, float clip_value,
                      void *stream);
/* The same launch with at most `max_workgroups` workgroups walking the table: an update issued beside another
 * stream's kernels (a forked hipGraph branch) leaves them CU slots instead of filling the chip. */
int sig3d_adamw_table_bounded(int nchunks, const void *table, const float *step, float lr,
                              const float *lr_device, float beta1, float beta2, float eps, float clip_value,
                              int max_workgroups, void *stream);
int sig3d_gather_table(int nchunks, const void *table, void *stream);

/* ---- Q-Former dense layers -------------------------------------------------------------- */

/* Round 4: the exact-f32 MFMA GEMM the step RUNS on (csrc/gemm16_core.h; round 2's first family, sig3d_gemm /
 * sig3d_gemm_group, left the library in round 5: nothing called it).  Replaces the rocBLAS / hipBLASLt launches behind
 *   Qformer.py:116-118 (query / key / value: x W^T + b), :238 and :320 (the dense halves of BertSelfOutput /
 *   BertOutput), :305-313 (BertIntermediate: dense + erf-GELU) and the input-gradient products dX = dY W of
 *   their backward passes (torch: F.linear / addmm / bmm / mm).
 * For every batch element i < batch (operands advance by their stride_* elements):
 *     C (m x n, row stride ldc)  =  A (m x k, rows k-contiguous) * B  [+ bias]  [epilogue]  [+ addend]
 * bmode 0: B(l,j) = B[j*ldb + l] -- an nn.Linear weight in the forward product (rows k-contiguous);
 * bmode 1: B(l,j) = B[l*ldb + j] -- the same weight in the input-gradient product (rows n-contiguous).
 * bias   : (n) per column, or NULL.       addend : same layout as C, or NULL; may BE C (the residual path's gradient).
 * act    : 0 none;  1 C = gelu_erf(acc + bias), aux (layout of C, may be NULL) receives acc + bias;
 *          2 C = (acc + bias) * gelu_erf'(aux)  (BertIntermediate backward).
 * splits : s >= 1 workgroups share the reduction of a tile; split 0 writes C (with bias / addend), split z >= 1
 *          writes C_slabs + (z-1)*slab_stride (+ i*stride_c, row stride ldc) and the CONSUMER adds the slabs while
 *          it loads them (sig3d_dropout_add_ln_fwd_slabs / _bwd_slabs): no atomics, no zero fill, no fold launch.
 *          act != 0 requires splits == 1.  sig3d_gemm16_splits proposes the count measured best on MI355X.
 * config : 0 = choose; 1 = 64x64 workgroup tiles (8 waves), 2 = 32x64 (4 waves), 3 = 64x128 (8 waves).
 * k % 4 == 0, 16-byte aligned operand rows (n % 4 == 0 for bmode 1); every operand below 2 GB per batch element. */
typedef struct sig3d_gemm16_problem {
  const float *A; int lda; long stride_a;
  const float *B; int ldb; long stride_b;
  float *C; int ldc; long stride_c;
  float *C_slabs; long slab_stride;
  const float *bias; long stride_bias;
  const float *addend;
  float *aux;
  int bmode, batch, m, n, k, act, splits, config;
} sig3d_gemm16_problem;
int sig3d_gemm16(const sig3d_gemm16_problem *problem, void *stream);
int sig3d_gemm16_splits(int bmode, int batch, int m, int n, int k, int act, int config);

/* Round 5: the f32 GEMM on the bf16 matrix cores over operands that arrive SPLIT (csrc/gemmp_core.h).  Replaces the
 * rocBLAS / hipBLASLt launches behind the layer-batched weight gradients dW = dY^T X of the Q-Former's dense layers
 *   Qformer.py:116-118, :238, :305, :320 (torch: bmm / mm in the backward pass)
 * and offers the forward (x W^T) and input-gradient (dY W) forms of the same layers.
 * An f32 matrix X (R rows, C columns, C % 32 == 0) is stored as "chunked planes": x = p1 + p2 + p3 exactly, three bf16
 * terms (round to nearest even, subtract, repeat), laid out [C / 32][row capacity][3][32] bf16 -- the three planes of 32
 * consecutive columns of a row are 192 consecutive bytes, rows follow each other, 32-column chunks are `chunk` elements
 * apart (chunk >= rows * 96).  sig3d_planes_split writes that form (batch matrices; src rows `ld` floats apart).
 * sig3d_gemmp, for every batch element i < batch (operands advance by their stride_* elements):
 *     C (m x n, row stride ldc)  =  A * B  [+ bias]  [epilogue]  [+ addend]     (six bf16 products per f32 product,
 *                                                                  f32 accumulation: f32-equivalent, not reduced)
 * modes 0: A = planes of x (m, k), B = planes of W (n, k)           y = x W^T        k % 32 == 0
 *       1: A = planes of dY (m, k), B = planes of W (k, n)          dX = dY W        k % 32 == 0, n % 32 == 0
 *       2: A = planes of dY (k, m), B = planes of X (k, n)          dW = dY^T X      m % 32 == 0, n % 32 == 0, any k
 * (a matrix is stored ONCE: its planes serve with its columns or with its rows as the reduction index).
 * bytes_a / bytes_b: how many bytes may be read from A / B of one batch element (requests beyond return zeros; < 4 GB).
 * bias / addend / act / aux: as sig3d_gemm16.  C may be NULL when only C_planes (chunked planes of the result after
 * the epilogue, chunk stride chunk_c, batch stride stride_cp) is wanted.
 * splits > 1: that many workgroups share a tile's reduction; each parks its partial tile in `work`
 * (sig3d_gemmp_work_floats floats), the last to arrive adds them in split order and runs the epilogue; `counters` (one
 * unsigned per tile, zero before the first launch, left zero) order them.  config: 0 choose, 1 = 64 x 64 tiles,
 * 2 = 64 x 128, 3 = 128 x 128. */
int sig3d_planes_split(int batch, int rows, int cols, const float *src, int ld, long src_stride, void *planes,
                       long chunk_stride, long planes_stride, void *stream);
typedef struct sig3d_gemmp_problem {
  const void *A; long chunk_a; long stride_a; long bytes_a;
  const void *B; long chunk_b; long stride_b; long bytes_b;
  float *C; int ldc; long stride_c;
  void *C_planes; long chunk_c; long stride_cp;
  const float *bias; long stride_bias;
  const float *addend;
  float *aux;
  float *work; unsigned *counters;
  int modes, batch, m, n, k, act, splits, config;
} sig3d_gemmp_problem;
int sig3d_gemmp(const sig3d_gemmp_problem *problem, void *stream);
long sig3d_gemmp_work_floats(int batch, int m, int n, int splits, int config);

/* ---- Q-Former attention ---------------------------------------------------------------- */

/* replaces BertSelfAttention.forward's core
 *   3DLLM_BLIP2-base/lavis/models/blip2_models/Qformer.py:185-223
 * context = softmax(Q K^T * scale + mask) V  per (batch, head), fp32, exact-f32 MFMA.
 * q (b,nq,h*d), k (b,nk,h*d), v (b,nk,h*d) contiguous and TOKEN-MAJOR, i.e. exactly what the
 * query/key/value nn.Linear layers produce (Qformer.py:164-176) -- the transpose_for_scores
 * permute (:140-147) is folded into the kernel's addressing; mask additive (b,nk) or NULL
 * (the reference's (B,1,1,Nk) extended mask, Qformer.py:700-732); d is 64 (Q-Former, 768/12) or 96
 * (the MCAN blocks of the native SIG3D head, mcan_sqa_module.py:113-126: 768/8).  The backward pass runs
 * in chunks of 128 (d = 64) / 64 (d = 96) query rows, dK / dV then meet through float atomics.
 * out (b,nq,h*d) -- already in the permuted "context_layer" layout of Qformer.py:225-227.
 * lse (b,h,nq) receives log-sum-exp rows for the backward pass (may be NULL).
 * ldq / ldk / ldv: row strides in floats of q, k, v (h*d for dense tensors; 3*h*d when q, k, v are
 * column slices of ONE fused QKV projection output, which saves two GEMM launches per attention).
 * In the backward pass dq / dk / dv use the same strides as their inputs.
 * Token -> storage row map of the query-side tensors (q, out, grad_out, dq: q_seg, q_base2, q_rows)
 * and of the key-side tensors (k, v, dk, dv: k_seg, k_base2, k_rows):
 *   seg == n (or 0): plain (b, n) order, row = bi*n + i;
 *   seg <  n       : two segments -- the first `seg` tokens of every batch element are stored first
 *                    (row = bi*seg + i), the remaining n-seg tokens of every batch element from row
 *                    `base2` on (row = base2 + bi*(n-seg) + i-seg; base2 >= b*seg).
 *   rows > 0       : the tensors have `rows` storage rows; every row that holds no token (the gap
 *                    before base2, the tail) is ZERO-FILLED in out (forward) and dq / dk / dv (backward).
 * The Q-Former keeps [query tokens | text tokens] that way, both segments padded to a common row count,
 * so that the per-part feed-forward blocks of BertLayer.forward (Qformer.py:375-405) are contiguous row
 * ranges AND batch into one strided GEMM.  mask (b,nk) and lse (b,h,nq) are always indexed by
 * (batch, token). */
int sig3d_attention_fwd(int b, int h, int nq, int nk, int d, int q_seg, int k_seg, int q_base2,
                        int k_base2, int q_rows, int k_rows, int ldq, int ldk, int ldv, float scale,
                        const float *q,
                        const float *k, const float *v, const float *mask, float *out,
                        float *lse, float p_drop, unsigned call_id, const unsigned *rng_counter,
                        int key_splits, float *workspace, void *stream);
/* key_splits > 1 (few queries, many keys -- the 3D-LLM shapes of 5000..80000 scene tokens): the key
 * range is cut into key_splits pieces so that the launch covers the chip, and a second kernel folds
 * the pieces; workspace = b*h*roundup32(nq)*key_splits*(d+2) floats.  key_splits <= 1: one pass, workspace
 * may be NULL. */
/* p_drop > 0: attention-probability dropout (Qformer.py:219).  The keep bit of element
 * (b, head, query, key) is hash(*rng_counter, call_id, index), identical in forward and backward;
 * advance the device counter once per forward pass (sig3d_counter_increment).  p_drop == 0 is the
 * eval-mode / deterministic path (rng_counter may be NULL). */

/* Backward of sig3d_attention_fwd.  grad_out (b,nq,h*d); out/lse from the forward.
 * -> dq (b,nq,h*d), dk (b,nk,h*d), dv (b,nk,h*d), token-major like the inputs. */
int sig3d_attention_bwd(int b, int h, int nq, int nk, int d, int q_seg, int k_seg, int q_base2,
                        int k_base2, int q_rows, int k_rows, int ldq, int ldk, int ldv, float scale,
                        const float *q,
                        const float *k, const float *v, const float *mask,
                        const float *out, const float *lse, const float *grad_out,
                        float *dq, float *dk, float *dv, float p_drop, unsigned call_id,
                        const unsigned *rng_counter, void *stream);
/* _z: dq arrives ZEROED (few queries, many keys: the key range is split over workgroups and dq is accumulated with
 * float atomics; sig3d_attention_bwd clears it with a memset of its own, this one does not). */
int sig3d_attention_bwd_z(int b, int h, int nq, int nk, int d, int q_seg, int k_seg, int q_base2,
                          int k_base2, int q_rows, int k_rows, int ldq, int ldk, int ldv, float scale,
                          const float *q,
                          const float *k, const float *v, const float *mask,
                          const float *out, const float *lse, const float *grad_out,
                          float *dq, float *dk, float *dv, float p_drop, unsigned call_id,
                          const unsigned *rng_counter, void *stream);

/* ---- small dense layers around the Q-Former as single launches (csrc/small_mlp.hip) ----------------------
 * The positional MLP of the scene tokens (situation3d/models/sqa_module.py:274-278, applied at :319-321):
 *   out (rows,cout) = residual + Linear(hid,cout)(GELU_erf(Linear(cin,hid)(x)))      cin <= 4, hid <= 128 (4 | hid)
 * x (rows,cin), w1 (hid,cin), b1 (hid), w2 (cout,hid), b2 (cout), residual (rows,cout) or NULL;
 * pre (rows,hid) receives the first layer's pre-activation (kept for the backward pass).
 * _bwd: dy (rows,cout) -> dpre (rows,hid) = (dy w2) * gelu'(pre) [dX = dpre w1 is the caller's, when needed],
 * grads = ONE buffer [dw1 (hid*cin) | db1 (hid) | dw2 (cout*hid) | db2 (cout)]: OVERWRITTEN (zeroed here with one
 * memset, filled with float atomics).  The residual's gradient is dy itself. */
int sig3d_pos_mlp_fwd(int rows, int cin, int hid, int cout, const float *x, const float *w1, const float *b1,
                      const float *w2, const float *b2, const float *residual, float *pre, float *out, void *stream);
/* _posed: the situational re-encode of the token positions (situation3d/utils/temp.py:86-97; sig3d_situational_transform)
 * folded into the launch: x (b*tokens, 3) = R(q_b)^T (points - t_b) [inverse != 0] or R(q_b) points + t_b, formed from
 * pose (b,7) and points (b*tokens,3) with the transform's own arithmetic, written to x_out and fed to the MLP. */
int sig3d_pos_mlp_fwd_posed(int b, int tokens, int hid, int cout, int inverse, const float *pose, const float *points,
                            float *x_out, const float *w1, const float *b1, const float *w2, const float *b2,
                            const float *residual, float *pre, float *out, void *stream);
int sig3d_pos_mlp_bwd(int rows, int cin, int hid, int cout, const float *x, const float *w2, const float *pre,
                      const float *dy, float *dpre, float *grads, void *stream);
/* _z: grads arrives ZEROED and is accumulated into (no memset here). */
int sig3d_pos_mlp_bwd_z(int rows, int cin, int hid, int cout, const float *x, const float *w2, const float *pre,
                        const float *dy, float *dpre, float *grads, void *stream);

/* Weight gradient of a SharedMLP layer as a k-streaming split product on the f32 matrix cores (csrc/gemm16.hip over
 * gemm16_core.h): dW (cout, cin) = sum_{b, e} dY[b, co, e] * a[b, ci, e], a = x or relu(x * pscale + pshift) as in
 * sig3d_mlp_layer_dw, both operands read along their rows.  n_act: compact lists (sample i reduces over its first
 * n_act[i] positions; e stays the row stride), or NULL.  dW is OVERWRITTEN; every (sample, split) pair writes a slab of
 * `work` (sig3d_mlp_layer_dw_stream_work_floats floats) and one small launch folds them -- fixed order, no atomics.
 * e % 4 == 0, 16-byte aligned operands; cout * cin % 4 == 0.  1.5-2.5 x faster than sig3d_mlp_layer_dw at the step's
 * shapes. */
int sig3d_mlp_layer_dw_stream(int b, int cin, int cout, long e, const float *dY, const float *x,
                              const float *pscale, const float *pshift, const int *n_act, float *dW,
                              float *work, void *stream);
long sig3d_mlp_layer_dw_stream_work_floats(int b, int cin, int cout, long e);
/* The same product without its fold -- dW holds the first slab, `work` the others (work_floats / roundup4(cout * cin) of
 * them, roundup4(cout * cin) floats apart) -- and the fold of several such results in ONE launch: dst[0 .. n) += the sum
 * of nslabs slabs, slab_stride floats apart, added in the order sig3d_mlp_layer_dw_stream adds them. */
int sig3d_mlp_layer_dw_stream_nofold(int b, int cin, int cout, long e, const float *dY, const float *x,
                                     const float *pscale, const float *pshift, const int *n_act, float *dW,
                                     float *work, void *stream);
/* A compact level's layer in the backward pass, both products that read its dY in ONE launch (two workgroup ranges): the
 * weight gradient as sig3d_mlp_layer_dw_stream_nofold and the input gradient dA (b, cin, e) = W^T dY as sig3d_mlp_layer_dx
 * (w (cout, cin) as stored).  Same results bit for bit; shapes that take other kernel instances run the two launches. */
int sig3d_mlp_layer_dw_dx(int b, int cin, int cout, long e, const float *dY, const float *x, const float *pscale,
                          const float *pshift, const int *n_act, const float *w, float *dW, float *work, float *dA,
                          void *stream);
#define SIG3D_SUM_SLABS_MAX_JOBS 8
typedef struct sig3d_sum_slabs_job {
  float *dst;
  const float *slabs;
  long n, slab_stride;
  int nslabs, pad;
} sig3d_sum_slabs_job;
/* cvt_n > 0: the same launch also converts cvt_n doubles to floats (cvt_dst[i] = (float)cvt_src[i]: the stack's f64
 * BatchNorm-gradient sums -> its f32 d gamma / d beta). */
int sig3d_sum_slabs_multi(int njobs, const sig3d_sum_slabs_job *jobs, const double *cvt_src, float *cvt_dst, int cvt_n,
                          void *stream);

/* ---- the two MLP heads on the pooled Q-Former output (csrc/heads.hip) ----------------------------------------
 * situation3d/models/sqa_module.py: the fused query tokens are averaged per sample and feed
 *   aux    (b, n_aux) = Linear(hidden, n_aux)(GELU(Linear(hidden, hidden)(pooled)))             [w1a b1a w2a b2a]
 *   answer (b, n_ans) = Linear(hidden, n_ans)(Dropout(GELU(Linear(hidden, hidden)(pooled))))    [w1c b1c w2c b2c]
 * b <= 16 samples, hidden % 16 == 0, hidden <= 1024.  rows (b * q, hidden): the q query rows of sample i are rows
 * [i*q, (i+1)*q).  Outputs of the forward that the backward needs: pooled (b, hidden), pre (2, b, hidden) first-layer
 * pre-activations [aux | answer], h (2, b, hidden) second-layer inputs (the answer head's after dropout).
 * Dropout keep bits: hash(*rng_counter, call_id, element), as in sig3d_dropout_add_ln_fwd; p_drop == 0 in eval mode.
 * _bwd: daux (b, n_aux), dans (b, n_ans) -> grads = ONE buffer
 *   [dw1a (hidden^2) | db1a (hidden) | dw2a (n_aux*hidden) | db2a (n_aux) | dw1c | db1c | dw2c (n_ans*hidden) | db2c]
 * and drows (b * q, hidden) = d pooled / q in every query row; all OVERWRITTEN; work: sig3d_pooled_heads_work_floats(b,
 * hidden) floats of scratch (partial input gradients of the four layers).  n_aux, n_ans <= 1024.
 * Three launches forward, three backward; sums in fixed orders (no atomics). */
int sig3d_pooled_heads_fwd(int b, int q, int hidden, int n_aux, int n_ans, const float *rows,
                           const float *w1a, const float *b1a, const float *w2a, const float *b2a,
                           const float *w1c, const float *b1c, const float *w2c, const float *b2c,
                           float p_drop, unsigned call_id, const unsigned *rng_counter, float *pooled,
                           float *pre, float *h, float *aux, float *ans, void *stream);
int sig3d_pooled_heads_bwd(int b, int q, int hidden, int n_aux, int n_ans, const float *daux, const float *dans,
                           const float *pooled, const float *pre, const float *h, const float *w1a,
                           const float *w2a, const float *w1c, const float *w2c, float p_drop, unsigned call_id,
                           const unsigned *rng_counter, float *work, float *grads, float *drows, void *stream);
long sig3d_pooled_heads_work_floats(int b, int hidden);

/* ---- compact mode: set abstraction over the DISTINCT neighbours only ------------------------------------
 * ball_query pads a short list by repeating its first hit (ball_query_gpu.cu:30-40); every padded entry is
 * an identical column of the grouped tensor, of each SharedMLP layer above it (pytorch_utils.py:11-36) and of
 * the max-pool (pointnet2_modules.py:251-262).  sig3d_compact_neighbour_lists lists the distinct
 * (centre, neighbour) pairs of every batch element back to back:
 *   idx (b,m,nsample) -> cidx (b,m*nsample) point index, centre_of (b,m*nsample) centre, mult (b,m*nsample) f32
 *   how many equal columns the entry stands for, seg_off (b,m+1) first entry of every centre, n_act (b) count;
 *   only the first n_act[b] entries of a row are defined.
 * The *_compact entry points below are the dense ones restricted to positions [0, n_act[b]) of every
 * (batch, channel) row (row stride e = m*nsample unchanged); batch statistics are weighted by mult and the
 * BatchNorm backward correction terms carry mult, so results equal the dense ones up to summation order.
 * Gradients of a position are the SUM over the columns it stands for. */
int sig3d_compact_neighbour_lists(int b, int m, int nsample, const int *idx, int *cidx, int *centre_of,
                                  float *mult, int *seg_off, int *n_act, void *stream);
/* grouped tensor of the distinct neighbours: features_pm (b,n,ld) point-major (wide levels) or NULL, then
 * features (b,c,n) channel-major is gathered; out (b, (use_xyz?3:0)+c, m*nsample) */
int sig3d_query_group_compact(int b, int n, int m, int c, int ld, int nsample, int use_xyz, int normalize_xyz,
                              float radius, const float *xyz, const float *new_xyz, const float *features,
                              const float *features_pm, const int *cidx, const int *centre_of,
                              const int *n_act, float *out, void *stream);
/* SURVEY.md 8(f) rank 1 -- the first SharedMLP layer of a set-abstraction level WITHOUT the grouped tensor:
 * replaces QueryAndGroup (pointnet2_utils.py:348-359: grouping_operation x 2, centre subtraction, / radius,
 * concat) + the first Conv2d of the stack (pytorch_utils.py:11-36, pointnet2_modules.py:242-259).  The MFMA
 * operand is gathered while it is loaded: c feature channels from the point-major copy features_pm (b, n, c)
 * at row idx[b][e], 3 channels xyz[idx] - new_xyz[centre] (/ radius when normalize_xyz), in the reference's
 * channel order for the weight w (cout, 3 + c).  y (b, cout, m*nsample) and the BatchNorm statistics as
 * sig3d_mlp_layer_fwd.  c must be a multiple of 32 (narrower levels keep the stored tensor: 29 MB at SA1).
 * Compact lists (csrc/compact.hip): idx = the compact lists, centre_of / n_act / mult as the *_compact calls. */
int sig3d_mlp_layer0_gather_fwd(int b, int n, int m, int nsample, int c, int cout, int normalize_xyz,
                                float radius, const float *xyz, const float *new_xyz,
                                const float *features_pm, const int *idx, const float *w, float *y,
                                double *stat_sum, double *stat_sq, int accumulate, const int *centre_of,
                                const int *n_act, const float *mult, void *stream);

/* Backward of the gathering first layer (same operands):
 *   _gather_dw : dW (cout, 3 + c) [+]= sum_{b,e} dY[b,:,e] X[b,:,e]^T with X gathered on load (dY (b, cout, e));
 *   _scatter_dx: grad_features_pm (b, n, c) += the feature rows of W^T dY, added at the neighbours' rows in the
 *                epilogue of the product (wt = W^T, (3 + c, cout); the caller zeroes the target; the xyz rows
 *                carry no gradient, pointnet2_utils.py:334).  Replaces grouping_operation's backward
 *                (group_points_gpu.cu:43-75) and the (b, 3 + c, npoint, nsample) gradient tensor. */
int sig3d_mlp_layer0_gather_dw(int b, int n, int m, int nsample, int c, int cout, int normalize_xyz,
                               float radius, const float *xyz, const float *new_xyz,
                               const float *features_pm, const int *idx, const float *dY, float *dW,
                               int accumulate, const int *centre_of, const int *n_act, void *stream);
int sig3d_mlp_layer0_scatter_dx(int b, int n, int m, int nsample, int c, int cout, const int *idx,
                                const float *dY, const float *wt, float *grad_features_pm,
                                const int *n_act, void *stream);
/* The same with the forward layer's weight AS STORED, w (cout, 3 + c): staged transposed inside the kernel. */
int sig3d_mlp_layer0_scatter_dx_w(int b, int n, int m, int nsample, int c, int cout, const int *idx,
                                const float *dY, const float *w, float *grad_features_pm,
                                const int *n_act, void *stream);
/* Input gradient of a SharedMLP layer (pytorch_utils.py:11-36: Conv2d 1x1 backward): dA (b, cin, e) = W^T dY with
 * W (cout, cin) as the forward layer stores it -- the weight tile is staged transposed, no W^T copy.  n_act: the
 * compact lists' live counts (first n_act[b] positions of every row), or NULL for dense rows. */
int sig3d_mlp_layer_dx(int b, int cin, int cout, long e, const float *dY, const float *w, float *dA,
                       const int *n_act, void *stream);

/* point_major != 0: grad (b,n,ld); else grad (b,c,n); zeroed here */
int sig3d_query_group_compact_grad(int b, int n, int m, int c, int ld, int nsample, int c_total, int c_off,
                                   const float *grad_out, const int *cidx, const int *n_act, int point_major,
                                   float *grad, void *stream);
int sig3d_mlp_layer_fwd_compact(int b, int cin, int cout, long e, const float *x, const float *w,
                                const float *pscale, const float *pshift, float *y, double *stat_sum,
                                double *stat_sq, int accumulate, const int *n_act, const float *mult,
                                void *stream);
int sig3d_mlp_layer_dw_compact(int b, int cin, int cout, long e, const float *dY, const float *x,
                               const float *pscale, const float *pshift, float *dW, int accumulate,
                               const int *n_act, void *stream);
/* arg = offset of the first maximum inside the centre's segment */
int sig3d_bn_relu_maxpool_compact(int b, int c, int p, long e, const float *y, const float *scale,
                                  const float *shift, const int *seg_off, float *out, int *arg, void *stream);
/* The FIRST SharedMLP layer of a level whose neighbours come straight from the raw scan (SA1): QueryAndGroup's
 * column -- (xyz[k] - new_xyz[j]) (/ radius), then the point's features, pointnet2_utils.py:348-359 -- formed in
 * registers from POINT-MAJOR rows points_pm (b, n, cpt) = [x y z f0 f1 ...] (the layout a scan arrives in) and
 * multiplied by w (cout, cin), cin = 3 + features used <= 8, cout a multiple of 64 (Conv2d 1x1 without bias,
 * pytorch_utils.py:11-36): y (b, cout, m*nsample) raw + per-channel batch statistics (doubles; zeroed here unless
 * accumulate; NULL: none).  No grouped tensor is written.  idx (b, m*nsample): ball-query lists (centre_of = n_act =
 * mult = NULL) or the compact lists of sig3d_compact_neighbour_lists (statistics weighted by mult, positions
 * [0, n_act[b]) only).  _dw: dW (cout, cin) (+)= sum over positions of dY[:, u] column(u)^T, the column gathered
 * again (recompute in backward). */
int sig3d_sa_first_layer_fwd(int b, int n, int m, int nsample, int cpt, int cin, int cout, int normalize_xyz,
                             float radius, const float *points_pm, const float *new_xyz, const int *idx,
                             const int *centre_of, const int *n_act, const float *mult, const float *w, float *y,
                             double *stat_sum, double *stat_sq, int accumulate, void *stream);
int sig3d_sa_first_layer_dw(int b, int n, int m, int nsample, int cpt, int cin, int cout, int normalize_xyz,
                            float radius, const float *points_pm, const float *new_xyz, const int *idx,
                            const int *centre_of, const int *n_act, const float *dY, float *dW, int accumulate,
                            void *stream);

/* sig3d_bn_relu_maxpool / _compact (seg_off != NULL: compact lists, e = row stride, s ignored) with a second,
 * POINT-MAJOR copy of the pooled features: out_pm (b,p,c) -- what the next level's gathers and the Q-Former's
 * scene tokens read (pointnet2_modules.py:259-262 followed by the `.transpose(1, 2)` of its callers), written
 * from the pooling pass instead of by a transpose launch. */
int sig3d_bn_relu_maxpool_pm(int b, int c, int p, int s, long e, const float *y, const float *scale,
                             const float *shift, const int *seg_off, float *out, int *arg, float *out_pm,
                             void *stream);
/* The max-pool gradient arriving point-major, dout_pm (b,p,c) (the twin sig3d_bn_relu_maxpool_pm handed on): one
 * launch writes its channel-major copy dout_cm (b,c,p) AND the top layer's BatchNorm-backward statistics
 * s1 = sum dZ, s2 = sum dZ*xhat (zeroed here unless accumulate).  Follow with sig3d_bn_relu_bwd[_compact](...,
 * dOut = dout_cm, ..., accumulate = 2): 2 = "statistics are complete, skip that pass".
 * seg_off NULL: dense lists (y (b,c,p,s), e ignored); else compact lists (y (b,c,e) rows, s ignored). */
int sig3d_bn_relu_bwd_top_from_pm(int b, int c, int p, int s, long e, const float *dout_pm, const int *arg,
                                  const float *y, const float *scale, const float *shift, const float *mean,
                                  const float *invstd, const int *seg_off, float *dout_cm, double *s1, double *s2,
                                  int accumulate, void *stream);
int sig3d_bn_relu_bwd_compact(int b, int c, long e, int p, const float *dA, const float *dOut, const int *arg,
                              const float *y, const float *scale, const float *shift, const float *mean,
                              const float *invstd, double *s1, double *s2, float *dY, int accumulate,
                              const int *n_act, const float *mult, const int *centre_of, const int *seg_off,
                              void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SIG3D_HIP_H */

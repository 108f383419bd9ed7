// sqa_loss.hip -- the SQA3D training loss and the situational localisation target as ONE launch each.
//
// Replaces, on the GPU hot path, what the reference computes with ~40 small torch kernels per step:
//   lib/loss_helper.py:195-227  compute_aux_situation_loss (MSE / L1 on position [:, :3] and rotation [:, 3:])
//                               compute_answer_classification_loss (BCE-with-logits, reduction 'sum' / batch)
//   lib/loss_helper.py:286-300  loss = SITUATION_W * aux + QA_W * answer;  loss *= 10
//   situation3d/models/sqa_module.py:328-338  w = exp(-|p_xy - t_xy|^2 / (2 * 0.16^2)), normalised per scene
// The loss kernel also emits d loss / d answer_scores and d loss / d aux_scores, so its backward is a
// multiplication by the incoming gradient (one more launch).  Sizes are tiny (B x 706 logits, B x 7 pose
// values, B x 256 token positions): one workgroup does all of it, what is saved is launches (~1.3 us of
// kernel boundary each inside the captured step, MI355X_MICROARCH.md "boundary").
#include "sig3d_common.h"

namespace {

__device__ __forceinline__ float block_sum_256(float v, float *s_red) {
  v = wave_allreduce_sum_f32(v);
  __syncthreads();
  if (lane_id() == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

constexpr int LOSS_THREADS = 1024;

__device__ __forceinline__ float block_sum_1024(float v, float *s_red) {
  v = wave_allreduce_sum_f32(v);
  __syncthreads();
  if (lane_id() == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < LOSS_THREADS / 64; ++i) t += s_red[i];
  return t;
}

// losses[5] = {loss, answer_loss, pos_loss, rot_loss, aux_loss}
// One workgroup of 1024 threads, four elements per thread and trip with all eight loads requested before the first is
// used: the first version (256 threads, one element per trip) spent 22 dependent memory round trips = 17 us on 45 KB.
__global__ __launch_bounds__(LOSS_THREADS) void sqa_loss_kernel(int b, int num_answers, int aux_dim, int l1,
                                                                const float *__restrict__ answer_scores,
                                                                const float *__restrict__ answer_targets,
                                                                const float *__restrict__ aux_scores,
                                                                const float *__restrict__ aux_targets, float qa_w,
                                                                float situation_w, float pos_w, float rot_w, float amplify,
                                                                float *__restrict__ losses, float *__restrict__ d_answer,
                                                                float *__restrict__ d_aux) {
  __shared__ float s_red[LOSS_THREADS / 64];
  const int tid = threadIdx.x;
  // auxiliary operands first: their loads travel with the answer loop's
  const int n_aux = b * aux_dim;
  const float ax = tid < n_aux ? aux_scores[tid] : 0.f, at = tid < n_aux ? aux_targets[tid] : 0.f;
  // answer loss: binary_cross_entropy_with_logits(x, z, reduction='sum') / B  (loss_helper.py:222-225)
  //   l = max(x, 0) - x z + log(1 + exp(-|x|)),   dl/dx = sigmoid(x) - z
  const float ga = amplify * qa_w / (float)b;
  const int n = b * num_answers;
  float acc = 0.f;
  for (int i0 = tid; i0 < n; i0 += 4 * LOSS_THREADS) {
    float x[4], z[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + u * LOSS_THREADS, n - 1);
      x[u] = answer_scores[i];
      z[u] = answer_targets[i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * LOSS_THREADS;
      if (i < n) {
        acc += fmaxf(x[u], 0.f) - x[u] * z[u] + log1pf(__expf(-fabsf(x[u])));
        const float sg = 1.f / (1.f + __expf(-x[u]));
        d_answer[i] = ga * (sg - z[u]);
      }
    }
  }
  const float answer_loss = block_sum_1024(acc, s_red) / (float)b;
  // auxiliary situation loss: mean over B x 3 position values and B x (aux_dim - 3) rotation values
  const int nrot = aux_dim - 3;
  const float gp = amplify * situation_w * pos_w / (float)(b * 3);
  const float gr = amplify * situation_w * rot_w / (float)(b * nrot);
  float accp = 0.f, accr = 0.f;
  for (int i = tid; i < n_aux; i += LOSS_THREADS) {
    const int c = i % aux_dim;
    const float d = i == tid ? ax - at : aux_scores[i] - aux_targets[i];
    const bool is_pos = c < 3;
    if (l1) {
      (is_pos ? accp : accr) += fabsf(d);
      const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      d_aux[i] = (is_pos ? gp : gr) * sgn;
    } else {
      (is_pos ? accp : accr) += d * d;
      d_aux[i] = (is_pos ? gp : gr) * 2.f * d;
    }
  }
  const float pos_loss = block_sum_1024(accp, s_red) / (float)(b * 3);
  const float rot_loss = block_sum_1024(accr, s_red) / (float)(b * nrot);
  if (tid == 0) {
    const float aux = pos_w * pos_loss + rot_w * rot_loss;
    losses[0] = amplify * (situation_w * aux + qa_w * answer_loss);
    losses[1] = answer_loss;
    losses[2] = pos_loss;
    losses[3] = rot_loss;
    losses[4] = aux;
  }
}

// g_answer = d_answer * *upstream, g_aux = d_aux * *upstream
__global__ __launch_bounds__(256) void sqa_loss_scale_kernel(int n_answer, int n_aux, const float *__restrict__ upstream,
                                                             const float *__restrict__ d_answer,
                                                             const float *__restrict__ d_aux,
                                                             float *__restrict__ g_answer, float *__restrict__ g_aux) {
  const float u = *upstream;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_answer) g_answer[i] = d_answer[i] * u;
  else if (i - n_answer < n_aux) g_aux[i - n_answer] = d_aux[i - n_answer] * u;
}

// one workgroup per scene: distance in the (x, y) plane to the agent, Gaussian weight, normalised over the tokens
__global__ __launch_bounds__(256) void gaussian_target_kernel(int t, int pdim, float inv_two_sigma2,
                                                              const float *__restrict__ positions,
                                                              const float *__restrict__ pose, int pose_dim,
                                                              float *__restrict__ out) {
  __shared__ float s_red[4];
  const int bi = blockIdx.x;
  const float tx = pose[(size_t)bi * pose_dim + 0], ty = pose[(size_t)bi * pose_dim + 1];
  const float *p = positions + (size_t)bi * t * pdim;
  float acc = 0.f;
  for (int i = threadIdx.x; i < t; i += 256) {
    const float dx = p[(size_t)i * pdim + 0] - tx, dy = p[(size_t)i * pdim + 1] - ty;
    // torch.norm then **2 in the reference: sqrt and square again (sqa_module.py:333-335)
    const float dist = sqrtf(dx * dx + dy * dy);
    const float w = expf(-(dist * dist) * inv_two_sigma2);
    out[(size_t)bi * t + i] = w;
    acc += w;
  }
  const float total = block_sum_256(acc, s_red);
  for (int i = threadIdx.x; i < t; i += 256) out[(size_t)bi * t + i] /= total;
}

}  // namespace

extern "C" int sig3d_sqa_loss(int b, int num_answers, int aux_dim, int l1, const float *answer_scores,
                              const float *answer_targets, const float *aux_scores, const float *aux_targets,
                              float qa_w, float situation_w, float pos_w, float rot_w, float amplify, float *losses,
                              float *d_answer, float *d_aux, void *stream_) {
  SIG3D_REQUIRE(b >= 1 && num_answers >= 1 && aux_dim >= 4, "need b >= 1, answers >= 1, aux_dim >= 4");
  hipLaunchKernelGGL(sqa_loss_kernel, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream_, b, num_answers, aux_dim, l1,
                     answer_scores, answer_targets, aux_scores, aux_targets, qa_w, situation_w, pos_w, rot_w, amplify,
                     losses, d_answer, d_aux);
  SIG3D_LAUNCH_CHECK("sqa_loss_kernel");
  return 0;
}

extern "C" int sig3d_sqa_loss_scale(int n_answer, int n_aux, const float *upstream, const float *d_answer,
                                    const float *d_aux, float *g_answer, float *g_aux, void *stream_) {
  SIG3D_REQUIRE(n_answer >= 0 && n_aux >= 0, "negative size");
  if (n_answer + n_aux == 0) return 0;
  hipLaunchKernelGGL(sqa_loss_scale_kernel, dim3(sig3d_ceil_div(n_answer + n_aux, 256)), dim3(256), 0,
                     (hipStream_t)stream_, n_answer, n_aux, upstream, d_answer, d_aux, g_answer, g_aux);
  SIG3D_LAUNCH_CHECK("sqa_loss_scale_kernel");
  return 0;
}

extern "C" int sig3d_gaussian_target(int b, int t, int pdim, float sigma, const float *positions, const float *pose,
                                     int pose_dim, float *out, void *stream_) {
  SIG3D_REQUIRE(b >= 0 && t >= 1 && pdim >= 2 && pose_dim >= 2 && sigma > 0.f, "bad sizes");
  if (b == 0) return 0;
  hipLaunchKernelGGL(gaussian_target_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream_, t, pdim,
                     1.f / (2.f * sigma * sigma), positions, pose, pose_dim, out);
  SIG3D_LAUNCH_CHECK("gaussian_target_kernel");
  return 0;
}

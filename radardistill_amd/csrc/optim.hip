// Optimizer step of the reference training loop as two multi-tensor kernels (no per-tensor launches, no host sync):
//   1. global gradient 2-norm (clip_grad_norm_(10), tools/train_utils/train_utils.py:62)
//   2. decoupled weight decay p *= 1 - wd*lr, then Adam (tools/train_utils/optimization/fastai_optim.py:135-152 +
//      torch.optim.Adam, betas = (momentum of the one-cycle schedule, 0.99), eps 1e-8), with the clip coefficient
//      read from device memory.
// HBM-bound: reads p, g, m, v and writes p, m, v once (28 B / parameter).  Tensors are described by a device table so
// the ~500 parameter tensors of the student are one launch; each block owns one 4096-element chunk of one tensor.
#include "common.hpp"

using namespace rd;

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int OPT_CHUNK = 4096;

// chunk table entry: tensor id + element offset of the chunk
// Gradients live either in per-tensor buffers (t.grad) or, after the data-parallel all-reduce, in ONE flat buffer laid out like the
// moment buffers (element offset of tensor t = t.exp_avg - tensors[0].exp_avg), to be multiplied by grad_scale (1 / world size).
__device__ __forceinline__ const float *grad_ptr(const rd_opt_tensor *tensors, const rd_opt_tensor &t, const float *flat) {
    return flat ? flat + (t.exp_avg - tensors[0].exp_avg) : t.grad;
}

// Does tensor `tid` take part in this step?  Single process: it has a gradient pointer.  With a flat all-reduced buffer the answer
// must be the SAME on every rank (a branch unused by one rank's batch has p.grad None there only; deciding from the rank-local
// pointer would give the ranks different norms, clip coefficients and Adam step counts): `present` holds, per tensor, the MAX over
// ranks of "has a gradient" (rd_grad_presence + all-reduce), and the values come from the flat buffer, where absent ranks packed zeros.
__device__ __forceinline__ bool takes_part(const rd_opt_tensor &t, int tid, const float *flat, const float *present) {
    return (flat && present) ? present[tid] > 0.f : t.grad != nullptr;
}

__global__ void k_grad_presence(const rd_opt_tensor *__restrict__ tensors, int n, float *present) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) present[i] = tensors[i].grad ? 1.f : 0.f;
}

// inv_loss_scale (device scalar or NULL): mixed-precision loss scaling (torch.cuda.amp.GradScaler, train_utils.py:23,57-64): the
// gradients in memory are S times too large; the norm and the Adam update use g / S without a separate unscale pass.
__global__ __launch_bounds__(256) void k_gradnorm_partial(const rd_opt_tensor *__restrict__ tensors, const int2 *__restrict__ chunks, int n_chunks,
                                                          const float *__restrict__ flat, float grad_scale, const float *__restrict__ inv_loss_scale,
                                                          const float *__restrict__ present, float *partial) {
    __shared__ float red[4];
    if (inv_loss_scale) grad_scale *= inv_loss_scale[0];
    const int2 ch = chunks[blockIdx.x];
    const rd_opt_tensor t = tensors[ch.x];
    const float *g = grad_ptr(tensors, t, flat) + ch.y;
    const int64_t n = takes_part(t, ch.x, flat, present) ? min((int64_t)OPT_CHUNK, t.numel - ch.y) : 0;       // no gradient this step: torch's clip skips the tensor
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        float v = g[i] * grad_scale;
        s += v * v;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void k_gradnorm_final(const float *partial, int n, float max_norm, float *out /*[2]: total_norm, clip_coef*/, int32_t *overflow_count) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float total = (float)sqrt(red[0]);
        out[0] = total;
        float coef = max_norm > 0.f ? max_norm / (total + 1e-6f) : 1.f;      // max_norm <= 0: norm only (loss-scaling overflow check)
        out[1] = coef < 1.f ? coef : 1.f;       // torch.clamp(max=1.0); NaN total -> NaN coef -> NaN update, as in torch
        if (!(coef == coef)) out[1] = coef;
        if (overflow_count && !isfinite(total)) overflow_count[0] += 1;      // a step GradScaler skips does not count as an Adam step
    }
}

// A tensor whose gradient pointer is NULL is only decayed (OptimWrapper.step decays every trainable parameter, torch.optim.Adam
// skips parameters without a gradient: no moment update, no step count).  `skipped[t]` = how many optimizer steps tensor t sat out,
// so its own Adam step count is step - skipped[t]; the kernel itself counts a sat-out step (device-owned, so that the data-parallel
// case needs no host read of the group-wide presence mask); with skipped == NULL every tensor is at `step` and the bias corrections
// come precomputed (in double) from the host.
__global__ __launch_bounds__(256) void k_adam(const rd_opt_tensor *__restrict__ tensors, const int2 *__restrict__ chunks, double lr, float beta1,
                                              float beta2, float eps, float decay, float step_size0, float bc2_sqrt0, double beta1d, double beta2d,
                                              int step, int32_t *__restrict__ skipped, const float *__restrict__ clip,
                                              const float *__restrict__ flat, float grad_scale, const float *__restrict__ inv_loss_scale,
                                              int skip_nonfinite, const int32_t *__restrict__ overflow_count, const float *__restrict__ present) {
    __shared__ float sh[2];
    // GradScaler.step: an overflowed step (non-finite gradient norm) leaves parameters, moments and the decoupled decay untouched
    if (skip_nonfinite && clip && !isfinite(clip[0])) return;
    if (inv_loss_scale) grad_scale *= inv_loss_scale[0];
    const int2 ch = chunks[blockIdx.x];
    const rd_opt_tensor t = tensors[ch.x];
    const int64_t n = min((int64_t)OPT_CHUNK, t.numel - ch.y);
    float *p = t.param + ch.y;
    if (!takes_part(t, ch.x, flat, present)) {
        for (int64_t i = threadIdx.x; i < n; i += 256) p[i] *= decay;
        if (skipped && ch.y == 0 && threadIdx.x == 0) skipped[ch.x] += 1;       // nobody reads skipped[ch.x] in a step the tensor sits out
        return;
    }
    float step_size = step_size0, bc2_sqrt = bc2_sqrt0;
    if (skipped != nullptr || overflow_count != nullptr) {
        if (threadIdx.x == 0) {
            const int own = step - (skipped ? skipped[ch.x] : 0) - (overflow_count ? overflow_count[0] : 0);
            sh[0] = (float)(lr / (1.0 - pow(beta1d, (double)own)));
            sh[1] = (float)sqrt(1.0 - pow(beta2d, (double)own));
        }
        __syncthreads();
        step_size = sh[0];
        bc2_sqrt = sh[1];
    }
    const float *g = grad_ptr(tensors, t, flat) + ch.y;
    float *m = t.exp_avg + ch.y, *v = t.exp_avg_sq + ch.y;
    const float cc = (clip ? clip[1] : 1.f) * grad_scale;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float gi = g[i] * cc;
        float pi = p[i] * decay;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= step_size * (mi / denom);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

// pack: flat[offset(t) + i] = t.grad[i] -- ONE launch gathers the ~500 gradient tensors into the buffer that is all-reduced
__global__ __launch_bounds__(256) void k_pack_grads(const rd_opt_tensor *__restrict__ tensors, const int2 *__restrict__ chunks, float *flat) {
    const int2 ch = chunks[blockIdx.x];
    const rd_opt_tensor t = tensors[ch.x];
    const int64_t n = min((int64_t)OPT_CHUNK, t.numel - ch.y);
    float *d = flat + (t.exp_avg - tensors[0].exp_avg) + ch.y;
    if (t.grad == nullptr) {                    // no gradient on this rank: contributes zeros to the sum
        for (int64_t i = threadIdx.x; i < n; i += 256) d[i] = 0.f;
        return;
    }
    const float *g = t.grad + ch.y;
    for (int64_t i = threadIdx.x; i < n; i += 256) d[i] = g[i];
}

// The same copy for an explicit list of (source, destination, count) triples passed BY VALUE in the kernel arguments: one bucket of
// the overlapped data-parallel exchange is packed the moment its last gradient exists, without a device-resident descriptor table.
struct PackList {
    rd_pack_job jobs[RD_PACK_LIST_MAX];
    int first_block[RD_PACK_LIST_MAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void k_pack_list(const PackList t) {
    int j = 0;
    while (j + 1 < t.n && (int)blockIdx.x >= t.first_block[j + 1]) ++j;
    const rd_pack_job job = t.jobs[j];
    const int64_t base = (int64_t)(blockIdx.x - t.first_block[j]) * OPT_CHUNK;
    const int64_t n = min((int64_t)OPT_CHUNK, job.numel - base);
    for (int64_t i = threadIdx.x; i < n; i += 256) job.dst[base + i] = job.src ? job.src[base + i] : 0.f;
}

extern "C" int rd_pack_grads_list(const rd_pack_job *jobs_host, int n_jobs, void *stream) {
    RD_REQUIRE(n_jobs >= 0 && n_jobs <= RD_PACK_LIST_MAX, "rd_pack_grads_list: at most %d tensors per call", RD_PACK_LIST_MAX);
    if (n_jobs == 0) return RD_OK;
    PackList t;
    int blocks = 0;
    for (int j = 0; j < n_jobs; ++j) {
        RD_REQUIRE(jobs_host[j].dst && jobs_host[j].numel > 0, "rd_pack_grads_list: bad job %d", j);
        t.jobs[j] = jobs_host[j];
        t.first_block[j] = blocks;
        blocks += (int)cdiv(jobs_host[j].numel, OPT_CHUNK);
    }
    t.first_block[n_jobs] = blocks;
    t.n = n_jobs;
    k_pack_list<<<blocks, 256, 0, S(stream)>>>(t);
    return check_launch("rd_pack_grads_list");
}

extern "C" int rd_pack_grads(const rd_opt_tensor *tensors_dev, const int32_t *chunks_dev, int n_chunks, float *flat, void *stream) {
    RD_REQUIRE(flat != nullptr, "rd_pack_grads: flat buffer is NULL");
    if (n_chunks <= 0) return RD_OK;
    k_pack_grads<<<n_chunks, 256, 0, S(stream)>>>(tensors_dev, reinterpret_cast<const int2 *>(chunks_dev), flat);
    return check_launch("rd_pack_grads");
}

extern "C" int rd_grad_norm(const rd_opt_tensor *tensors_dev, const int32_t *chunks_dev, int n_chunks, float max_norm, float *out2,
                            float *ws, int64_t ws_bytes, const float *flat_grad, float grad_scale, const float *inv_loss_scale_dev,
                            int32_t *overflow_count_dev, const float *present_dev, void *stream) {
    RD_REQUIRE(n_chunks >= 0 && ws_bytes >= (int64_t)n_chunks * 4, "rd_grad_norm: workspace too small");
    hipStream_t st = S(stream);
    if (n_chunks > 0)
        k_gradnorm_partial<<<n_chunks, 256, 0, st>>>(tensors_dev, reinterpret_cast<const int2 *>(chunks_dev), n_chunks, flat_grad, grad_scale,
                                                     inv_loss_scale_dev, present_dev, ws);
    k_gradnorm_final<<<1, 256, 0, st>>>(ws, n_chunks, max_norm, out2, overflow_count_dev);
    return check_launch("rd_grad_norm");
}

extern "C" int rd_adam_step(const rd_opt_tensor *tensors_dev, const int32_t *chunks_dev, int n_chunks, double lr, double beta1, double beta2,
                            double eps, double weight_decay, int step, int32_t *skipped_dev, const float *clip_dev,
                            const float *flat_grad, float grad_scale, const float *inv_loss_scale_dev, int skip_nonfinite,
                            const int32_t *overflow_count_dev, const float *present_dev, void *stream) {
    RD_REQUIRE(!skip_nonfinite || clip_dev, "rd_adam_step: skip_nonfinite needs the norm of rd_grad_norm (clip_dev)");
    RD_REQUIRE(step >= 1, "rd_adam_step: step must be >= 1");
    if (n_chunks <= 0) return RD_OK;
    // scalars are formed in double like the reference's Python floats and rounded once (torch applies them to fp32 tensors)
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    k_adam<<<n_chunks, 256, 0, S(stream)>>>(tensors_dev, reinterpret_cast<const int2 *>(chunks_dev), lr, (float)beta1, (float)beta2, (float)eps,
                                            (float)(1.0 - weight_decay * lr), (float)(lr / bc1), (float)sqrt(bc2), beta1, beta2, step, skipped_dev,
                                            clip_dev, flat_grad, grad_scale, inv_loss_scale_dev, skip_nonfinite, overflow_count_dev, present_dev);
    return check_launch("rd_adam_step");
}

extern "C" int rd_grad_presence(const rd_opt_tensor *tensors_dev, int n_tensors, float *present_dev, void *stream) {
    RD_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || (tensors_dev && present_dev)), "rd_grad_presence: NULL table or output");
    if (n_tensors == 0) return RD_OK;
    k_grad_presence<<<cdiv(n_tensors, 256), 256, 0, S(stream)>>>(tensors_dev, n_tensors, present_dev);
    return check_launch("rd_grad_presence");
}

extern "C" int rd_opt_chunk_elems(void) { return OPT_CHUNK; }

// Adam step of ALL parameter tensors of the models in one launch.
//
// The reference optimises with torch.optim.Adam built by name (train.py:189, trainer/trainer.py:216-238; step at
// trainer/trainer.py:120).  torch's own fused kernel (`torch._fused_adam_`, multi_tensor_apply) walks 64 K-element chunks
// with one workgroup each: the 3 M parameters of PNA(depth 4) + Net3D are ~50 workgroups per launch on a 256-CU chip -
// three launches of ~38 us per step, 3 % of the step's kernel time, for 84 MB of traffic that takes ~20 us at speed.
// This kernel applies the SAME update (expression for expression, including which products are formed in double - see
// adam_update) over 4 K-element chunks of a device-resident tensor table built once per optimizer, so that one launch
// fills the chip; it also advances the per-parameter step counters.
#include "common.h"

namespace i3d {

struct AdamChunk {      // device table entry: one <= 4096-element piece of a parameter tensor
    float* p;
    const float* g;
    float* m;
    float* v;
    int n;
};

constexpr int ADAM_CHUNK = 4096;

// torch/aten/src/ATen/native/cuda/fused_adam_utils.cuh: adam_math with opmath_t = float; lr, beta1, beta2, weight_decay, eps
// are doubles there, so these products are formed in double and rounded once on assignment
__device__ __forceinline__ void adam_update(float& param, float grad, float& exp_avg, float& exp_avg_sq, double lr, double beta1,
                                            double beta2, double weight_decay, double eps, float bias_correction1,
                                            float bias_correction2_sqrt) {
    if (weight_decay != 0) grad = (float)((double)grad + (double)param * weight_decay);
    exp_avg = (float)(beta1 * (double)exp_avg + (1 - beta1) * (double)grad);
    exp_avg_sq = (float)(beta2 * (double)exp_avg_sq + (1 - beta2) * (double)grad * (double)grad);
    const float step_size = (float)(lr / (double)bias_correction1);
    const float denom = (float)((double)(sqrtf(exp_avg_sq) / bias_correction2_sqrt) + eps);
    param -= step_size * exp_avg / denom;
}

__global__ void __launch_bounds__(256)
adam_kernel(const AdamChunk* __restrict__ table, int n_chunks, float* steps, int n_steps, double lr, double beta1, double beta2,
            double weight_decay, double eps, float bc1, float bc2_sqrt) {
    I3D_CHAIN_PRIO();
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < n_steps; i += 256) steps[i] += 1.f;
    const AdamChunk c = table[blockIdx.x];
    const bool vec = ((((uintptr_t)c.p | (uintptr_t)c.g | (uintptr_t)c.m | (uintptr_t)c.v) & 15) == 0);
    if (vec) {
        const int n4 = c.n / 4;
        for (int i = threadIdx.x; i < n4; i += 256) {
            float4 p = reinterpret_cast<float4*>(c.p)[i];
            const float4 g = reinterpret_cast<const float4*>(c.g)[i];
            float4 m = reinterpret_cast<float4*>(c.m)[i];
            float4 v = reinterpret_cast<float4*>(c.v)[i];
            adam_update(p.x, g.x, m.x, v.x, lr, beta1, beta2, weight_decay, eps, bc1, bc2_sqrt);
            adam_update(p.y, g.y, m.y, v.y, lr, beta1, beta2, weight_decay, eps, bc1, bc2_sqrt);
            adam_update(p.z, g.z, m.z, v.z, lr, beta1, beta2, weight_decay, eps, bc1, bc2_sqrt);
            adam_update(p.w, g.w, m.w, v.w, lr, beta1, beta2, weight_decay, eps, bc1, bc2_sqrt);
            reinterpret_cast<float4*>(c.p)[i] = p;
            reinterpret_cast<float4*>(c.m)[i] = m;
            reinterpret_cast<float4*>(c.v)[i] = v;
        }
        for (int i = n4 * 4 + threadIdx.x; i < c.n; i += 256) {
            float p = c.p[i], m = c.m[i], v = c.v[i];
            adam_update(p, c.g[i], m, v, lr, beta1, beta2, weight_decay, eps, bc1, bc2_sqrt);
            c.p[i] = p; c.m[i] = m; c.v[i] = v;
        }
    } else {
        for (int i = threadIdx.x; i < c.n; i += 256) {
            float p = c.p[i], m = c.m[i], v = c.v[i];
            adam_update(p, c.g[i], m, v, lr, beta1, beta2, weight_decay, eps, bc1, bc2_sqrt);
            c.p[i] = p; c.m[i] = m; c.v[i] = v;
        }
    }
}

}  // namespace i3d

using namespace i3d;

extern "C" int i3d_adam_chunk_elems(void) { return ADAM_CHUNK; }
extern "C" int i3d_adam_chunk_bytes(void) { return (int)sizeof(AdamChunk); }

extern "C" int i3d_adam_step(const void* chunk_table, int n_chunks, float* steps, int n_steps, double lr, double beta1,
                             double beta2, double weight_decay, double eps, double bias_correction1,
                             double bias_correction2_sqrt, void* stream) {
    I3D_CHECK_ARG(chunk_table != nullptr && n_chunks > 0 && n_steps >= 0, "bad arguments");
    hipLaunchKernelGGL(adam_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const AdamChunk*)chunk_table, n_chunks,
                       steps, n_steps, lr, beta1, beta2, weight_decay, eps, (float)bias_correction1, (float)bias_correction2_sqrt);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

// Shared helpers for the gfx950 kernels of lib3dinfomax_hip.so (CDNA4 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/infomax3d_hip.h"

namespace i3d {

void set_error(const char* fmt, ...);

// the process-wide collective table of synchronised BatchNorm (comm.hip), null when none is set
const I3dCollectives* collectives();

// Every C-ABI entry point returns 0 on success.  Launch errors are reported through
// hipGetLastError() right after the launch (no device synchronisation: the caller's stream
// stays asynchronous) and turned into a message retrievable with i3d_last_error().
#define I3D_CHECK_ARG(cond, msg)                                  \
    do {                                                          \
        if (!(cond)) {                                            \
            i3d::set_error("%s: invalid argument: %s", __func__, msg); \
            return I3D_ERR_INVALID;                               \
        }                                                         \
    } while (0)

#define I3D_CHECK_LAUNCH()                                                       \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            i3d::set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return I3D_ERR_LAUNCH;                                               \
        }                                                                        \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

constexpr int WAVE = 64;

// First statement of every kernel of the step's dependent chain (everything but the weight-gradient family of wgrad.hip and
// the 3D network's edge kernels, which run beside it on streams of their own): wave priority 3.  The weight-gradient panels
// are a hundred microseconds of back-to-back MFMA on every CU; at equal priority the SIMD's arbiter serves those (older)
// waves first and the small latency-bound kernels of the chain take 2-5 x as long next to them as alone (the BatchNorm
// backward reduction: 13 us alone, 74 us beside a panel kernel, 25-30 us with this; step 2.246 -> 2.176 ms).
#ifndef I3D_CHAIN_PRIO_LEVEL
#define I3D_CHAIN_PRIO_LEVEL 3
#endif
#define I3D_CHAIN_PRIO() __builtin_amdgcn_s_setprio(I3D_CHAIN_PRIO_LEVEL)

// apply_act / act_grad: the activations the FUSED kernels take (GEMM epilogues, statistics / BatchNorm passes, the edge stage):
// the reference's configurations use ReLU, SiLU (3D network), Sigmoid (gate) and LeakyReLU (tower mixing) only.  The switch is
// inlined into every one of those kernels - the transcendental cases of the elementwise set below cost the training step 5 %
// when they sat in it (round 4: 2.160 -> 2.277 ms, same box) - so the remaining activations exist in the elementwise kernels only
// (apply_act_any / act_grad_any: i3d_act_fwd / i3d_act_bwd) and the host side runs them as a pass of their own.
__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case I3D_ACT_RELU: return x > 0.f ? x : 0.f;
        case I3D_ACT_SILU: return x / (1.f + __expf(-x));
        case I3D_ACT_SIGMOID: return 1.f / (1.f + __expf(-x));
        case I3D_ACT_LEAKY_RELU: return x > 0.f ? x : 0.01f * x;
        default: return x;
    }
}

// derivative of act at pre-activation x
__device__ __forceinline__ float act_grad(float x, int act) {
    switch (act) {
        case I3D_ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case I3D_ACT_SILU: {
            float s = 1.f / (1.f + __expf(-x));
            return s * (1.f + x * (1.f - s));
        }
        case I3D_ACT_SIGMOID: {
            float s = 1.f / (1.f + __expf(-x));
            return s * (1.f - s);
        }
        case I3D_ACT_LEAKY_RELU: return x > 0.f ? 1.f : 0.01f;
        default: return 1.f;
    }
}

// The same with the activation class known at compile time.  GA = false: none / ReLU / LeakyReLU only - what every block of the 2D
// network's yml configurations uses; the kernels that take the class as a template parameter (statistics / BatchNorm passes,
// the fused GEMM epilogue, the edge gather-combine) are launched in that form whenever their activation codes allow: without
// the exp-based cases in the inlined switch the training step is 1.4 % shorter (round 4, same box: 2.141 -> 2.111 ms with the
// cases compiled out everywhere).  Same expressions: same bits.
inline bool relu_class(int act) { return act == I3D_ACT_NONE || act == I3D_ACT_RELU || act == I3D_ACT_LEAKY_RELU; }
template <bool GA>
__device__ __forceinline__ float apply_act_c(float x, int act) {
    if (GA) return apply_act(x, act);
    return act == I3D_ACT_RELU ? (x > 0.f ? x : 0.f) : (act == I3D_ACT_LEAKY_RELU ? (x > 0.f ? x : 0.01f * x) : x);
}
template <bool GA>
__device__ __forceinline__ float act_grad_c(float x, int act) {
    if (GA) return act_grad(x, act);
    return act == I3D_ACT_RELU ? (x > 0.f ? 1.f : 0.f) : (act == I3D_ACT_LEAKY_RELU ? (x > 0.f ? 1.f : 0.01f) : 1.f);
}

// every elementwise entry of the reference's SUPPORTED_ACTIVATION_MAP (models/base_layers.py:5), torch's default parameters
__device__ __forceinline__ float apply_act_any(float x, int act) {
    switch (act) {
        case I3D_ACT_TANH: return tanhf(x);
        case I3D_ACT_ELU: return x > 0.f ? x : expm1f(x);
        case I3D_ACT_SELU: return 1.0507009873554804934193349852946f * (x > 0.f ? x : 1.6732632423543772848170429916717f * expm1f(x));
        case I3D_ACT_SOFTPLUS: return x > 20.f ? x : log1pf(expf(x));
        default: return apply_act(x, act);
    }
}

__device__ __forceinline__ float act_grad_any(float x, int act) {
    switch (act) {
        case I3D_ACT_TANH: {
            const float t = tanhf(x);
            return 1.f - t * t;
        }
        case I3D_ACT_ELU: return x > 0.f ? 1.f : expf(x);
        case I3D_ACT_SELU: return 1.0507009873554804934193349852946f * (x > 0.f ? 1.f : 1.6732632423543772848170429916717f * expf(x));
        case I3D_ACT_SOFTPLUS: return x > 20.f ? 1.f : 1.f / (1.f + expf(-x));
        default: return act_grad(x, act);
    }
}

// the fused kernels' activation codes (see above)
inline bool fused_act(int act) { return act >= I3D_ACT_NONE && act <= I3D_ACT_LEAKY_RELU; }

// aggregate.hip: the next K4 forward launch of this thread carries these timing events (bench.py's in-step roofline figure)
void k4_time_next_launch(void* start, void* stop);

}  // namespace i3d

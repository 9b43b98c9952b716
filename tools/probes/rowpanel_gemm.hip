// Probe: "row panel" forward GEMM  C[M,N] = A[M,K] W[N,K]^T  for the step's N = 200, K = 200 / 800 shapes.
//   * a workgroup owns BM rows and ALL N columns: A is read once, every wave owns a 16-row strip and 13 accumulator
//     tiles (16x16x4 fp32 MFMA, 13 x 16 = 208 columns) - one A fragment feeds 13 independent MFMA chains;
//   * K in chunks of KC = 40, double-buffered in LDS ([idx][k] images, pitch 42, 8-byte fragment reads feeding two MFMA
//     steps through a k permutation), the chunk after the next one in flight in registers.
// Question: does this beat the 64x64 / 64x32 tiled kernel of gemm.hip (24-26 us at M = 16.6 k, K = 200) ?
// Answer (MI355X, round 2): no.  M = 8.4 k: 19.6 us (gemm.hip 15-17.5), M = 16.6 k: 29.2 us (24-26), M = 133 k: 148 us = 72 TF
// (160-170 us = 63-67 TF: the only win); rocprofv3 MfmaUtil 20-48 %: with 110 KB of LDS a CU holds ONE workgroup, so the
// prologue (two chunk loads), the per-chunk barrier and the epilogue are never covered by another workgroup's MFMAs - at
// batch 512 there is less than one row panel per CU, so a persistent loop has nothing to pipeline across either.
//     hipcc --offload-arch=gfx950 -O3 tools/probes/rowpanel_gemm.hip -o tools/probes/rowpanel_gemm && tools/probes/rowpanel_gemm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int KC = 40, PITCH = KC + 2, NT = 13, NPAD = NT * 16;      // v2: double-buffered LDS chunks of 40 k

// k permutation inside a block of 8: lane (li, lk) supplies k = 8 t + 2 lk + u in MFMA step 2 t + u (u = 0, 1), for A and B
// alike (the sum over k does not care): one 8-byte LDS read feeds two steps
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) rowpanel_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                              float* __restrict__ C, int M, int N, int K) {
    constexpr int BM = WAVES * 16, THREADS = WAVES * 64;
    constexpr int A_F4 = BM * (KC / 4), B_F4 = NPAD * (KC / 4);
    constexpr int A_PER = (A_F4 + THREADS - 1) / THREADS, B_PER = (B_F4 + THREADS - 1) / THREADS;
    constexpr int BUF = (BM + NPAD) * PITCH;
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM;
    float4 ra[A_PER], rb[B_PER];
    auto load_chunk = [&](int kc) {
#pragma unroll
        for (int it = 0; it < A_PER; ++it) {
            const int s = tid + it * THREADS;
            const int r = s / (KC / 4), q = s % (KC / 4);
            const int m = m0 + r;
            ra[it] = (s < A_F4 && m < M && kc + 4 * q < K) ? *reinterpret_cast<const float4*>(A + (long)m * K + kc + 4 * q)
                                                              : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < B_PER; ++it) {
            const int s = tid + it * THREADS;
            const int n = s / (KC / 4), q = s % (KC / 4);
            rb[it] = (s < B_F4 && n < N && kc + 4 * q < K) ? *reinterpret_cast<const float4*>(W + (long)n * K + kc + 4 * q)
                                                             : make_float4(0, 0, 0, 0);
        }
    };
    auto store_chunk = [&](float* buf) {
        float* As = buf;
        float* Bs = buf + BM * PITCH;
#pragma unroll
        for (int it = 0; it < A_PER; ++it) {
            const int s = tid + it * THREADS;
            if (s < A_F4) {
                float* p = As + (s / (KC / 4)) * PITCH + 4 * (s % (KC / 4));
                *reinterpret_cast<float2*>(p) = make_float2(ra[it].x, ra[it].y);
                *reinterpret_cast<float2*>(p + 2) = make_float2(ra[it].z, ra[it].w);
            }
        }
#pragma unroll
        for (int it = 0; it < B_PER; ++it) {
            const int s = tid + it * THREADS;
            if (s < B_F4) {
                float* p = Bs + (s / (KC / 4)) * PITCH + 4 * (s % (KC / 4));
                *reinterpret_cast<float2*>(p) = make_float2(rb[it].x, rb[it].y);
                *reinterpret_cast<float2*>(p + 2) = make_float2(rb[it].z, rb[it].w);
            }
        }
    };
    floatx4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, lk = lane >> 4;
    const int a_off = (wave * 16 + li) * PITCH + 2 * lk;
    const int b_off = BM * PITCH + li * PITCH + 2 * lk;
    const int nchunks = (K + KC - 1) / KC;
    load_chunk(0);
    store_chunk(smem);
    if (nchunks > 1) load_chunk(KC);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const float* buf = smem + (c & 1) * BUF;
        if (c + 1 < nchunks) store_chunk(smem + ((c + 1) & 1) * BUF);        // regs hold chunk c + 1
        if (c + 2 < nchunks) load_chunk((c + 2) * KC);
#pragma unroll
        for (int t = 0; t < KC / 8; ++t) {
            const float2 av = *reinterpret_cast<const float2*>(buf + a_off + 8 * t);
            float2 bv[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = *reinterpret_cast<const float2*>(buf + b_off + j * 16 * PITCH + 8 * t);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[j].x, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[j].y, acc[j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = j * 16 + li;
        if (n >= N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wave * 16 + 4 * lk + r;
            if (m < M) C[(long)m * N + n] = acc[j][r];
        }
    }
}

template <int WAVES>
float run(const float* A, const float* W, float* C, int M, int N, int K, int reps) {
    const int BM = WAVES * 16;
    const size_t lds = (size_t)2 * (BM + NPAD) * PITCH * sizeof(float);
    hipFuncSetAttribute((const void*)rowpanel_kernel<WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((M + BM - 1) / BM), block(WAVES * 64);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(rowpanel_kernel<WAVES>, grid, block, lds, 0, A, W, C, M, N, K);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(rowpanel_kernel<WAVES>, grid, block, lds, 0, A, W, C, M, N, K);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("launch error: %s\n", hipGetErrorString(err));
    return ms * 1e3f / reps;
}

int main() {
    const int N = 200;
    const int shapes[][2] = {{8409, 200}, {16638, 200}, {8409, 800}, {49000, 200}, {133440, 200}, {67434, 800}};
    for (auto& sh : shapes) {
        const int M = sh[0], K = sh[1];
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
        srand(1);
        for (auto& v : hA) v = (float)rand() / RAND_MAX - 0.5f;
        for (auto& v : hW) v = (float)rand() / RAND_MAX - 0.5f;
        float *A, *W, *C;
        hipMalloc(&A, hA.size() * 4); hipMalloc(&W, hW.size() * 4); hipMalloc(&C, (size_t)M * N * 4);
        hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
        const float us4 = run<4>(A, W, C, M, N, K, 20);
        const float us5 = run<5>(A, W, C, M, N, K, 20);
        const float us8 = run<8>(A, W, C, M, N, K, 20);
        std::vector<float> hC((size_t)M * N);
        hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
        double maxerr = 0.0;
        for (int t = 0; t < 200; ++t) {
            const int m = (int)((long)rand() * 7919 % M), n = rand() % N;
            double ref = 0.0;
            for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * hW[(size_t)n * K + k];
            maxerr = fmax(maxerr, fabs(ref - hC[(size_t)m * N + n]));
        }
        const double gf = 2.0 * M * N * K * 1e-9;
        printf("M=%6d K=%4d: BM=64 %7.1f us %6.1f TF | BM=80 %7.1f us %6.1f TF | BM=128 %7.1f us %6.1f TF | max err %.2e\n", M, K, us4,
               gf / us4 * 1e3, us5, gf / us5 * 1e3, us8, gf / us8 * 1e3, maxerr);
        hipFree(A); hipFree(W); hipFree(C);
    }
    return 0;
}

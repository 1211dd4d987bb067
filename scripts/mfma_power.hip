// Energy per FLOP of the two bf16 MFMA shapes, registers only (no memory traffic): every wave keeps
// 128 accumulator registers and issues independent MFMAs back to back for ~1.5 s per shape while the
// host samples rocm-smi.  Build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_power.hip -o scripts/bin/mfma_power
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <atomic>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void mfma_loop(float* out, int iters) {
    bf16x8_t a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x % 7 + e)); }
    float sink = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16_t acc[8];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) sink += acc[i][0];
    } else {
        f32x4_t acc[32];
        for (int i = 0; i < 32; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 32; ++i) sink += acc[i][0];
    }
    if (sink == 123.456f) out[0] = sink;
}

static std::string smi() {
    FILE* f = popen("rocm-smi -P -c --csv 2>/dev/null | tail -1", "r");
    char buf[512] = {0};
    if (f) { if (!fgets(buf, sizeof buf, f)) buf[0] = 0; pclose(f); }
    return buf;
}

template <int SHAPE>
void run(const char* name) {
    float* out; hipMalloc(&out, 4);
    const int iters = 4000;                       // per launch: 32 MFMA x iters per wave (both shapes: same flops)
    const double flops_per_launch = 256.0 * 8 * 2 /*waves per CU: 2 WG x 8*/ * iters * 32.0 * 32768.0 / (SHAPE == 32 ? 1 : 1);
    std::atomic<bool> stop{false};
    std::string mid;
    std::thread poll([&] { std::this_thread::sleep_for(std::chrono::milliseconds(900)); mid = smi(); });
    auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    double secs = 0;
    while (secs < 1.8) {
        for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(512), dim3(512), 0, 0, out, iters);
        hipDeviceSynchronize();
        n += 10;
        secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    poll.join();
    // flops: 32x32x16 = 32768 per MFMA x 32 per iter; 16x16x32 = 16384 x 64 per iter: identical per iteration
    printf("%s: %.0f TF/s sustained over %.1f s; rocm-smi mid-run: %s", name, flops_per_launch * n / secs / 1e12, secs, mid.c_str());
    hipFree(out);
}

int main() {
    run<32>("v_mfma_f32_32x32x16_bf16");
    run<16>("v_mfma_f32_16x16x32_bf16");
    run<32>("v_mfma_f32_32x32x16_bf16");
    run<16>("v_mfma_f32_16x16x32_bf16");
    return 0;
}

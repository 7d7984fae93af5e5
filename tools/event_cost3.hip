// Cross-stream hand-over cost, emulating the sweep schedule: main runs a 50 us kernel per iteration and needs the result
// of the side stream's 30 us kernel of the iteration BEFORE THE PREVIOUS one (two entities alternate); the side kernel needs main's kernel of this iteration.
// With free hand-overs an iteration takes 50 us.  Variants: hipEventRecord after the kernel; the event attached to the
// kernel's own dispatch packet (hipExtLaunchKernelGGL stopEvent); a device flag written by the kernel + a gate kernel.
//   hipcc --offload-arch=gfx950 -O2 tools/event_cost3.hip -o /tmp/event_cost3 && /tmp/event_cost3
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void busy(long ticks, unsigned* flag, unsigned value) {
    long t0m = wall_clock64();
    while ((long)wall_clock64() - t0m < ticks) __builtin_amdgcn_s_sleep(4);
    if (flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void busy_poll(long ticks, unsigned* flag, unsigned value, unsigned* wait_flag, unsigned wait_value, long max_ticks, unsigned* err) {
    long t0m = wall_clock64();
    while ((long)wall_clock64() - t0m < ticks) __builtin_amdgcn_s_sleep(4);
    // the "prior load" point: every wave needs the side stream's result from here on
    if (threadIdx.x % 64 == 0)
        while (__hip_atomic_load(wait_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < wait_value) {
            __builtin_amdgcn_s_sleep(8);
            if ((long)wall_clock64() - t0m > max_ticks) { *err = 2; break; }
        }
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void gate(unsigned* flag, unsigned value, long max_ticks, unsigned* err) {
    long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < value) {
        __builtin_amdgcn_s_sleep(8);
        if ((long)wall_clock64() - t0 > max_ticks) { *err = 1; return; }
    }
}

int main(int argc, char** argv) {
    const unsigned evflags = hipEventDisableTiming | (argc > 1 ? hipEventReleaseToDevice : 0u);
    printf("event flags 0x%x\n", evflags);
    hipStream_t m, s;
    CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const double tpu = 100.0;      // wall_clock64(): 100 MHz
    long t50 = (long)(50 * tpu), t30 = (long)(30 * tpu);
    unsigned *flags, *err;
    CK(hipMalloc(&flags, 256)); CK(hipMemset(flags, 0, 256));
    CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    const int NB = 1024, NT = 256;      // fills the chip like K1 does (spinning waves)
    const int N = 300;
    std::vector<hipEvent_t> evM(N + 1), evS(N + 1);
    for (int i = 0; i <= N; i++) { CK(hipEventCreateWithFlags(&evM[i], evflags)); CK(hipEventCreateWithFlags(&evS[i], evflags)); }
    unsigned base = 0;
    for (int mode = 0; mode < 8; mode++) {
        const char* names[] = {"main only", "main + independent side", "hipEventRecord both ways", "stopEvent on the kernels (hipExtLaunchKernelGGL)",
                               "flag + gate kernel to side, event back", "stopEvent to side, event back", "flag + gate kernel both ways (no events)",
                               "flag + gate to side, flag back polled by the main kernel itself"};
        for (int rep = 0; rep < 2; rep++) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 1; i <= N; i++) {
                unsigned v = base + i;
                switch (mode) {
                case 0: busy<<<NB, NT, 0, m>>>(t50, nullptr, 0); break;
                case 1: busy<<<NB, NT, 0, m>>>(t50, nullptr, 0); busy<<<48, NT, 0, s>>>(t30, nullptr, 0); break;
                case 2:
                    if (i > 2) CK(hipStreamWaitEvent(m, evS[i - 2], 0));
                    busy<<<NB, NT, 0, m>>>(t50, nullptr, 0);
                    CK(hipEventRecord(evM[i], m));
                    CK(hipStreamWaitEvent(s, evM[i], 0));
                    busy<<<48, NT, 0, s>>>(t30, nullptr, 0);
                    CK(hipEventRecord(evS[i], s));
                    break;
                case 3:
                    if (i > 2) CK(hipStreamWaitEvent(m, evS[i - 2], 0));
                    hipExtLaunchKernelGGL(busy, dim3(NB), dim3(NT), 0, m, nullptr, evM[i], 0, t50, (unsigned*)nullptr, 0u);
                    CK(hipStreamWaitEvent(s, evM[i], 0));
                    hipExtLaunchKernelGGL(busy, dim3(48), dim3(NT), 0, s, nullptr, evS[i], 0, t30, (unsigned*)nullptr, 0u);
                    break;
                case 4:
                    if (i > 2) CK(hipStreamWaitEvent(m, evS[i - 2], 0));
                    busy<<<NB, NT, 0, m>>>(t50, flags, v);
                    gate<<<1, 64, 0, s>>>(flags, v, (long)(200000 * tpu), err);
                    busy<<<48, NT, 0, s>>>(t30, nullptr, 0);
                    CK(hipEventRecord(evS[i], s));
                    break;
                case 6:
                    if (i > 2) gate<<<1, 64, 0, m>>>(flags + 16, v - 2, (long)(200000 * tpu), err);
                    busy<<<NB, NT, 0, m>>>(t50, flags, v);
                    gate<<<1, 64, 0, s>>>(flags, v, (long)(200000 * tpu), err);
                    busy<<<48, NT, 0, s>>>(t30, flags + 16, v);
                    break;
                case 7:
                    busy_poll<<<NB, NT, 0, m>>>(t50, flags, v, flags + 16, i > 2 ? v - 2 : 0, (long)(200000 * tpu), err);
                    gate<<<1, 64, 0, s>>>(flags, v, (long)(200000 * tpu), err);
                    busy<<<48, NT, 0, s>>>(t30, flags + 16, v);
                    break;
                case 5:
                    if (i > 2) CK(hipStreamWaitEvent(m, evS[i - 2], 0));
                    hipExtLaunchKernelGGL(busy, dim3(NB), dim3(NT), 0, m, nullptr, evM[i], 0, t50, (unsigned*)nullptr, 0u);
                    CK(hipStreamWaitEvent(s, evM[i], 0));
                    busy<<<48, NT, 0, s>>>(t30, nullptr, 0);
                    CK(hipEventRecord(evS[i], s));
                    break;
                }
            }
            CK(hipDeviceSynchronize());
            double us = 1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / N;
            if (rep == 1) { printf("%-52s %7.2f us per iteration\n", names[mode], us); fflush(stdout); }
            base += N;
        }
    }
    unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("gate timeouts: %u\n", herr);
    return 0;
}

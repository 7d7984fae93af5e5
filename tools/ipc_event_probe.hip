// ipc_event_probe.hip -- do interprocess HIP events order work ACROSS two processes on the device, without host synchronisation?
// Two processes (fork before any HIP call) on one GPU.  A launches a kernel that spins `spin_ms` and then fills its buffer with a
// pattern, records an interprocess event behind it and tells B over a pipe -- without synchronising.  B then makes its stream wait
// for the opened event, copies A's buffer through its IPC mapping, and checks the pattern: stale data means the wait did not order
// the copy behind A's kernel.  Printed: how long B's hipStreamWaitEvent call took on the host (a call that blocks until the event
// completes is no use for overlap) and when B's copy finished relative to A's kernel.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[%d] %s failed: %s (line %d)\n", getpid(), #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
__global__ void k_spin_fill(unsigned *buf, size_t n, unsigned pattern, long long ticks)
{
    const long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    __syncthreads();
    // (only block 0 spins; the others would finish early: make them wait on block 0 through a flag in buf[n])
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(buf + n, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(buf + n, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(8);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = pattern + (unsigned)i;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Msg { hipIpcMemHandle_t mem; hipIpcEventHandle_t ev; };
int main(int argc, char **argv)
{
    const double spin_ms = argc > 1 ? atof(argv[1]) : 5.0;
    const size_t n = (size_t)(argc > 2 ? atol(argv[2]) : (16 << 20)) / 4;     // words
    int ab[2], ba[2];
    if (pipe(ab) || pipe(ba)) return 1;
    const pid_t pid = fork();
    const bool isA = pid != 0;
    const int rd = isA ? ba[0] : ab[0], wr = isA ? ab[1] : ba[1];
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned *buf; CK(hipMalloc((void **)&buf, (n + 64) * 4)); CK(hipMemset(buf, 0, (n + 64) * 4)); CK(hipDeviceSynchronize());
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventInterprocess));
    Msg mine, theirs;
    CK(hipIpcGetMemHandle(&mine.mem, buf)); CK(hipIpcGetEventHandle(&mine.ev, ev));
    if (write(wr, &mine, sizeof(mine)) != sizeof(mine) || read(rd, &theirs, sizeof(theirs)) != sizeof(theirs)) return 3;
    void *peer; CK(hipIpcOpenMemHandle(&peer, theirs.mem, hipIpcMemLazyEnablePeerAccess));
    hipEvent_t pev; CK(hipIpcOpenEventHandle(&pev, theirs.ev));
    char c = 'x';
    for (int round = 0; round < 3; round++) {
        const unsigned pattern = 0x1000000u * (unsigned)(round + 1);
        if (isA) {
            CK(hipMemsetAsync(buf + n, 0, 4, st));
            const double t0 = now_ms();
            hipLaunchKernelGGL(k_spin_fill, dim3(256), dim3(256), 0, st, buf, n, pattern, (long long)(spin_ms * 1e5));     // 100 MHz clock
            CK(hipEventRecord(ev, st));
            const double t1 = now_ms();
            if (write(wr, &c, 1) != 1) return 4;                // "recorded" -- no synchronisation before it
            CK(hipStreamSynchronize(st));
            const double t2 = now_ms();
            printf("[A] round %d: enqueue + record %.3f ms, kernel done after %.3f ms (absolute %.3f)\n", round, t1 - t0, t2 - t0, t2);
            if (read(rd, &c, 1) != 1) return 5;                 // B has verified: next round may overwrite
        } else {
            unsigned *local; CK(hipMalloc((void **)&local, n * 4)); CK(hipMemsetAsync(local, 0, n * 4, st)); CK(hipStreamSynchronize(st));
            if (read(rd, &c, 1) != 1) return 4;
            const double t0 = now_ms();
            CK(hipStreamWaitEvent(st, pev, 0));
            const double t1 = now_ms();
            CK(hipMemcpyAsync(local, peer, n * 4, hipMemcpyDeviceToDevice, st));
            const double t2 = now_ms();
            CK(hipStreamSynchronize(st));
            const double t3 = now_ms();
            unsigned *h = (unsigned *)malloc(n * 4); CK(hipMemcpy(h, local, n * 4, hipMemcpyDeviceToHost));
            size_t bad = 0; for (size_t i = 0; i < n; i++) bad += h[i] != pattern + (unsigned)i;
            printf("[B] round %d: hipStreamWaitEvent call %.3f ms on the host, copy enqueued after %.3f ms, copy complete after %.3f ms (absolute %.3f); %zu of %zu words stale -> %s\n",
                   round, t1 - t0, t2 - t0, t3 - t0, t3, bad, n, bad ? "NOT ORDERED" : "ordered");
            free(h); CK(hipFree(local));
            if (write(wr, &c, 1) != 1) return 5;
        }
    }
    fflush(stdout);
    if (isA) { int stt; waitpid(pid, &stt, 0); }
    return 0;
}

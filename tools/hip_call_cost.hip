// Host cost of the HIP calls the native sweep makes (GPU box): plain launch, launch with a stop event on the dispatch packet
// (hipExtLaunchKernelGGL), hipStreamWaitEvent, hipEventRecord, hipEventQuery -- mean microseconds per call over 2000 calls
// while the GPU keeps up (tiny kernels).
//   hipcc --offload-arch=gfx950 -O3 -o hip_call_cost tools/hip_call_cost.hip && ./hip_call_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void k_nop(int *p) { if (p && threadIdx.x == 12345) *p = 1; }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t a, b;
    (void)hipStreamCreateWithFlags(&a, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    hipEvent_t ev[4];
    for (auto &e : ev) (void)hipEventCreate(&e);
    const int n = 2000;
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now_us();
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, a, (int *)nullptr);
        double t1 = now_us();
        (void)hipStreamSynchronize(a);
        for (int i = 0; i < n; i++) hipExtLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, a, nullptr, ev[i & 1], 0, (int *)nullptr);
        double t2 = now_us();
        (void)hipStreamSynchronize(a);
        for (int i = 0; i < n; i++) hipExtLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, a, ev[2], ev[3], 0, (int *)nullptr);
        double t3 = now_us();
        (void)hipStreamSynchronize(a);
        double t4 = now_us();
        for (int i = 0; i < n; i++) (void)hipEventRecord(ev[i & 1], a);
        double t5 = now_us();
        (void)hipStreamSynchronize(a);
        (void)hipEventRecord(ev[0], a);
        (void)hipStreamSynchronize(a);
        double t6 = now_us();
        for (int i = 0; i < n; i++) (void)hipStreamWaitEvent(b, ev[0], 0);
        double t7 = now_us();
        (void)hipStreamSynchronize(b);
        double t8 = now_us();
        for (int i = 0; i < n; i++) (void)hipEventQuery(ev[0]);
        double t9 = now_us();
        for (int i = 0; i < n; i++) (void)hipEventSynchronize(ev[0]);
        double t10 = now_us();
        if (rep)
            printf("plain launch %.2f us | launch + stop event %.2f | launch + start and stop events %.2f | hipEventRecord %.2f | "
                   "hipStreamWaitEvent (completed event) %.2f | hipEventQuery %.2f | hipEventSynchronize (completed) %.2f\n",
                   (t1 - t0) / n, (t2 - t1) / n, (t3 - t2) / n, (t5 - t4) / n, (t7 - t6) / n, (t9 - t8) / n, (t10 - t9) / n);
    }
    return 0;
}

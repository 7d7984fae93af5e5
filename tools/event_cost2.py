"""hipStreamWriteValue32 / hipStreamWaitValue32 as the cross-stream hand-over, next to events (tools/event_cost.py)."""
import ctypes as C, os, importlib.util, torch
spec = importlib.util.find_spec("torch")
hip = C.CDLL(os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so"))
dev = torch.device("cuda", 0)
x = torch.zeros(1 << 24, device=dev)      # ~25 us per add_: the loop is GPU-bound, not host-bound
y = torch.zeros(1 << 16, device=dev)
main, side = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
ptr = C.c_void_p()
hipMallocSignalMemory = 0x2
rc = hip.hipExtMallocWithFlags(C.byref(ptr), C.c_size_t(8), C.c_uint(hipMallocSignalMemory))
print("hipExtMallocWithFlags rc", rc, hex(ptr.value or 0))
hip.hipMemset(ptr, 0, C.c_size_t(8))
hip.hipStreamWriteValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint]
hip.hipStreamWaitValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint, C.c_uint32]
GTE = 0   # hipStreamWaitValueGte
val = [0]
def run(mode, n=400):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(main)
    for k in range(n):
        with torch.cuda.stream(main):
            x.add_(1.0)
        if mode == "write":
            val[0] += 1
            rc = hip.hipStreamWriteValue32(C.c_void_p(main.cuda_stream), ptr, val[0], 0)
        elif mode == "write+sidewait":
            val[0] += 1
            hip.hipStreamWriteValue32(C.c_void_p(main.cuda_stream), ptr, val[0], 0)
            hip.hipStreamWaitValue32(C.c_void_p(side.cuda_stream), ptr, val[0], GTE, 0xffffffff)
        elif mode == "write+sidewait+sidekernel":
            val[0] += 1
            hip.hipStreamWriteValue32(C.c_void_p(main.cuda_stream), ptr, val[0], 0)
            hip.hipStreamWaitValue32(C.c_void_p(side.cuda_stream), ptr, val[0], GTE, 0xffffffff)
            with torch.cuda.stream(side):
                y.add_(1.0)
        elif mode == "event+sidewait+sidekernel":
            ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
            with torch.cuda.stream(side):
                y.add_(1.0)
        elif mode == "event+sidewait":
            ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
        elif mode == "event":
            ev = torch.cuda.Event(); ev.record(main)
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for mode in ("plain", "event", "event+sidewait", "event+sidewait+sidekernel", "write", "write+sidewait", "write+sidewait+sidekernel"):
    run(mode, 50)
    print(f"{mode:28s} {run(mode):7.2f} us per iteration")

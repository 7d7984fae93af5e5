"""What an event record / a wait on an event costs a HIP stream (in-stream time), next to back-to-back small kernels."""
import time, torch
dev = torch.device("cuda", 0)
x = torch.zeros(1 << 16, device=dev)
main, side = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def run(mode, n=400):
    evs = [torch.cuda.Event() for _ in range(n)]
    done = torch.cuda.Event(); done.record(side); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(main):
        e0.record(main)
        for k in range(n):
            x.add_(1.0)
            if mode == "record": evs[k].record(main)
            elif mode == "wait_done": main.wait_event(done)
            elif mode == "record+sidewait": evs[k].record(main); side.wait_event(evs[k])
            elif mode == "pingpong":            # main -> side kernel -> main
                evs[k].record(main); side.wait_event(evs[k])
                with torch.cuda.stream(side): x.add_(1.0)
                done2 = torch.cuda.Event(); done2.record(side); main.wait_event(done2)
        e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for mode in ("plain", "record", "wait_done", "record+sidewait", "pingpong"):
    run(mode, 50)
    print(f"{mode:18s} {run(mode):7.2f} us per iteration")

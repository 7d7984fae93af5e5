"""Kernels and memory copies of one rank (rocprofv3 --kernel-trace --memory-copy-trace, csv) on one clock: the last users' half-sweep
with its chunks' row kernels and the peer copies of the exchange, and how much of every device-to-device copy ran while a row kernel
of a LATER chunk was running.  usage: peer_overlap_dump.py <rocprof output dir>"""
import csv, glob, sys
d = sys.argv[1]
kt = glob.glob(d + "/*/*kernel_trace.csv")
mt = glob.glob(d + "/*/*memory_copy_trace.csv")
if not kt or not mt:
    print("no traces under", d, kt, mt); sys.exit(0)
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:46]) for r in csv.DictReader(open(kt[0]))]
M = []
rd = csv.DictReader(open(mt[0]))
print("memory-copy-trace columns:", rd.fieldnames)
for r in rd:
    size = next((int(v) for k, v in r.items() if k and ("byte" in k.lower() or "size" in k.lower()) and str(v).isdigit()), 0)
    M.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?"), size))
big = [m for m in M if "DEVICE_TO_DEVICE" in m[2].upper() and (m[3] >= (4 << 20) or (m[3] == 0 and m[1] - m[0] > 20_000))]        # (this rocprofv3 writes no size column: by duration)
# (a copy out of an IPC mapping of the SAME device may be done by the runtime's copy kernel instead of an SDMA engine: a kernel then)
blit = [(s, e, "copy kernel", 0) for (s, e, n) in K if "copyBuffer" in n and e - s > 100_000]
print(f"{len(big)} large device-to-device copy records, {len(blit)} runtime copy kernels of > 100 us")
big = sorted(big + blit)
print(f"{len(K)} kernels, {len(M)} copies, {len(big)} device-to-device copies of >= 4 MiB or > 20 us (the peer copies)")
rows = [k for k in K if k[2].startswith("k_rows") or k[2].startswith("k_rowmat") or k[2].startswith("k_lr_prep")]
if not big:
    for m in M[-10:]: print(m)
    sys.exit(0)
# overlap of every peer copy with row kernels
tot = ov = 0
for (s, e, _, b) in big:
    o = sum(max(0, min(e, ke) - max(s, ks)) for (ks, ke, _) in rows)
    tot += e - s; ov += min(o, e - s)
print(f"peer copies: {tot / 1e6:.2f} ms in all, {ov / 1e6:.2f} ms of it ({100.0 * ov / max(tot, 1):.0f} %) while a row kernel was running")
# the last 12 peer copies with what ran around them
t0 = big[-12][0] if len(big) >= 12 else big[0][0]
t0 -= 3_000_000
ev = [(s, e, "KERNEL " + n) for (s, e, n) in K if e >= t0] + [(s, e, f"COPY   device-to-device (peer copy out of the other rank's IPC mapping)") for (s, e, _, b) in big if e >= t0]
ev.sort()
base = ev[0][0]
for (s, e, n) in ev:
    if (e - s) < 20_000 and n.startswith("KERNEL") and not n.startswith("KERNEL k_rows"): continue       # (small kernels left out)
    print(f"{(s - base) / 1e3:10.1f} -> {(e - base) / 1e3:10.1f} us  {(e - s) / 1e3:9.1f} us  {n}")

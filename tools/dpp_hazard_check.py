"""Static check of the hand-scheduled DPP hazards: the row kernels' v_fmac_f64_dpp / v_mov_b64_dpp are inline
assembly, which the compiler's hazard recogniser does not see into, and several of them drop the `s_nop 1` on the strength of
"the DPP source was written long before".  gfx950 needs 2 wait states between a vector instruction's write of a VGPR and a DPP
read of it (and the full pass count after a matrix instruction).  This walks the device assembly the build leaves in
csrc/*.s (make: hipcc --cuda-device-only -S) and fails if a straight-line predecessor within that window writes the register a
DPP instruction takes its broadcast from, or writes EXEC from the vector pipe (5 wait states).  Predecessors are followed across
labels: the fall-through block and the tail of every block that branches to the label (loop back-edges included).
   python tools/dpp_hazard_check.py [file.s ...]     exit code 1 and a listing on a violation"""
import glob, os, re, sys

def regs(op):
    """VGPR numbers named by an operand: v12, v[12:13]"""
    op = op.strip().lstrip("-|").rstrip("|")
    m = re.fullmatch(r"v(\d+)", op)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()

BRANCH = re.compile(r"s_c?branch\w*\s+(\.?\w+)")
WINDOW = 24                     # instructions kept per history: more than the longest window (18 wait states)

def parse(line):
    t = line.split(";")[0].strip()
    if not t or t.startswith("//") or (t.startswith(".") and not t.endswith(":")):
        return None
    if t.endswith(":"):
        return ("label", t[:-1], None, None)
    parts = t.split(None, 1)
    mn = parts[0]
    ops = [o.strip() for o in re.split(r",(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
    # operands after the last register/literal are modifiers separated by spaces: keep the first token of each
    ops = [o.split()[0] if o else o for o in ops]
    return (mn, t, ops, None)

def entry(mn, ops):
    """(mnemonic, VGPRs written, wait states provided, writes EXEC from the vector pipe)"""
    if mn == "s_nop":
        return (mn, set(), int(ops[0], 0) + 1 if ops else 1, False)
    if mn.startswith("v_"):
        return (mn, regs(ops[0]) if ops else set(), 1, mn.startswith("v_cmpx") or (bool(ops) and ops[0] in ("exec", "exec_lo", "exec_hi")))
    return (mn, set(), 1, False)

def check(path):
    """Every DPP read against EVERY path that can reach it inside its window: the straight-line predecessors, and -- across a
    label -- the tail of each block that branches to that label (back-edges of loops included), so that a DPP instruction at a
    loop head is checked against the end of the previous iteration.  Rules: a vector write of the DPP source needs 2 wait states,
    a matrix write 18, a vector write of EXEC (v_cmpx, v_readlane-style writes to exec) 5."""
    lines = [parse(l) for l in open(path)]
    # pass 1: for every label, the tails of the blocks that branch to it
    tails, hist = {}, []
    for rec in lines:
        if rec is None:
            continue
        mn, t, ops, _ = rec
        if mn == "label":
            continue
        m = BRANCH.match(t)
        hist.append(entry(mn, ops))
        hist = hist[-WINDOW:]
        if m:
            tails.setdefault(m.group(1), []).append(list(hist))
    # pass 2
    bad, n_dpp = [], 0
    hists = [[]]
    since_label = 10 ** 9
    for ln, rec in enumerate(lines, 1):
        if rec is None:
            continue
        mn, t, ops, _ = rec
        if mn == "label":
            hists = [hists[0]] + [list(h) for h in tails.get(t, [])]
            since_label = 0
            continue
        if mn.endswith("_dpp") and len(ops) >= 2:
            n_dpp += 1
            src = regs(ops[1])
            for h in hists:
                ws, hit = 0, None
                for pm, pdst, pws, pexec in reversed(h):
                    if ws >= 18:
                        break
                    need = 18 if pm.startswith("v_mfma") else 2
                    if pdst & src and ws < need:
                        hit = (path, ln, t, pm, ws, need)
                        break
                    if pexec and ws < 5:
                        hit = (path, ln, t, pm + " (writes EXEC)", ws, 5)
                        break
                    ws += pws
                if hit:
                    bad.append(hit)
                    break
        e = entry(mn, ops)
        for h in hists:
            h.append(e)
            del h[:-WINDOW]
        since_label += 1
        if since_label > WINDOW and len(hists) > 1:
            hists = hists[:1]           # the alternatives have converged inside the window
    return bad, n_dpp

if __name__ == "__main__":
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesiandatafusion.jl_amd", "csrc")
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(root, "*.s")))
    total, allbad = 0, []
    for f in files:
        bad, n = check(f)
        total += n
        allbad += bad
        print(f"{os.path.basename(f)}: {n} DPP instructions, {len(bad)} inside a hazard window")
    for b in allbad[:40]:
        print("  %s:%d  %s   <- %s %d wait state(s) before, needs %d" % b)
    sys.exit(1 if allbad or total == 0 else 0)

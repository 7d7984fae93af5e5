"""Static check of the hand-scheduled DPP hazards: the row kernels' v_fmac_f64_dpp / v_mov_b64_dpp are inline
assembly, which the compiler's hazard recogniser does not see into, and several of them drop the `s_nop 1` on the strength of
"the DPP source was written long before".  gfx950 needs 2 wait states between a vector instruction's write of a VGPR and a DPP
read of it (and the full pass count after a matrix instruction).  This walks the device assembly the build leaves in
csrc/*.s (make: hipcc --cuda-device-only -S) and fails if a straight-line predecessor within that window writes the register a
DPP instruction takes its broadcast from.  (Predecessors across a label are not followed: the kernels have no branch in front of
such an instruction -- their DPP runs are branch-free by construction -- and a label resets the window.)
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

def check(path):
    bad, n_dpp = [], 0
    hist = []                       # (mnemonic, dst regs, wait states it provides)
    for ln, line in enumerate(open(path), 1):
        t = line.split(";")[0].strip()
        if not t or t.startswith(".") or t.startswith("//"):
            continue
        if t.endswith(":"):
            hist = []
            continue
        parts = t.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in re.split(r",(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
        # operands after the last register/literal are modifiers separated by spaces: keep the first token of each
        ops = [o.split()[0] if o else o for o in ops]
        if mn.endswith("_dpp") and len(ops) >= 2:
            n_dpp += 1
            src = regs(ops[1])
            ws = 0
            for pm, pdst, pws in reversed(hist):
                need = 18 if pm.startswith("v_mfma") else 2
                if ws >= 18:
                    break
                if pdst & src and ws < need:
                    bad.append((path, ln, t, pm, ws, need))
                    break
                ws += pws
        if mn == "s_nop":
            hist.append((mn, set(), int(ops[0], 0) + 1 if ops else 1))
        elif mn.startswith("v_"):
            hist.append((mn, regs(ops[0]) if ops else set(), 1))
        else:
            hist.append((mn, set(), 1))
        if len(hist) > 40:
            hist = hist[-40:]
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

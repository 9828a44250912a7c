#!/usr/bin/env python3
"""Wait-state check for the inline-asm VALU instructions of the HIP sources (make check-hazards).

hipcc's hazard recogniser pads wait states only around instructions it selected itself; the operands of an asm
statement are invisible to it.  This scan compiles a source to gfx950 assembly and walks every basic block, reporting an
asm VALU instruction that

  T  reads a register written by a transcendental (v_exp / v_rcp / v_rsq / v_sqrt / v_log / v_sin / v_cos) in the
     instruction directly before it             (gfx950: one wait state between a transcendental and its VALU reader)
  M  reads or overwrites the destination of an MFMA issued fewer than MFMA_STATES wait states earlier (MFMA_STATES_32
     behind a 32x32 form)
                                                (matrix-pipe write -> VALU read / write)
  W  overwrites the C operand of an MFMA issued fewer than SRC_STATES wait states earlier (A and B are read at issue)
  D  is a DPP instruction reading a register a VALU instruction wrote fewer than two wait states earlier
  S  is a v_fma_mixlo/hi_f16 whose fp32 operand is not (a copy of) the operand its f16 hi half was converted from
  A  writes a register an MFMA reads as an operand fewer than two wait states later
  X  (any MFMA, the compiler's own included) takes as an operand the result of an MFMA of a DIFFERENT form issued fewer
     than MFMA_STATES wait states earlier: gfx950 forwards an accumulator between two MFMAs of one form only, and hipcc
     (ROCm 7.2) pads nothing between v_mfma_f32_16x16x32_f16 and a dependent v_mfma_f32_16x16x16_f16 in either order
     (tools/ubench/mixed_mfma_repro.hip: half the outputs wrong, from plain builtins)

Round 3's f16-product featurizer bug was a T (a v_cvt_pk_f16_f32 asm straight behind v_exp_f32); the conversion is now
left to the compiler, and this scan keeps the remaining asm statements (GELU tail, column sums, the lo halves of the
split) honest after every scheduling change.  Exit status 1 on any finding.

usage: tools/hazard_scan.py <source.hip> [extra hipcc flags]"""
import re
import subprocess
import sys
import tempfile

# v_mfma_f32_16x16x32_f16 is a 4-pass instruction on gfx950; hipcc's own padding for it (read off its output: s_nop
# before a v_cvt_pkrtz that rewrites an accumulator) is 8 wait states from the MFMA to a VALU write or read of its
# result and 4 to a VALU overwrite of its C operand.  One instruction of any kind = one wait state, as hipcc counts.
MFMA_STATES = 8
MFMA_STATES_32 = 20  # the 32x32 forms run twice to four times as many passes: the table's worst case (16 passes + 2), + 2
SRC_STATES = 4
MFMA_PIPE = 1


def mfma_states(text):
    return MFMA_STATES_32 if "_32x32x" in text.split()[0] else MFMA_STATES
TRANS = ("v_exp_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_sin_", "v_cos_")


def regs(tok):
    tok = tok.strip().lstrip("-").strip("|")
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def operands(text):
    parts = text.split(None, 1)
    if len(parts) < 2:
        return []
    body = re.split(r"\s+(?:row_|op_sel|bank_mask|quad_perm|offset|off\b)", parts[1])[0]
    return [o for o in (x.strip() for x in body.split(",")) if o]


def states(text):
    """Issue slots an instruction puts between its neighbours: s_nop N is N + 1; an MFMA holds the matrix pipe for at
    least four passes, and the next instruction of the wave behind a second MFMA waits for that (a lower bound)."""
    m = re.match(r"s_nop (\d+)", text)
    return int(m.group(1)) + 1 if m else 1


def scan(asm_text):
    findings = []
    kernel = "?"
    block = []  # (line, text, in_asm)
    in_asm = False

    def flush():
        for k, (ln, t, a) in enumerate(block):
            if t.startswith("v_mfma"):
                ops = operands(t)
                srcs = set().union(*[regs(x) for x in ops[1:]]) if len(ops) > 1 else set()
                dist = 0
                for j in range(k - 1, -1, -1):
                    pl, pt, pa = block[j]
                    if pa and pt.startswith("v_") and dist < 2 and regs(operands(pt)[0]) & srcs:
                        findings.append(f"A {kernel}: line {pl} '{pt}' feeds the MFMA at line {ln} {dist} wait states later")
                    dist += states(pt)
                    if dist >= 2:
                        break
                dist = 0
                for j in range(k - 1, -1, -1):
                    pl, pt, pa = block[j]
                    if pt.startswith("v_mfma"):
                        if pt.split()[0] != t.split()[0] and dist < mfma_states(pt) and regs(operands(pt)[0]) & srcs:
                            findings.append(f"X {kernel}: line {ln} '{t}' takes the result of '{pt}' (line {pl}), another "
                                            f"MFMA form, {dist} wait states behind it")
                        break  # a third MFMA further up has left the pipe by the time this one issues
                    dist += states(pt)
                    if dist >= MFMA_STATES_32:
                        break
            if a and t.startswith(("v_fma_mixlo_f16", "v_fma_mixhi_f16")):
                # S: the lo half must be computed from the operand its hi half was converted from
                #    (v_cvt_pkrtz H, A, B ... v_fma_mixlo L, -H, 1.0, A' / v_fma_mixhi L, -H, 1.0, B' with A' = A, B' = B
                #    or plain copies of them): this hipcc has tied an asm operand to the wrong vector element before
                mo = operands(t)
                hreg, sreg = mo[1].lstrip("-"), mo[3]
                want = None
                for j in range(k - 1, -1, -1):
                    pt = block[j][1]
                    po = operands(pt)
                    if po and po[0] == hreg and not pt.startswith("v_fma_mix"):
                        if pt.startswith("v_cvt_pkrtz_f16_f32"):
                            want = (po[1] if "mixlo" in t else po[2], j)
                        break
                if want:
                    src, jdef = want
                    cur = sreg
                    for j in range(k - 1, -1, -1):  # follow the copies of the operand back past the conversion
                        pt = block[j][1]
                        po = operands(pt)
                        if j <= jdef and cur == src:
                            break
                        if po and po[0] == cur and not pt.startswith(("v_cmp", "global_store", "ds_write", "scratch_store")):
                            if pt.startswith("v_mov_b32_e32"):
                                cur = po[1]
                            elif pt.startswith("v_fma_mixlo_f16") and block[j][2] and po[3] == cur:
                                continue  # the in-place mixlo of the same pair kept the register
                            else:
                                if j > jdef:
                                    cur = None
                                break
                    ok = cur == src
                    if not ok:
                        findings.append(f"S {kernel}: line {ln} '{t}' does not read the operand of the conversion at line "
                                        f"{block[jdef][0]} ('{block[jdef][1]}')")
            if not a or not t.startswith("v_"):
                continue
            ops = operands(t)
            used = set().union(*[regs(x) for x in ops]) if ops else set()
            reads = set().union(*[regs(x) for x in ops[1:]]) if len(ops) > 1 else set()
            if "_dpp" in t or "row_" in t:
                reads |= regs(ops[0])
            if k:
                pl, pt, pa = block[k - 1]
                if pt.startswith(TRANS) and regs(operands(pt)[0]) & reads:
                    findings.append(f"T {kernel}: line {ln} '{t}' directly behind '{pt}'")
            dist = 0
            live = set(used)  # registers whose value still comes from further up (not redefined in between)
            for j in range(k - 1, -1, -1):
                pl, pt, pa = block[j]
                if pt.startswith("v_mfma"):
                    mops = operands(pt)
                    if regs(mops[0]) & live and dist < mfma_states(pt):
                        findings.append(f"M {kernel}: line {ln} '{t}' {dist} wait states behind '{pt}' (line {pl})")
                    elif dist < SRC_STATES and regs(ops[0]) & live & regs(mops[-1]):
                        findings.append(f"W {kernel}: line {ln} '{t}' overwrites the C operand of '{pt}' (line {pl}) "
                                        f"{dist} wait states behind it")
                if ("_dpp" in t or "row_" in t) and dist < 2 and pt.startswith("v_") and not pt.startswith("v_mfma") \
                        and regs(operands(pt)[0]) & reads:
                    findings.append(f"D {kernel}: line {ln} '{t}' {dist} wait states behind '{pt}' (line {pl})")
                if not pa and not pt.startswith(("v_mfma", "s_")):
                    # read, stored or redefined here by an instruction hipcc selected (and padded): whatever wrote the
                    # register further up has completed
                    for x in operands(pt):
                        live -= regs(x)
                dist += states(pt) if not pt.startswith("v_mfma") or j == k - 1 else MFMA_PIPE
                if dist >= MFMA_STATES_32:
                    break
        block.clear()

    for i, raw in enumerate(asm_text.split("\n")):
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if t.endswith(":") or re.match(r"^[A-Za-z_.$][\w.$]*:", t):
            flush()
            if raw.startswith("_Z"):
                kernel = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", t.split(":")[0])[:48]
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        block.append((i + 1, t, in_asm))
        if t.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            flush()
    flush()
    return findings


def main():
    src, extra = sys.argv[1], sys.argv[2:]
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Iinclude", "-Impnn_amd/csrc",
                        "-S", "--cuda-device-only", "-o", f.name, src] + extra, check=True, stderr=subprocess.DEVNULL)
        text = open(f.name).read()
    found = scan(text)
    n_asm = text.count(";;#ASMSTART")
    for line in found[:40]:
        print(line)
    print(f"{src} {' '.join(extra)}: {n_asm} asm statements, {len(found)} findings")
    sys.exit(1 if found else 0)


if __name__ == "__main__":
    main()

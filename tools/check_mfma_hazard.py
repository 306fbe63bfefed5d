"""CPU-side guard for the float64 MFMAs that csrc/ta_lstm_f64.hip issues as INLINE ASSEMBLY.

The recurrence kernel names AGPR-resident weights in the matrix instruction itself, which the compiler's own
`__builtin_amdgcn_mfma_f64_16x16x4f64` does not do (it copies such an operand out with two `v_accvgpr_read_b32` per MFMA:
a quarter of the MFMA time, profiles/r04_mfma_f64_ops.txt).  To the compiler an asm statement is opaque, so its hazard
recogniser inserts none of the wait states a matrix instruction needs; the kernel provides them (`mfma_begin`,
`mfma_settle`).  This script disassembles the gfx950 code objects inside libta_hip.so and fails if, for any
`v_mfma_f64_16x16x4_f64`,

  (a) an instruction other than the next MFMA of the same accumulator chain touches (reads or writes) one of its result
      registers before 18 wait states have passed (16 passes; what the compiler leaves after the builtin), or
  (b) a VALU instruction writes one of its source registers fewer than 2 wait states before it (the compiler's distance
      between such a write and the builtin).

An instruction counts one wait state, `s_nop N` counts N + 1 (matrix instructions in between count one each although they
take longer: the check errs on the strict side).  Run by tests/test_abi.py (no GPU needed).

Usage: python tools/check_mfma_hazard.py [libta_hip.so]   (exit code 1 and a listing on a finding)"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.check_store_hazard import code_objects, disassemble   # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MFMA = "v_mfma_f64_16x16x4_f64"
SETTLE = 18          # wait states between the last MFMA of a chain and the first use of its result
BEGIN = 2            # wait states between a VALU write of a source and the MFMA
_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def _regs(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), k) for k in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def findings(text):
    out = []
    kernel = "?"
    pending = []          # results in flight: [wait states so far, regs, dst operand text, description]
    recent = []           # VALU writes: [wait states since, regs, description]
    nmfma = 0
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line.strip())
        if m:
            kernel, pending, recent = m.group(1), [], []
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//", line)
        if not m:
            continue
        mnem, rest = m.group(1), m.group(2)
        ops = [o.strip() for o in rest.split(",")] if rest else []
        touched = _regs(rest)
        if mnem == MFMA:
            nmfma += 1
            dst, srcs = ops[0], ops[1:4]
            # (b) a source written by a VALU instruction just before
            for since, regs, desc in recent:
                hit = regs & _regs(" ".join(srcs))
                if hit and since < BEGIN:
                    out.append("%s: `%s` wrote %s %d wait state(s) before `%s %s`" % (kernel, desc, sorted(hit), since, mnem, rest))
            # (a) results in flight: only the next MFMA of the same chain (same dst, used as srcC only) may follow
            keep = []
            for p in pending:
                hit = p[1] & touched
                chain = dst == p[2] and len(ops) > 3 and ops[3] == p[2] and not (p[1] & _regs(" ".join(ops[1:3])))
                if hit and not chain:
                    out.append("%s: `%s %s` touches %s, result of `%s`, after %d wait state(s)" % (kernel, mnem, rest, sorted(hit)[:4], p[3], p[0]))
                if not hit:
                    keep.append(p)
            pending = keep
            pending.append([0, _regs(dst), dst, "%s %s" % (mnem, rest)])
        else:
            for p in pending:
                hit = p[1] & touched
                if hit and p[0] < SETTLE:
                    out.append("%s: `%s %s` touches %s, result of `%s`, after %d wait state(s)" % (kernel, mnem, rest, sorted(hit)[:4], p[3], p[0]))
        cost = int(ops[0], 0) + 1 if mnem == "s_nop" and ops else 1
        # the instruction itself has now issued: everything before it is one (or N + 1) wait states older
        for p in pending:
            if not (mnem == MFMA and p is pending[-1]):
                p[0] += cost
        pending = [p for p in pending if p[0] < SETTLE]
        for r in recent:
            r[0] += cost
        recent = [r for r in recent if r[0] < BEGIN]
        if mnem.startswith("v_") and mnem != MFMA and ops and not mnem.startswith(("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane")):
            recent.append([0, _regs(ops[0]), "%s %s" % (mnem, rest)])
    return out, nmfma


def check(so_path):
    blobs = code_objects(so_path)
    if not blobs:
        raise RuntimeError("no gfx950 code object found in %s" % so_path)
    found, nmfma = [], 0
    for blob in blobs:
        f, n = findings(disassemble(blob))
        found += f
        nmfma += n
    return found, nmfma, len(blobs)


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "text_alignment_amd", "libta_hip.so")
    found, nmfma, nco = check(path)
    for f in found:
        print(f)
    print("%d f64 MFMAs in %d code object(s), %d finding(s)" % (nmfma, nco, len(found)))
    sys.exit(1 if found else 0)

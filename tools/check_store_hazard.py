"""CPU-side guard for the 16-byte buffer-store data hazard of DESIGN.md section 4.4 (6a).

On gfx950 a `buffer_store_dwordx4` (and x3) reads its data VGPRs late: with the offset in an SGPR
(`soffset`) hipcc's hazard recogniser does not cover it, and a VALU write to one of the data
registers in the next two issue slots is what reaches memory (seen in round 2 as ~12 % wrong first
words when a second workgroup shared the CU).  The kernels therefore keep the whole offset in the
VGPR (+ immediate), the form the recogniser does cover.  Nothing but a numerically wrong batch would
notice a compiler update that changes either side, so this script disassembles the gfx950 code
objects inside libta_hip.so and fails if

  (a) any `buffer_store_dwordx3/x4` carries a non-null SGPR `soffset`, or
  (b) any such store is followed within two issue slots (instructions; `s_nop N` counts N + 1) by a
      VALU instruction that writes one of its data VGPRs.

Usage: python tools/check_store_hazard.py [libta_hip.so]   (exit code 1 and a listing on a finding)
Run by tests/test_abi.py (no GPU needed)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so_path):
    """gfx950 code objects (bytes) embedded in the .hip_fatbin section of a host shared library."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so_path])
        with open(fat, "rb") as f:
            data = f.read()
    out = []
    pos = data.find(MAGIC)
    while pos >= 0:
        (count,) = struct.unpack_from("<Q", data, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                out.append(data[pos + off:pos + off + size])
        pos = data.find(MAGIC, pos + len(MAGIC))
    return out


def disassemble(blob):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(blob)
        f.flush()
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", f.name],
                                       text=True)


_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _vregs(operand):
    regs = set()
    for m in _REG.finditer(operand):
        if m.group(1) is not None:
            regs.add(int(m.group(1)))
        else:
            regs.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return regs


def _writes(mnemonic, operands):
    """VGPRs a VALU instruction writes (its first operand).  Only VALU writes can hit the hazard: the
    store's data is read a few cycles after issue, long before a later vector-memory or LDS load (which
    queues behind the store in the same in-order memory path, or returns tens of cycles later) lands."""
    if not operands or not mnemonic.startswith("v_"):
        return set()
    if mnemonic.startswith(("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane")):
        return set()
    return _vregs(operands[0])


def findings(text):
    out = []
    kernel = "?"
    window = []           # [(slots_left, data regs, description)]
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line.strip())
        if m:
            kernel = m.group(1)
            window = []
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//", line)
        if not m:
            continue
        mnem, rest = m.group(1), m.group(2)
        ops = [o.strip() for o in rest.split(",")] if rest else []
        # (b) a write to a pending store's data registers
        wr = _writes(mnem, ops)
        for slots, regs, desc in window:
            hit = wr & regs
            if hit:
                out.append("%s: `%s %s` writes v%s within two issue slots of `%s`" % (kernel, mnem, rest, sorted(hit), desc))
        cost = 1
        if mnem == "s_nop":
            cost = int(ops[0], 0) + 1 if ops else 1
        window = [(s - cost, r, d) for s, r, d in window if s - cost > 0]
        if mnem in ("buffer_store_dwordx4", "buffer_store_dwordx3"):
            # operands: vdata, vaddr|off, srsrc, soffset [offen] [offset:n] ...
            soff = ops[3].split()[0] if len(ops) > 3 else "0"
            if soff not in ("0", "null") and re.match(r"^(s\d+|s\[|m0|ttmp)", soff):
                out.append("%s: `%s %s` carries an SGPR soffset (%s)" % (kernel, mnem, rest, soff))
            window.append((2, _vregs(ops[0]), "%s %s" % (mnem, rest)))
    return out


def check(so_path):
    blobs = code_objects(so_path)
    if not blobs:
        raise RuntimeError("no gfx950 code object found in %s" % so_path)
    found, nstores = [], 0
    for blob in blobs:
        text = disassemble(blob)
        nstores += len(re.findall(r"\bbuffer_store_dwordx[34]\b", text))
        found += findings(text)
    return found, nstores, len(blobs)


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "text_alignment_amd", "libta_hip.so")
    bad, n, nco = check(path)
    print("%d gfx950 code objects, %d buffer_store_dwordx3/x4 instructions, %d findings" % (nco, n, len(bad)))
    for b in bad:
        print("  " + b)
    sys.exit(1 if bad else 0)

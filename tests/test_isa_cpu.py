"""CPU (hipcc cross-compiles gfx950 here): properties of the generated ISA that the source cannot express.

attention_f16.hip issues its LDS fragment reads as inline asm (the compiler would otherwise serialise them behind every
LDS-direct load in flight) and waits for them with an explicit s_waitcnt.  The compiler does not know the reads are
asynchronous: if register pressure ever made it copy or reuse a destination register between the read and the wait, the
kernel would compute on garbage (this happened in a 512-register lab variant: profiles/r04/attn_f16_ablation.txt).  The
check walks the compiled kernel and demands that no instruction touches a destination of an in-flight read."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pope_amd", "csrc")


def _regs(tok):
    tok = tok.strip().split()[0] if tok.strip() else ""
    m = re.match(r"[va]\[(\d+):(\d+)\]", tok)
    if m:
        return {(tok[0], i) for i in range(int(m.group(1)), int(m.group(2)) + 1)}
    m = re.match(r"([va])(\d+)$", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def test_attention_f16_asm_reads_are_not_touched_before_their_wait():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "a.s")
        flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only"]
        res = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "attention_f16.hip", "-o", out], cwd=CSRC, capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-2000:]
        text = open(out).read()
    body = text[text.index("attn_f16_dma_kernel"):]
    meta = re.search(r"\.vgpr_count:\s+(\d+)", body)
    spill = re.search(r"\.vgpr_spill_count:\s+(\d+)", body)
    assert meta and int(meta.group(1)) < 230, "the kernel must stay well below the register limit (asm reads need their destinations to stay put)"
    assert spill and int(spill.group(1)) == 0
    pending, reads, waits = set(), 0, 0
    for line in body.split("\n"):
        t = line.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        args = t[len(op):].split(",")
        if op in ("ds_read_b64_tr_b16", "ds_read_b128"):
            pending |= _regs(args[0])
            reads += 1
            continue
        if op == "s_waitcnt" and "lgkmcnt(0)" in t:
            pending = set()
            waits += 1
            continue
        if op == "s_endpgm":
            break
        if pending:
            used = set()
            for a in args:
                used |= _regs(a)
            assert not (used & pending), f"`{t}` touches {sorted(used & pending)[:4]} while their LDS reads are in flight"
    assert reads >= 48 and waits >= 6   # the loop was found: K (8) + V (16) reads per iteration form, several forms

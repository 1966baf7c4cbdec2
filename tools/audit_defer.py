#!/usr/bin/env python3
"""Audit of the deferred-epilogue NT GEMM kernels' ISA (ldmae_amd/csrc/probe/gemm_nt_defer.hip), after every edit of that file:

  1. vector-register spills / scratch traffic: where they sit relative to the K loops (inside = a VMEM instruction the hand-written
     vmcnt counts do not know, and a compiler vmcnt wait that drains the DMA ring);
  2. the registers an inline-asm buffer_load writes are unprotected until the asm `s_waitcnt vmcnt(N)` that retires the load: no
     instruction in between may read or write them (a register copy there would copy data that has not landed);
  3. the vector-memory instructions between every ring DMA group and the counted waits, per K-step (printed, to be compared with the
     NWAIT constants of kstep<>).

    python tools/audit_defer.py            # compiles the file to /tmp/defer.s itself
Linear scan in program order (the kernels are straight-line K-steps with short forward branches around the waits and one two-step
loop): a register pending at the loop's back edge is checked against the code up to the branch."""
import os, re, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "ldmae_amd", "csrc", "probe", "gemm_nt_defer.hip")
out = "/tmp/defer_audit.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-Wno-inline-asm",
                "-Wno-unused-value", "-S", "--cuda-device-only", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
text = open(out).read().split("\n")


def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def all_vregs(line):
    s = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", line.split(";")[0]):
        s |= regs_of(tok)
    return s


ok = True
starts = [i for i, l in enumerate(text) if re.match(r"^_Z20gemm_nt_defer_kernel.*:\s", l)]
for ks in starts:
    name = text[ks].split(":")[0]
    end = next(i for i in range(ks, len(text)) if ".end_amdhsa_kernel" in text[i] or text[i].startswith(".Lfunc_end"))
    body = text[ks:end]
    mf = [i for i, l in enumerate(body) if "v_mfma" in l]
    print(f"== {name}: {len(body)} lines, MFMAs in lines {mf[0]}..{mf[-1]}")
    # 1. scratch
    sc = [(i, l.strip()) for i, l in enumerate(body) if "scratch_" in l]
    inside = [x for x in sc if mf[0] <= x[0] <= mf[-1]]
    print(f"   scratch instructions: {len(sc)} ({len(inside)} between the first and the last MFMA)")
    for i, l in inside:
        print(f"      {i}: {l}")
    # 2. pending asm loads
    in_asm = False
    vmem_seq = []          # (line, kind, dest regs)
    pending = []           # [(index into vmem_seq, regs)]
    bad = 0
    for i, l in enumerate(body):
        s = l.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if s.startswith(";;#ASMEND"):
            in_asm = False; continue
        code = s.split(";")[0].strip()
        if not code or code.endswith(":"):
            continue
        op = code.split()[0]
        is_vmem = op.startswith(("buffer_", "global_", "scratch_", "flat_"))
        if in_asm and op == "s_waitcnt" and "vmcnt" in code:
            n = int(re.search(r"vmcnt\((\d+)\)", code).group(1))
            keep = len(vmem_seq) - n
            pending = [p for p in pending if p[0] >= keep]
            continue
        if not in_asm and op == "s_waitcnt" and "vmcnt" in code:
            n = int(re.search(r"vmcnt\((\d+)\)", code).group(1))
            keep = len(vmem_seq) - n
            pending = [p for p in pending if p[0] >= keep]
            if mf[0] <= i <= mf[-1]:
                print(f"   NOTE compiler vmcnt wait inside the K loops: line {i}: {code}")
            continue
        if is_vmem:
            dest = set()
            if in_asm and op == "buffer_load_dwordx4":
                dest = regs_of(code.split()[1].rstrip(","))
            vmem_seq.append((i, op, dest))
            # operands of this instruction (address / data registers) must not be pending either
            used = all_vregs(code) - dest
            for idx, regs in pending:
                if used & regs:
                    print(f"   HAZARD line {i}: `{code}` uses v{sorted(used & regs)} still pending from line {vmem_seq[idx][0]}"); bad += 1
            if dest:
                pending.append((len(vmem_seq) - 1, dest))
            continue
        used = all_vregs(code)
        for idx, regs in pending:
            if used & regs:
                print(f"   HAZARD line {i}: `{code}` touches v{sorted(used & regs)} pending from the asm load at line {vmem_seq[idx][0]}"); bad += 1
    print(f"   asm-load register hazards: {bad}")
    ok &= bad == 0
    # 3. VMEM sequence per DMA group
    seq = []
    for i, l in enumerate(body):
        code = l.strip().split(";")[0].strip()
        if not code:
            continue
        op = code.split()[0]
        if op == "global_load_lds_dwordx4": seq.append("D")
        elif op == "buffer_load_dwordx4": seq.append("L")
        elif op == "buffer_store_dwordx4": seq.append("S")
        elif op.startswith("scratch_"): seq.append("x")
        elif op.startswith(("global_", "buffer_", "flat_")): seq.append("o")
        elif op == "s_waitcnt" and "vmcnt" in code: seq.append("w" + re.search(r"vmcnt\((\d+)\)", code).group(1))
        elif op == "s_barrier": seq.append("|")
        elif op == "v_mfma_f32_16x16x32_bf16" and (not seq or seq[-1] != "M"): seq.append("M")
    print("   sequence (D dma piece, L/S deferred load/store, x scratch, o other VMEM, wN vmcnt wait, | barrier, M MFMA run):")
    line = "      "
    for t in seq:
        line += t + " "
        if len(line) > 150:
            print(line); line = "      "
    print(line)
print("AUDIT OK" if ok else "AUDIT FAILED")
sys.exit(0 if ok else 1)

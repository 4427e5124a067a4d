"""fdnn_ppo.hip keeps its accumulators in the accumulation registers a0..a159 behind the compiler's back.  The compiler
would use those registers for one thing only -- as spill space for architectural registers -- so the kernel must compile
without such spills.  This check reads the kernel's assembly (hipcc --cuda-device-only -S) and fails if any COMPILER-generated
instruction (anything outside an inline-assembly block) names an accumulation register, or if a kernel does not have exactly
96 + 160 registers.  csrc/Makefile runs it on every build of fdnn_ppo.o.
usage: check_hidden_regs.py file.s"""
import re, sys

path = sys.argv[1]
bad, in_asm, n_asm, kernels, shape_ok = [], False, 0, 0, True
for no, line in enumerate(open(path), 1):
    t = line.strip()
    if "#ASMSTART" in t:
        in_asm = True
        n_asm += 1
        continue
    if "#ASMEND" in t:
        in_asm = False
        continue
    if t.startswith(".amdhsa_next_free_vgpr"):
        kernels += 1
        shape_ok &= t.split()[-1] == "256"
    if t.startswith(".amdhsa_accum_offset"):
        shape_ok &= t.split()[-1] == "96"
    if in_asm or not t or t.startswith((";", ".", "//")):
        continue
    code = t.split(";")[0]
    if re.search(r"\ba(\d+)\b|\ba\[\d+:\d+\]|accvgpr", code):
        bad.append((no, t))
print("%s: %d kernels, %d inline-assembly blocks, %d compiler instructions on accumulation registers, register split %s"
      % (path, kernels, n_asm, len(bad), "96 + 160" if shape_ok else "NOT 96 + 160"))
for no, t in bad[:20]:
    print("  line %d: %s" % (no, t))
sys.exit(1 if bad or kernels == 0 or not shape_ok else 0)

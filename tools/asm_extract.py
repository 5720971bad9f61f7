"""Print one kernel's gfx950 assembly out of a `hipcc -S --cuda-device-only` file: python tools/asm_extract.py file.s <mangled-substring>"""
import sys
lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
print("\n".join(lines[start:end]))

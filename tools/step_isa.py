#!/usr/bin/env python3
"""Instruction mix of one unrolled step of an L2-gather walk, from the compiler's assembly (no GPU needed):
tools/step_isa.py <ISECT> [extra hipcc flags...]   — ISECT 7: quantised culled walk (c5), 9: exact culled walk (mesh)."""
import collections, re, subprocess, sys
isect = sys.argv[1] if len(sys.argv) > 1 else "7"
out = "/tmp/step_isa.s"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-unroll-loops", "-fno-slp-vectorize",
                "-Iinclude", "-Iray_tracer_s8_amd/csrc", "--cuda-device-only", "-S", "ray_tracer_s8_amd/csrc/rt_kernels_trav.hip", "-o", out]
               + sys.argv[2:], check=True, stderr=subprocess.DEVNULL)
name = f"_ZN3rtk14rt_tile_kernelILi{isect}ELb0ELi256ELb0EEEvNS_7KParamsE"
lines, on = [], False
for ln in open(out):
    if ln.startswith(name + ":"):
        on = True
    elif on and ln.lstrip().startswith(".size") and name in ln:
        break
    elif on:
        t = ln.split(";")[0].strip()
        if t and not t.endswith(":") and not t.startswith("."):
            lines.append(t)
# a step starts at the first of its node loads: the 16-byte load WITHOUT an offset that is followed by one with offset:16
starts = [i for i, t in enumerate(lines[:-3]) if t.startswith("global_load_dwordx4") and "offset" not in t and any("offset:16" in u for u in lines[i + 1:i + 4])]
gaps = collections.Counter(b - a for a, b in zip(starts, starts[1:]))
step = gaps.most_common(1)[0][0]
i0 = next(a for a, b in zip(starts, starts[1:]) if b - a == step)
body = lines[i0:i0 + step]
kinds = collections.Counter()
for t in body:
    op = t.split()[0]
    k = ("VALU cvt" if op.startswith("v_cvt") else "VALU cndmask" if op.startswith("v_cndmask") else "VALU cmp" if op.startswith("v_cmp")
         else "VALU min/max" if re.match(r"v_(min|max)", op) else "VALU fma/mul/add" if re.match(r"v_(fma|fmac|mul_f32|add_f32|sub_f32)", op)
         else "VALU other" if op.startswith("v_") else "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_") else "VMEM")
    kinds[k] += 1
print(f"kernel <{isect}>: {len(starts)} unrolled steps, {step} instructions each: " + ", ".join(f"{k} {v}" for k, v in sorted(kinds.items())))
print(f"  VALU total {sum(v for k, v in kinds.items() if k.startswith('VALU'))}")
if "-v" in sys.argv:
    print("\n".join("    " + t for t in body))

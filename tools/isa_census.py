#!/usr/bin/env python3
"""Static instruction census of kernels in a gfx950 assembly listing (hipcc --cuda-device-only -S): per kernel, instructions by class and the most frequent opcodes.
The kernels of this library are straight-line per tile (template-unrolled), so the static mix of the loop body is the dynamic mix to a few percent.
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip --cuda-device-only -S -DMS_ONLY_FIELD=1 mini-stark_amd/csrc/ntt_plan.cpp -o /tmp/ntt_bb.s
  python3 tools/isa_census.py /tmp/ntt_bb.s 'PassKernel2I2BBS3_Lb0ELi10ELi3ELi256ELi3ELi2E' 'PassKernel2I2BBS3_Lb0ELi10ELi4ELi512ELi3ELi1E'"""
import collections, re, sys
txt = open(sys.argv[1]).read().split("\n")
for pat in sys.argv[2:]:
    start = next((i for i, l in enumerate(txt) if l.startswith("_Z") and pat in l and l.rstrip().split(":")[0].endswith("ParamsE")), None)
    if start is None:
        print(pat, ": not found"); continue
    ops = collections.Counter()
    for l in txt[start + 1:]:
        if l.startswith(".Lfunc_end") or l.startswith("\t.section"):
            break
        m = re.match(r"^\t([a-z_0-9]+)\b", l)
        if m and not m.group(1).startswith(("s_nop",)):
            ops[m.group(1)] += 1
    tot = sum(ops.values())
    cls = collections.Counter()
    for o, n in ops.items():
        c = ("valu_mul" if re.match(r"v_(mul_lo|mul_hi|mad_u64|mad_u32|mul_u32)", o) else "valu_other") if o.startswith("v_") else ("lds" if o.startswith("ds_") else ("vmem" if o.startswith(("global_", "buffer_", "flat_", "scratch_")) else ("salu" if o.startswith("s_") else "other")))
        cls[c] += n
    print(f"{pat}: {tot} instructions; by class {dict(cls)}")
    print("   top opcodes:", ", ".join(f"{o} {n}" for o, n in ops.most_common(18)))

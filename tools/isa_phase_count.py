"""Per-phase VALU instruction count of k_deferred_shade<true, 257, false> from the compiler's gfx950 ISA: shade.hip is
compiled once as shipped and once per PBR_EXP_* switch that removes one phase of shade_pixel; the difference of the kernels'
static v_* instruction counts is that phase (straight-line per-pixel code: static = executed; the light walk is counted
separately from its loop body x trips).  No GPU needed.  usage: python tools/isa_phase_count.py [out.md]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "direct12pbrrenderer_amd", "csrc", "shade.hip")
KERNEL = "_Z16k_deferred_shadeILb1ELi257ELb0EEv11ShadeParamsii10ShadeRects"


def kernel_isa(defines):
    with tempfile.TemporaryDirectory() as d:
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable",
               "-I" + os.path.join(ROOT, "include"), "-save-temps=obj", "-c", SRC, "-o", os.path.join(d, "shade.o")] + ["-D" + x for x in defines]
        subprocess.run(cmd, check=True, capture_output=True)
        text = open(os.path.join(d, "shade-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    a = text.index(KERNEL + ":")
    b = text.index(".end_amdhsa_kernel", a)
    return text[a:b].splitlines()


def count(lines):
    valu = [ln.split()[0] for ln in lines if re.match(r"\s+v_", ln)]
    trans = sum(1 for v in valu if re.match(r"v_(rcp|rsq|log|exp|sqrt|sin|cos)_", v))
    pk = sum(1 for v in valu if v.startswith("v_pk_"))
    mov = sum(1 for v in valu if v.startswith("v_mov_") or v.startswith("v_pk_mov"))
    return len(valu), trans, pk, mov


def loops(lines):
    """(label, VALU count, packed, transcendental) of every innermost loop body (a label that a later s_cbranch jumps back to)."""
    out = []
    labels = {ln.split(":")[0]: i for i, ln in enumerate(lines) if re.match(r"\.LBB\d+_\d+:", ln)}
    for i, ln in enumerate(lines):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = lines[labels[m.group(1)]:i + 1]
            if not any(re.match(r"\.LBB\d+_\d+:", b) for b in body[1:]):
                out.append((m.group(1),) + count(body))
    return out


base = kernel_isa([])
total = count(base)
rows = [("kernel as shipped (static, every path)", total)]
variants = [("light walk (cluster index + all four instantiations of the list walk + the sums' zeroing)", ["PBR_EXP_NOLOOP"]),
            ("IBL specular: reflection vector, cube face, two trilinear levels from the footprint layout, their lerps", ["PBR_EXP_NOENV"]),
            ("split-sum LUT fetch + bilinear", ["PBR_EXP_NOLUT"]),
            ("SH9 irradiance (EnvironmentDiffuse)", ["PBR_EXP_NOSH"]),
            ("all of the IBL specular term (env + LUT)", ["PBR_EXP_NOIBL"])]
doc = ["# k_deferred_shade<true, 257, false>: VALU instructions per phase (static count of the gfx950 ISA)", "",
       f"whole kernel: {total[0]} v_* instructions ({total[2]} packed, {total[1]} transcendental, {total[3]} moves)", "",
       "| phase removed (-D switch) | v_* removed | of them packed | transcendental | moves |", "|---|---|---|---|---|"]
for name, defs in variants:
    c = count(kernel_isa(defs))
    doc.append(f"| {name} (`{' '.join(defs)}`) | {total[0] - c[0]} | {total[2] - c[2]} | {total[1] - c[1]} | {total[3] - c[3]} |")
doc += ["", "innermost loops of the shipped kernel (the list walks; one of them runs per pixel, `trips` = pairs of lights):", "",
        "| loop | v_* per trip | packed | transcendental | moves |", "|---|---|---|---|---|"]
for lb, n, t, pk, mv in loops(base):
    doc.append(f"| {lb} | {n} | {pk} | {t} | {mv} |")
text = "\n".join(doc) + "\n"
print(text)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(text)

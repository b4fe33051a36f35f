#!/usr/bin/env python3
"""Per-opcode histogram of the headline render kernel, weighted by how often each phase of its loop runs.

The kernel's loop has phases (shading, starting paths, traversal set-up, node steps, leaf tests, scheduling
decisions); -DRT_ISA_MARKS puts a comment line at each boundary into the generated assembly.  This tool
compiles rt_kernels.hip that way, counts the VALU opcodes between the marks of the loop's first copy
(blocks the compiler placed behind the loop are shading code and are counted there), multiplies by the
executions of each phase per 64 rays that the instrumented kernel reports (tools/sched_stats.py) and
prices every opcode with the issue rates of tools/microbench (cycles per wave64 instruction per SIMD
with four waves resident).  A static count of a phase is an upper bound of what one execution issues
(a branch no lane takes is skipped); with 40 lanes in flight nearly every branch is taken by some lane.

usage: python tools/opcode_hist.py [--lds 1|0] [--node N --leaf N --shade N --start N]  (defaults: bouncing_spheres)
"""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RATES = [  # (regex, class, cycles per wave64 instruction)
    (r"v_(rcp|rsq|sqrt|sin|cos|exp|log)_f32|v_mul_lo_u32|v_mul_hi_u32|v_mad_[iu]64_[iu]32|v_div_scale_f32|v_div_fmas_f32|v_div_fixup_f32", "quarter", 8.5),
    (r"v_(min|max|med3|min3|max3)_|v_cmp_|v_cmpx_|v_cndmask_|v_addc_co|v_subb_co", "half", 4.5),
    (r"v_(fma|fmac|fmamk|fmaak|mad)_", "fma", 3.0),
    (r"v_pk_", "packed", 4.6),
    (r"v_(readlane|writelane|readfirstlane)_", "lane", 4.0),
    (r"v_", "full", 2.4),
]


def classify(op):
    for rx, cls, cyc in RATES:
        if re.match(rx, op):
            return cls, cyc
    return None, 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lds", type=int, default=1, help="1: the kernel of scenes held in LDS, 0: of scenes read from HBM")
    # executions per 64 rays (tools/sched_stats.py, bouncing_spheres): node blocks, leaf blocks, shading phases
    ap.add_argument("--node", type=float, default=5.7)
    ap.add_argument("--leaf", type=float, default=4.0)
    ap.add_argument("--shade", type=float, default=1.5)
    ap.add_argument("--start", type=float, default=1.5)
    ap.add_argument("--lanes", type=str, default="node=37.4,leaf=19.3,shade=42.8,start=42.8,begin=42.8")
    ap.add_argument("--top", type=int, default=14)
    ap.add_argument("--phase", default=None, help="also list the opcodes of one phase (static counts)")
    args = ap.parse_args()
    src = os.path.join(ROOT, "raytracing_amd", "csrc")
    out = "/tmp/opcode_hist.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                    "-DRT_MEGA_WAVES_PER_SIMD=4", "-DRT_ISA_MARKS", "-DRT_LEAN", f"-I{ROOT}/include", f"-I{src}", "-S",
                    "--cuda-device-only", "-o", out, os.path.join(src, "rt_kernels.hip")], check=True, capture_output=True)
    lines = open(out).read().split("\n")
    name = f"_ZN12_GLOBAL__N_115render_mega_bvhILb{args.lds}ELb0ELb1ELb0EEE"
    start = next(i for i, l in enumerate(lines) if l.startswith(name) and l.rstrip().endswith(":") is False and ":" in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    marks = [(i, l.split("RT_MARK")[1].strip()) for i, l in enumerate(body) if "RT_MARK" in l]
    first = [m for m in marks]
    # the loop's first copy: up to the second "shade" mark
    second_shade = [i for i, n in marks if n == "shade"][1]
    marks = [(i, n) for i, n in marks if i < second_shade] + [(second_shade, "end")]
    per_phase = collections.defaultdict(collections.Counter)
    for (a, n), (b, _) in zip(marks, marks[1:]):
        phase = n
        if (b, _) == marks[-1]:
            phase = "sched+cold"  # the last stretch: the end of the traversal loop and blocks moved behind the loop
        for l in body[a:b]:
            m = re.match(r"\s+(v_[a-z0-9_]+)", l)
            if m:
                per_phase[phase][re.sub(r"_e(32|64)$", "", m.group(1))] += 1
    # cold blocks behind the loop are shading code (they hold the v_sin / v_cos of the scatter directions)
    weight = {"shade": args.shade, "start": args.start, "begin": args.start, "loop": args.shade, "node": args.node,
              "leaf": args.node, "sched": args.node, "sched+cold": args.shade}
    # (the leaf stretch is entered at every scheduling decision; its primitive test runs args.leaf times per 64 rays)
    total = collections.Counter()
    cyc_by_class = collections.Counter()
    dyn_phase, cyc_phase = {}, {}
    for phase, ops in per_phase.items():
        w = weight.get(phase, 1.0)
        if phase == "leaf":
            w = args.leaf
        n = sum(ops.values())
        dyn_phase[phase] = n * w
        c = 0.0
        for op, k in ops.items():
            cls, cy = classify(op)
            total[op] += k * w
            cyc_by_class[cls] += k * w * cy
            c += k * w * cy
        cyc_phase[phase] = c
    print(f"kernel {name}...  (per 64 rays; phase executions: node {args.node}, leaf {args.leaf}, shade/start {args.shade})")
    print(f"{'phase':12s} {'static VALU':>11s} {'x runs':>7s} {'dynamic':>9s} {'issue cycles':>13s}")
    for phase in ("shade", "sched+cold", "start", "begin", "loop", "sched", "node", "leaf"):
        if phase in per_phase:
            w = args.leaf if phase == "leaf" else weight.get(phase, 1.0)
            print(f"{phase:12s} {sum(per_phase[phase].values()):11d} {w:7.1f} {dyn_phase[phase]:9.0f} {cyc_phase[phase]:13.0f}")
    tot_i, tot_c = sum(dyn_phase.values()), sum(cyc_phase.values())
    print(f"{'total':12s} {'':11s} {'':7s} {tot_i:9.0f} {tot_c:13.0f}   = {tot_i / 64:.1f} wave-instructions and {tot_c / 64:.0f} issue cycles per ray, "
          f"{tot_c / tot_i:.2f} cycles per instruction")
    print("\nby issue-rate class (share of issue cycles):")
    for cls, c in cyc_by_class.most_common():
        print(f"  {cls:8s} {c / tot_c * 100:5.1f} %")
    print(f"\ntop {args.top} opcodes by issue cycles:")
    rows = sorted(((k * classify(op)[1], op, k) for op, k in total.items()), reverse=True)[:args.top]
    for c, op, k in rows:
        print(f"  {op:22s} {k:8.0f} instructions  {c / tot_c * 100:5.1f} % of the cycles  ({classify(op)[0]})")
    if args.phase and args.phase in per_phase:
        print(f"\nphase {args.phase} (static):")
        for op, k in per_phase[args.phase].most_common():
            print(f"  {op:22s} {k:4d}  ({classify(op)[0]})")


if __name__ == "__main__":
    main()

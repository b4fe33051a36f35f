import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import abi, binding
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
scene = sys.argv[1] if len(sys.argv) > 1 else "bouncing_spheres"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
if scene == "synth1m":  # the 10^6-primitive scene comes from the scene program (as in bench.py)
    import subprocess
    out = "/tmp/sched_synth1m.rtscene"
    prog = os.path.join(os.path.dirname(G), "..", "examples", "bin", "rt_scenes")
    subprocess.run([prog, "synth1m", "--half", "500", "--dump", out], check=True, capture_output=True)
    arrays = abi.SceneArrays.load(out); cam, _ = abi.load_camera_file(out + ".cam")
else:
    arrays = abi.SceneArrays.load(f"{G}/{scene}.rtscene"); cam, _ = abi.load_camera_file(f"{G}/{scene}.cam")
sc = binding.Scene(arrays)
acc, st = sc.render(cam, spp, counters=True)
acc, st0 = sc.render(cam, spp)
s = list(st.sched)
w = st.rays / 64
print(f"{scene}: rays {st.rays} box/ray {st.box_tests/st.rays:.1f} prim/ray {(st.sphere_tests+st.quad_tests)/st.rays:.2f} kernel_ms {st0.kernel_ms:.3f} (instrumented {st.kernel_ms:.3f})")
if os.environ.get("RT_AMD_SCHEDULER", "2") in ("1", "2") and s[0]:
    print(f"inner blocks {s[0]} ({s[0]/w:.1f} per 64 rays) active {s[1]/s[0]:.1f}/64 | leaf blocks {s[2]} ({s[2]/w:.1f} per 64 rays) active {s[3]/max(s[2],1):.1f}/64 | shade phases {s[4]} ({s[4]/w:.2f} per 64 rays) lanes {s[5]/max(s[4],1):.1f}/64")
elif s[0]:
    tot = s[0]
    print(f"wave cycles: trace {s[1]/tot:.3f} shade {s[2]/tot:.3f} rest(regen, loop) {(tot-s[1]-s[2])/tot:.3f} | wave iterations {s[3]} ({s[3]/w:.2f} per 64 rays) live lanes {s[4]/max(s[3],1):.1f}/64 | cycles/iter trace {s[1]/s[3]:.0f} shade {s[2]/s[3]:.0f} rest {(tot-s[1]-s[2])/s[3]:.0f}")
c = list(st.cycles)
if c[0]:
    print(f"wave cycles: shade {c[1]/c[0]:.3f} start paths {c[2]/c[0]:.3f} node steps {c[3]/c[0]:.3f} leaf tests {c[4]/c[0]:.3f} rest {(c[0]-c[1]-c[2]-c[3]-c[4])/c[0]:.3f}"
          f" | cycles per block: node x4 {c[3]/max(s[0],1):.0f} leaf {c[4]/max(s[2],1):.0f} shade {c[1]/max(s[4],1):.0f} start {c[2]/max(s[4],1):.0f}")
if st.coherence[0]:
    print(f"node steps: {st.coherence[1] / st.coherence[0]:.3f} of the participating lanes are at the same node as the wave's "
          f"first active lane ({st.coherence[0] / max(s[0], 1):.1f} lanes per step on average)")
print(f"launch profile (instrumented kernel): first wave out of new paths at {st.tail_us[0]:.0f} us, last at {st.tail_us[1]:.0f} us, last wave exit {st.tail_us[2]:.0f} us")
